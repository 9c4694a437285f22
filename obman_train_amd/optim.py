"""The optimizer step of the training loop as ONE multi-tensor HIP kernel (``csrc/stepops.hip``, ``obman_adam_step``).

The reference builds ``torch.optim.Adam(model.parameters(), lr, weight_decay)`` (``traineval.py:112-116``) and calls ``step()`` once
per batch (``epochpass3d.py:86-91``).  ``ObmanAdam`` IS a ``torch.optim.Adam`` (constructor, ``param_groups``, ``state_dict`` /
``load_state_dict`` layout with ``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter - checkpoints written by either load into the
other), with ``step()`` replaced:

* every parameter that has a gradient is updated by one launch per <= 64 tensors (torch's fused Adam: three launches at 2.1 TB/s
  over the 350 MB it touches; this kernel streams 16-byte vectors of p, g, m, v);
* the per-parameter step counters live on the device (one fp32 array), so the step records into a hipGraph like torch's
  ``capturable=True`` form, without its per-step host work;
* a parameter that carries a bf16 SHADOW (``attach_bf16_shadows``: the filters of a bf16-autocast encoder) gets
  ``shadow = bf16(p_new)`` written by the same kernel - the autocast encoder then reads the shadow
  (``ops.shadow_conv2d``) instead of launching one cast kernel per filter and step.
No CPU path: parameters must be fp32 ROCm tensors (``trainer.make_optimizer`` only builds this optimizer for such models).
"""
import ctypes

import torch

from . import _lib, ops


class ObmanAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kw):
        if amsgrad or kw.get("maximize"):
            raise ValueError("ObmanAdam implements plain Adam (traineval.py:112-116): no amsgrad / maximize")
        kw.pop("fused", None)
        kw.pop("capturable", None)
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False,
                         capturable=True)  # capturable: ``step`` entries are device tensors, as this kernel keeps them
        for group in self.param_groups:
            for p in group["params"]:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise _lib.ObmanHipError("ObmanAdam needs fp32 ROCm parameters (there is no CPU path)")

    def _init_state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        stream = torch.cuda.current_stream().cuda_stream
        for group in self.param_groups:
            todo = []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("ObmanAdam does not support sparse gradients")
                st = self._init_state(p)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if g.stride() != p.stride() or not ops._is_dense(p):
                    if not ops._is_dense(p):
                        raise RuntimeError("ObmanAdam needs dense parameters (contiguous or channels_last)")
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)  # element order of the parameter
                if m.stride() != p.stride():  # a state loaded from a checkpoint written with another layout
                    m = st["exp_avg"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(m)
                    v = st["exp_avg_sq"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(v)
                step = st["step"]
                if not (torch.is_tensor(step) and step.is_cuda and step.dtype == torch.float32):
                    step = st["step"] = torch.as_tensor(float(step), dtype=torch.float32, device=p.device)
                sh = getattr(p, "_obman_shadow", None)
                if sh is not None and sh.stride() != p.stride():
                    sh = p._obman_shadow = ops.bf16_shadow(p.detach())  # the parameter was re-laid out (channels_last) after the shadow was made
                    p._obman_shadow_version = p._version
                todo.append((p, g, m, v, sh, step))
            if not todo:
                continue
            arr = (_lib.AdamTensor * len(todo))()
            for i, (p, g, m, v, sh, step) in enumerate(todo):
                a = arr[i]
                a.p, a.g, a.m, a.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                a.shadow_bf16 = None if sh is None else sh.data_ptr()
                a.step, a.n = step.data_ptr(), p.numel()
            b1, b2 = group["betas"]
            lr = group["lr"]
            if torch.is_tensor(lr):
                lr = float(lr)
            _lib.check(lib.obman_adam_step(ctypes.addressof(arr), len(todo), float(lr), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), stream), "obman_adam_step")
        return loss


def attach_bf16_shadows(module):
    """Give every bias-free ``nn.Conv2d`` filter of ``module`` (the bf16-autocast encoder) a bf16 shadow in the filter's own memory
    order (``weight._obman_shadow``).  ``ObmanAdam.step`` keeps the shadows current; ``ops.shadow_conv2d`` reads them.  Call again
    after anything else writes the filters (``load_state_dict``, a broadcast): ``refresh_bf16_shadows``."""
    n = 0
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d) and m.bias is None and m.weight.is_cuda and m.weight.dtype == torch.float32:
            m.weight._obman_shadow = ops.bf16_shadow(m.weight.detach())
            m.weight._obman_shadow_version = m.weight._version
            n += 1
    return n


def refresh_bf16_shadows(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d) and getattr(m.weight, "_obman_shadow", None) is not None:
            if m.weight._obman_shadow.stride() != m.weight.stride():
                m.weight._obman_shadow = ops.bf16_shadow(m.weight.detach())
            else:
                ops.bf16_shadow(m.weight.detach(), out=m.weight._obman_shadow)  # in place: a recorded graph keeps its address
            m.weight._obman_shadow_version = m.weight._version


def detach_bf16_shadows(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d) and hasattr(m.weight, "_obman_shadow"):
            del m.weight._obman_shadow
