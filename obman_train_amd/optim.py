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
        stream = ops._stream()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            cache = self._plans.get(gi) if hasattr(self, "_plans") else None
            if cache is None or not self._refresh(cache, params):
                cache = self._build(params)
                if not hasattr(self, "_plans"):
                    self._plans = {}
                self._plans[gi] = cache
            b1, b2 = group["betas"]
            lr = group["lr"]
            if torch.is_tensor(lr):
                lr = float(lr)
            _lib.check(lib.obman_adam_step(cache["addr"], len(params), float(lr), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), stream), "obman_adam_step")
        return loss

    def _build(self, params):
        """The launch's descriptor table (host array of device pointers) for this set of parameters; everything but the gradient
        pointers is constant from step to step, so ``_refresh`` only rewrites those (the table was rebuilt field by field every step
        at first: ~0.5 ms of host time per step, in a step whose enqueue time is within 10 % of its GPU time)."""
        arr = (_lib.AdamTensor * len(params))()
        keep = []
        for i, p in enumerate(params):
            g = p.grad
            if g.is_sparse:
                raise RuntimeError("ObmanAdam does not support sparse gradients")
            if not ops._is_dense(p):
                raise RuntimeError("ObmanAdam needs dense parameters (contiguous or channels_last)")
            st = self._init_state(p)
            if st["exp_avg"].stride() != p.stride():  # a state loaded from a checkpoint written with another layout
                st["exp_avg"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st["exp_avg"])
                st["exp_avg_sq"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st["exp_avg_sq"])
            step = st["step"]
            if not (torch.is_tensor(step) and step.is_cuda and step.dtype == torch.float32):
                step = st["step"] = torch.as_tensor(float(step), dtype=torch.float32, device=p.device)
            sh = getattr(p, "_obman_shadow", None)
            if sh is not None and sh.stride() != p.stride():
                sh = p._obman_shadow = ops.bf16_shadow(p.detach())  # the parameter was re-laid out (channels_last) after the shadow was made
                p._obman_shadow_version = p._version
            if g.stride() != p.stride():
                g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)  # element order of the parameter
            a = arr[i]
            a.p, a.g, a.m, a.v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            a.shadow_bf16 = None if sh is None else sh.data_ptr()
            a.step, a.n = step.data_ptr(), p.numel()
            keep.append((p, st["exp_avg"], st["exp_avg_sq"], step, sh, g))
        return {"arr": arr, "addr": ctypes.addressof(arr), "keep": keep, "ids": [id(p) for p in params],
                "ptrs": [p.data_ptr() for p in params], "strides": [p.stride() for p in params]}

    def _refresh(self, cache, params):
        """Same parameters, same storage, same state tensors as when the table was built?  Then only the gradient pointers change."""
        if len(params) != len(cache["ids"]):
            return False
        arr, keep = cache["arr"], cache["keep"]
        for i, p in enumerate(params):
            kp, m, v, step, sh, _ = keep[i]
            if kp is not p or p.data_ptr() != cache["ptrs"][i] or p.stride() != cache["strides"][i]:
                return False
            st = self.state[p]
            if st.get("exp_avg") is not m or st.get("exp_avg_sq") is not v or st.get("step") is not step or getattr(p, "_obman_shadow", None) is not sh:
                return False  # load_state_dict / a re-created shadow: rebuild
            g = p.grad
            if g.stride() != cache["strides"][i]:
                g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
            arr[i].g = g.data_ptr()
            keep[i] = (kp, m, v, step, sh, g)  # the gradient stays referenced until the launch below has been enqueued
        return True


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
