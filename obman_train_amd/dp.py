"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL/xGMI.

The reference's only multi-GPU construct is single-process ``nn.DataParallel`` (``traineval.py:130``),
in practice broken.  DP semantics are therefore defined here the way DataParallel
replicas behave: every rank runs the whole path on its own shard of the batch (BatchNorm statistics and
the batch-global masked means stay rank-local) and only gradients are exchanged: mean over ranks.

MI355X specifics: xGMI is point-to-point, so a ring all-reduce is bound by one ~77 GB/s link direction;
the 51.7 MB of fp32 gradients are packed into flat >= 8 MB buckets (one multi-tensor copy per bucket), large
enough to run near link rate, followed by a few shrinking tail buckets so that only a 0.3 MB LAST bucket (layer1 + stem,
ready at the very end of backward) is exposed; the others are on the wire while ResNet's backward is still running.  Buckets
are laid out in reverse parameter order, all-reduced (AVG) asynchronously on RCCL's own stream as soon as their last
gradient has been accumulated, and waited for before the optimizer step (``finish()``).  With ``world_size == 1`` everything is a no-op.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    """Flat gradient buckets, one multi-tensor copy per bucket, no copy back.

    Autograd assigns fresh gradient tensors (``zero_grad`` = set to None, so no accumulate-add kernels); when the last
    gradient of a bucket has arrived its post-accumulate hook packs the whole bucket with ONE ``torch._foreach_copy_``
    launch, re-points every ``.grad`` at its slice of the flat buffer (strided like the parameter: fused optimizers need
    grad.layout == param.layout, conv filters are channels_last) and starts the asynchronous all-reduce (``AVG`` on RCCL,
    so no scaling pass).  ``finish()`` waits before the optimizer step.

    The collective sequence is STATIC: buckets are all-reduced strictly in index order (a bucket that completes early waits
    for its predecessors), so ranks whose autograd graphs differ in a step (a data-dependent branch active on one rank only)
    still issue identical collectives.  A parameter without a gradient on THIS rank contributes zeros.  Whether ANY rank had
    a gradient for it travels with the last bucket (one flag per parameter behind its gradients, 1 = "had one"): if some rank
    had, every rank ends the step with ``.grad`` = the averaged view; if none had, every rank restores ``.grad = None`` - so
    the optimizer treats such a parameter exactly as the single-process path does (no weight-decay / momentum update from a
    zero gradient).  Only a rank that itself lacked a gradient has to look at the flags (one small device->host read); the
    common step, where every parameter receives a gradient everywhere, pays nothing.  Parameters that can never receive a
    gradient (``base_net.fc``, unused by the feature extractor) are passed in ``exclude``: they are left out of the buckets
    and keep ``grad = None`` as in the reference."""

    def __init__(self, params, bucket_bytes=8 * 1024 * 1024, group=None, force=False, exclude=(), tail_bytes=512 * 1024,
                 accumulate_in_place=False):
        self.group = group
        # accumulate_in_place: ``zero_grad()`` zeroes the flat buffers (one fill per bucket) and leaves every ``.grad`` pointing
        # at its view, autograd then ADDS each gradient into the bucket (one add kernel per parameter) and there is no pack
        # copy.  Default off: measured slower on MI355X than stolen gradients + one multi-tensor copy per bucket
        # (profiles/r03_dp_overhead.md); kept as the alternative the measurement was made against.
        self.in_place = bool(accumulate_in_place)
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        skip = {id(p) for p in exclude}
        self.params = [p for p in params if p.requires_grad and id(p) not in skip]
        self.enabled = self.world > 1 or (force and live)  # force: single-rank self-test of the whole mechanism
        self.buckets = []      # (flat buffer, [(param, offset, numel)])
        self._where = {}
        self._pending = []
        self._works = []
        self._seen = set()
        self._missing = []     # parameters without a local gradient in this step
        self._lacked = False   # this rank had such parameters: read the flags after the last all-reduce
        self._next = 0         # next bucket to all-reduce (index order)
        if not self.enabled:
            return
        self.backend = dist.get_backend(group)
        self._avg = self.backend == "nccl"  # RCCL averages in the collective; gloo has no AVG
        # Bucket targets shrink towards the end of backward: the LAST bucket's all-reduce is the only exposed one (nothing is
        # left to hide it behind), so once less than one full bucket of gradients remains the target halves each time
        # (never below tail_bytes).  The configs[1] model: 13.7 / 9.0 / 9.5 / 9.0 MB, then 3.5 MB (layer3.0), 1.7 MB, 0.56 MB and a
        # final 0.32 MB (layer1 + stem) instead of one 6.1 MB bucket that waits for conv1's gradient.
        remaining = sum(p.numel() * p.element_size() for p in self.params)
        cur, cur_bytes = [], 0
        target = bucket_bytes if remaining > bucket_bytes else max(remaining // 2, tail_bytes)
        groups = []
        for p in reversed(self.params):  # backward produces the last layers' gradients first
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= target:
                groups.append(cur)
                remaining -= cur_bytes
                cur, cur_bytes = [], 0
                target = bucket_bytes if remaining > bucket_bytes else max(remaining // 2, tail_bytes)
        if cur:
            groups.append(cur)
        for k, plist in enumerate(groups):
            self._close(plist, extra=len(self.params) if k == len(groups) - 1 else 0)
        if self.buckets:
            flat = self.buckets[-1][0]
            self._flags = flat[flat.numel() - len(self.params):]  # one "some rank had a gradient" flag per parameter
            self._ones = torch.ones_like(self._flags)
            self._flag_of = {p: k for k, p in enumerate(self.params)}
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    @staticmethod
    def _view(flat, off, p):
        dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
        if dense:
            return torch.as_strided(flat, p.shape, p.stride(), off)
        return flat[off:off + p.numel()].view(p.shape)

    def _close(self, plist, extra=0):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total + extra, dtype=plist[0].dtype, device=plist[0].device)
        slots, off = [], 0
        for p in plist:
            slots.append((p, off, p.numel(), self._view(flat, off, p)))  # the views are built once, not per step
            self._where[p] = len(self.buckets)
            off += p.numel()
        self.buckets.append((flat, slots))
        self._pending.append(len(slots))

    def describe(self):
        """What a scaling run should be able to verify from the bench line: group size, backend, bucket layout."""
        if not self.enabled:
            return {"enabled": False, "world_size": self.world}
        return {"enabled": True, "world_size": self.world, "backend": self.backend,
                "reduce_op": "AVG" if self._avg else "SUM + 1/world",
                "parameters": len(self.params), "gradient_bytes": sum(p.numel() * p.element_size() for p in self.params),
                "bucket_bytes": [flat.numel() * flat.element_size() for flat, _ in self.buckets],
                "flag_words_in_last_bucket": len(self.params)}

    def zero_grad(self):
        if self.enabled and self.in_place:
            for flat, slots in self.buckets:
                flat.zero_()
                for p, off, n, v in slots:
                    if p.grad is not v:
                        p.grad = v
            return
        for p in self.params:
            p.grad = None

    def _launch(self, b):
        flat, slots = self.buckets[b]
        views, grads = [], []
        for p, off, n, v in slots:
            if p not in self._seen:  # no gradient arrived on this rank in this step: contributes zeros
                if p.grad is not v or not self.in_place:
                    v.zero_()
                self._missing.append(p)
            elif p.grad is not v:  # a stolen (fresh) gradient tensor: pack it; in-place mode accumulated into the view already
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if b == len(self.buckets) - 1:  # the flags ride with the last bucket: every local gradient of the step is known by now
            if self._missing:
                have = torch.ones(len(self.params), dtype=flat.dtype)
                for p in self._missing:
                    have[self._flag_of[p]] = 0.0
                self._flags.copy_(have)
                self._lacked = True
            else:
                views.append(self._flags)
                grads.append(self._ones)
        if views:
            torch._foreach_copy_(views, grads)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append((b, dist.all_reduce(flat, op=op, group=self.group, async_op=True)))

    def _drain(self, everything=False):
        while self._next < len(self.buckets) and (everything or self._pending[self._next] == 0):
            self._launch(self._next)
            self._next += 1

    def _on_grad(self, p):
        if p in self._seen:  # a second accumulation into the same parameter in one step (not on this path)
            return
        self._seen.add(p)
        self._pending[self._where[p]] -= 1
        self._drain()

    def finish(self):
        """Wait for every bucket (launching, in index order, the ones left open by gradient-less parameters).  Call before
        ``optimizer.step()``."""
        if not self.enabled:
            return
        self._drain(everything=True)
        for b, work in self._works:
            work.wait()
            if not self._avg:
                self.buckets[b][0].mul_(1.0 / self.world)
        if self._lacked:  # a parameter no rank had a gradient for keeps grad = None, as without data parallelism
            had = self._flags.cpu()
            for p in self._missing:
                if float(had[self._flag_of[p]]) == 0.0:
                    p.grad = None
        self._works = []
        self._seen = set()
        self._missing = []
        self._lacked = False
        self._next = 0
        self._pending = [len(slots) for _, slots in self.buckets]


def init_rccl(device, rank=None, world_size=None):
    """``init_process_group("nccl")`` (= RCCL) for one process per GPU, with the collectives on a HIGH-PRIORITY stream.

    Why the priority matters here: HIP multiplexes a process's streams onto a few hardware queues and kernels of one
    queue run back to back.  With the default (normal-priority) pool stream the RCCL kernels of this workload landed on
    the SAME hardware queue as the training stream (rocprofv3 ``Queue_Id``, ``profiles/r02_force_dist_trace.md``): every
    all-reduce kernel was serialised with ResNet's backward - zero overlap however early its bucket was ready.  A
    high-priority stream gets a queue of its own, and the few-workgroup ring kernels are scheduled ahead of the
    chip-filling convolution kernels instead of behind them."""
    import os

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by this driver for RCCL across processes
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    kw = {}
    if rank is not None:
        kw.update(rank=rank, world_size=world_size)
    dist.init_process_group("nccl", device_id=device, pg_options=opts, **kw)


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank ``src``'s weights and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
