"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL/xGMI.

The reference's only multi-GPU construct is single-process ``nn.DataParallel`` (``traineval.py:130``),
in practice broken.  DP semantics are therefore defined here the way DataParallel
replicas behave: every rank runs the whole path on its own shard of the batch (BatchNorm statistics and
the batch-global masked means stay rank-local) and only gradients are exchanged: mean over ranks.

MI355X specifics: xGMI is point-to-point, so a ring all-reduce is bound by one ~77 GB/s link direction;
gradient tensors of >= 1 MB (45 of the 51.7 MB) are all-reduced in place, each large enough to run near link rate; the
small ones are packed into flat buckets (one multi-tensor copy per bucket) with shrinking tail buckets, so that only a small
LAST bucket (layer1 + stem, ready at the very end of backward) is exposed; the others are on the wire while ResNet's backward is
still running.  Collectives are issued in reverse parameter order, asynchronously (AVG) on RCCL's own stream as soon as their
last gradient has been accumulated, and waited for before the optimizer step (``finish()``).  With ``world_size == 1`` everything is a no-op.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    """Gradient exchange plan: big gradients are all-reduced IN PLACE, small ones packed into flat buckets.

    Autograd assigns fresh gradient tensors (``zero_grad`` = set to None, so no accumulate-add kernels).
    * A parameter of at least ``direct_bytes`` (1 MB: the 3x3 filters of ResNet's last two stages, the big linear layers -
      45 of the 51.7 MB of the configs[1] model in 12 tensors) is a bucket of its own: the tensor autograd produced is handed
      to RCCL as it is - no pack copy, ``.grad`` untouched (measured: the multi-tensor pack copies of the previous all-packed
      plan cost 86 us of the training stream per step, profiles/r03_dp_overhead.md).
    * The remaining small gradients (BatchNorm parameters, biases, early filters: ~300 tensors in 6.7 MB) are packed into
      flat buckets with ONE ``torch._foreach_copy_`` launch per bucket when the bucket's last gradient has arrived; every
      ``.grad`` is re-pointed at its slice of the flat buffer (strided like the parameter: fused optimizers need
      grad.layout == param.layout, conv filters are channels_last).
    Either way the all-reduce is asynchronous (``AVG`` on RCCL, so no scaling pass); ``finish()`` waits before the optimizer step.

    The collective sequence is STATIC: buckets are all-reduced strictly in index order (a bucket that completes early waits
    for its predecessors), so ranks whose autograd graphs differ in a step (a data-dependent branch active on one rank only)
    still issue identical collectives.  A parameter without a gradient on THIS rank contributes zeros.  Whether ANY rank had
    a gradient for it travels with the last bucket (one flag per parameter behind its gradients, 1 = "had one"): if some rank
    had, every rank ends the step with ``.grad`` = the average; if none had, every rank restores ``.grad = None`` - so
    the optimizer treats such a parameter exactly as the single-process path does (no weight-decay / momentum update from a
    zero gradient).  Only a rank that itself lacked a gradient has to look at the flags (one small device->host read); the
    common step, where every parameter receives a gradient everywhere, pays nothing.  Parameters that can never receive a
    gradient (``base_net.fc``, unused by the feature extractor) are passed in ``exclude``: they are left out of the plan
    and keep ``grad = None`` as in the reference."""

    def __init__(self, params, bucket_bytes=8 * 1024 * 1024, group=None, force=False, exclude=(), tail_bytes=512 * 1024,
                 accumulate_in_place=False, direct_bytes=1024 * 1024):
        self.group = group
        # accumulate_in_place: ``zero_grad()`` zeroes the flat buffers (one fill per bucket) and leaves every packed ``.grad``
        # pointing at its view, autograd then ADDS each gradient into the bucket (one add kernel per parameter) and there is no
        # pack copy.  Default off: measured slower on MI355X than stolen gradients + one multi-tensor copy per bucket
        # (11.74 vs 11.56 ms per step, profiles/r03_dp_overhead.md); kept as the alternative the measurement was made against.
        self.in_place = bool(accumulate_in_place)
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        skip = {id(p) for p in exclude}
        self.params = [p for p in params if p.requires_grad and id(p) not in skip]
        self.enabled = self.world > 1 or (force and live)  # force: single-rank self-test of the whole mechanism
        self.buckets = []      # packed: (flat buffer, [[param, offset, numel, view]]) / direct: (None, param)
        self._where = {}
        self._pending = []
        self._works = []
        self._seen = set()
        self._missing = []     # parameters without a local gradient in this step
        self._lacked = False   # this rank had such parameters: read the flags after the last all-reduce
        self._next = 0         # next bucket to all-reduce (index order)
        self.capturing = False  # inside a stream capture: nothing that needs the host may run (no gradient-less parameters)
        self.last_missing = 0
        if not self.enabled:
            return
        self.backend = dist.get_backend(group)
        import os

        # measurement aid for the 1-rank self-test only (`bench.py --force-dist`): hooks, packing and views as usual, collectives skipped
        self._no_reduce = self.world == 1 and os.environ.get("OBMAN_DP_DEBUG") == "noreduce"
        self._avg = self.backend == "nccl"  # RCCL averages in the collective; gloo has no AVG
        # Packed-bucket targets shrink towards the end of backward: the LAST bucket's all-reduce is the only exposed one
        # (nothing is left to hide it behind), so once less than one full bucket of small gradients remains the target halves
        # each time (never below tail_bytes).
        nbytes = lambda p: p.numel() * p.element_size()  # noqa: E731
        big = (lambda p: nbytes(p) >= direct_bytes) if direct_bytes and direct_bytes > 0 else (lambda p: False)
        remaining = sum(nbytes(p) for p in self.params if not big(p))
        cur, cur_bytes = [], 0
        target = bucket_bytes if remaining > bucket_bytes else max(remaining // 2, tail_bytes)
        plan = []  # ("D", param) | ("P", [params]) in launch order
        for p in reversed(self.params):  # backward produces the last layers' gradients first
            if big(p):  # ready before the small gradients still being collected: takes the next index
                plan.append(("D", p))
                continue
            cur.append(p)
            cur_bytes += nbytes(p)
            if cur_bytes >= target:
                plan.append(("P", cur))
                remaining -= cur_bytes
                cur, cur_bytes = [], 0
                target = bucket_bytes if remaining > bucket_bytes else max(remaining // 2, tail_bytes)
        if cur or not plan or plan[-1][0] == "D":
            plan.append(("P", cur))  # the last collective must be a packed one: it carries the flags
        for k, (kind, item) in enumerate(plan):
            if kind == "D":
                self._where[item] = len(self.buckets)
                self.buckets.append((None, item))
                self._pending.append(1)
            else:
                self._close(item, extra=len(self.params) if k == len(plan) - 1 else 0)
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
        ref = plist[0] if plist else self.params[0]
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total + extra, dtype=ref.dtype, device=ref.device)
        slots, off = [], 0
        for p in plist:
            slots.append([p, off, p.numel(), None])  # the view is built at first use (the encoder re-lays its filters out as
            #                                          channels_last in its first forward) and kept while the strides still match
            self._where[p] = len(self.buckets)
            off += p.numel()
        self.buckets.append((flat, slots))
        self._pending.append(len(slots))

    def describe(self):
        """What a scaling run should be able to verify from the bench line: group size, backend, exchange plan."""
        if not self.enabled:
            return {"enabled": False, "world_size": self.world}
        nbytes = lambda p: p.numel() * p.element_size()  # noqa: E731
        return {"enabled": True, "world_size": self.world, "backend": self.backend,
                "reduce_op": "AVG" if self._avg else "SUM + 1/world",
                "parameters": len(self.params), "gradient_bytes": sum(nbytes(p) for p in self.params),
                "collectives_per_step": len(self.buckets),
                "in_place_tensor_bytes": [nbytes(b[1]) for b in self.buckets if b[0] is None],
                "packed_bucket_bytes": [b[0].numel() * b[0].element_size() for b in self.buckets if b[0] is not None],
                "order": "".join("D" if b[0] is None else "P" for b in self.buckets),
                "flag_words_in_last_bucket": len(self.params)}

    def _slot_view(self, flat, slot):
        p, off, n, v = slot
        if v is None or v.stride() != p.stride():
            v = slot[3] = self._view(flat, off, p)
        return v

    def zero_grad(self):
        if self.enabled and self.in_place:
            for flat, slots in self.buckets:
                if flat is None:
                    slots.grad = None  # direct: `slots` is the parameter
                    continue
                flat.zero_()
                for slot in slots:
                    v = self._slot_view(flat, slot)
                    if slot[0].grad is not v:
                        slot[0].grad = v
            return
        for p in self.params:
            p.grad = None

    def _launch(self, b):
        flat, slots = self.buckets[b]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if flat is None:  # one big gradient, reduced where autograd left it
            p = slots
            if p not in self._seen or p.grad is None:
                p.grad = torch.zeros_like(p)
                self._missing.append(p)
            elif p.grad.stride() != p.stride():
                # RCCL reduces raw memory: every rank must hand over the SAME element order.  The zero contribution above is
                # laid out like the parameter, so the produced gradient is normalised to the parameter's strides too (autograd
                # normally does that already; a contiguous gradient for a channels_last filter would otherwise be averaged
                # element-permuted against another rank's zeros)
                p.grad = torch.empty_like(p).copy_(p.grad)
            if not self._no_reduce:
                self._works.append((b, dist.all_reduce(p.grad, op=op, group=self.group, async_op=True)))
            return
        views, grads = [], []
        for slot in slots:
            p = slot[0]
            v = self._slot_view(flat, slot)
            if p not in self._seen:  # no gradient arrived on this rank in this step: contributes zeros
                if p.grad is not v or not self.in_place:
                    v.zero_()
                self._missing.append(p)
            elif p.grad is not v:  # a stolen (fresh) gradient tensor: pack it; in-place mode accumulated into the view already
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if b == len(self.buckets) - 1:  # the flags ride with the last bucket: every local gradient of the step is known by now
            if self._missing and self.capturing:
                raise RuntimeError("GradientBuckets: %d parameter(s) received no gradient while the step was being recorded into a "
                                   "graph (first: shape %s).  A recorded data-parallel step must have a fixed autograd graph on "
                                   "every rank: the 'some rank had a gradient' flags need a host round trip.  Pass such parameters "
                                   "in `exclude` or run the step eagerly." % (len(self._missing), tuple(self._missing[0].shape)))
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
        if not self._no_reduce:
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
                flat, slots = self.buckets[b]
                (slots.grad if flat is None else flat).mul_(1.0 / self.world)
        if self._lacked:  # a parameter no rank had a gradient for keeps grad = None, as without data parallelism
            had = self._flags.cpu()
            for p in self._missing:
                if float(had[self._flag_of[p]]) == 0.0:
                    p.grad = None
        self.last_missing = len(self._missing)  # parameters that had no local gradient in the step just finished
        self._works = []
        self._seen = set()
        self._missing = []
        self._lacked = False
        self._next = 0
        self._pending = [1 if flat is None else len(slots) for flat, slots in self.buckets]


    def some_rank_lacked_a_gradient(self):
        """After ``finish()``: did ANY rank of the group lack a gradient for a planned parameter in the step just finished?
        Read from the REDUCED flag words of the last bucket (mean over ranks of "had one": below 1 = some rank had none), so every
        rank gets the same answer without an extra collective - a refusal based on it is raised by all ranks together (the
        rank-local ``last_missing`` is not: a rank that raises alone leaves its peers inside collectives that never complete).
        One small device->host read; not for the recorded path."""
        if not self.enabled:
            return False
        if self._no_reduce:
            return bool(self.last_missing)
        return bool(float(self._flags.min().item()) < 1.0 - 0.5 / self.world)  # one lacking rank lowers the mean by 1 / world


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
    # the flight recorder's status record is how trainer.wait_for_watchdog SEES the watchdog thread drop finished eager work before
    # a step with RCCL collectives is recorded into a hipGraph (a 64-entry ring; one record per collective, no stack traces)
    os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "64")
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "64")
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
