"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL/xGMI.

The reference's only multi-GPU construct is single-process ``nn.DataParallel`` (``traineval.py:130``),
in practice broken.  DP semantics are therefore defined here the way DataParallel
replicas behave: every rank runs the whole path on its own shard of the batch (BatchNorm statistics and
the batch-global masked means stay rank-local) and only gradients are exchanged: mean over ranks.

MI355X specifics: xGMI is point-to-point, so a ring all-reduce is bound by one ~77 GB/s link direction;
the 51.7 MB of fp32 gradients are packed into flat 8 MB buckets: large enough to run near
link rate, small enough that only the LAST bucket (conv1/layer1, ready at the very end of backward, < 8 MB
~ 0.2 ms on the ring) is exposed; the others are on the wire while ResNet's backward is still running.  Buckets are filled in reverse parameter order by
post-accumulate-grad hooks, reduced asynchronously on RCCL's own stream, and copied back before the
optimizer step (``finish()``).  With ``world_size == 1`` everything is a no-op.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    def __init__(self, params, bucket_bytes=8 * 1024 * 1024, group=None, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.enabled = self.world > 1 or (force and dist.is_available() and dist.is_initialized())  # force: 1-rank self-test
        self.buckets = []      # (flat buffer, [(param, offset, numel)])
        self._where = {}
        self._pending = []
        self._works = []
        if not self.enabled:
            return
        cur, cur_bytes = [], 0
        for p in reversed(self.params):  # backward produces the last layers' gradients first
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        slots, off = [], 0
        for p in plist:
            slots.append((p, off, p.numel()))
            self._where[p] = (len(self.buckets), off)
            off += p.numel()
        self.buckets.append((flat, slots))
        self._pending.append(len(slots))

    def _on_grad(self, p):
        b, off = self._where[p]
        flat, slots = self.buckets[b]
        flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._works.append((b, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def finish(self):
        """Wait for the reductions, write mean gradients back into ``param.grad``.  Call before ``optimizer.step()``."""
        if not self.enabled:
            return
        # parameters that received no gradient this step (e.g. base_net.fc) leave their bucket open: flush
        for b, left in enumerate(self._pending):
            if left > 0:
                flat, slots = self.buckets[b]
                for p, off, n in slots:
                    if p.grad is None:
                        flat[off:off + n].zero_()
                self._works.append((b, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        inv = 1.0 / self.world
        for b, work in self._works:
            work.wait()
            flat, slots = self.buckets[b]
            for p, off, n in slots:
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad) * inv)
        self._works = []
        self._pending = [len(slots) for _, slots in self.buckets]


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank ``src``'s weights and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
