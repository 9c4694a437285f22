"""SELF-TEST harness for the data-parallel path on boxes with fewer GPUs than ranks (never imported by the product path).

``bench.py --backend gloo`` and the two-process GPU tests run several ranks on ONE device over gloo.  Where this torch build's
gloo cannot reduce device tensors, ``stage_collectives_through_host_if_needed`` replaces ``torch.distributed.all_reduce`` /
``broadcast`` PROCESS-WIDE by versions that reduce a host copy and write it back - which is why it lives here and not in
``dp.py``: the product (``dp.GradientBuckets`` over RCCL, ``dp.init_rccl``) issues exactly the same calls and never patches
anything."""
import torch
import torch.distributed as dist


def stage_collectives_through_host_if_needed(device):
    """SELF-TEST AID for the gloo backend (``bench.py --backend gloo``, the two-process GPU tests): gloo with device tensors
    works on torch builds whose gloo has the HIP transport; otherwise ``dist.all_reduce`` / ``dist.broadcast`` are wrapped so
    that they reduce / broadcast a host copy and write it back (synchronous; ``async_op=True`` returns an already-finished
    handle).  The callers - ``GradientBuckets``, ``broadcast_parameters``, ``bench.py`` - issue exactly the calls they issue on
    RCCL.  Never used with the ``nccl`` backend.  Returns a description of what is in effect."""
    if dist.get_backend() != "gloo":
        raise RuntimeError("host staging is a gloo self-test aid; the product path is RCCL (init_rccl)")
    try:
        probe = torch.ones(2, device=device)
        dist.all_reduce(probe)
        if float(probe[0]) == float(dist.get_world_size()):
            return "on device tensors"
    except Exception:  # noqa: BLE001 - any backend error means "not supported here"
        pass
    real_reduce, real_bcast = dist.all_reduce, dist.broadcast

    class _Done:
        def wait(self):
            return True

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return real_reduce(t, op=op, group=group, async_op=async_op)
        host = t.detach().cpu()
        real_reduce(host, op=op, group=group)
        t.copy_(host)
        return _Done() if async_op else None

    def broadcast(t, src=0, group=None, async_op=False):
        if not t.is_cuda:
            return real_bcast(t, src=src, group=group, async_op=async_op)
        host = t.detach().cpu()
        real_bcast(host, src=src, group=group)
        t.copy_(host)
        return _Done() if async_op else None

    dist.all_reduce, dist.broadcast = all_reduce, broadcast
    return "staged through host memory (self-test harness)"
