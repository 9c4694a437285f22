"""Debug 3: how often does the two-process split-graph test go wrong, and does a device synchronisation around the exchange cure it?"""
import os, sys, subprocess, tempfile
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, out_dir, sync):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OBMAN_MANO_SYNTHETIC="1")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from obman_train_amd.dp import GradientBuckets, broadcast_parameters
    from obman_train_amd.dp_selftest import stage_collectives_through_host_if_needed
    from obman_train_amd.synthetic import make_batch
    from obman_train_amd.trainer import GraphedTrainStep, make_optimizer
    from tests.test_dp_graph_gpu import _build
    mode = stage_collectives_through_host_if_needed(torch.device("cuda", 0))
    model = _build()
    broadcast_parameters(model)
    opt = make_optimizer(model, "adam", lr=1e-4, capturable=True)
    buckets = GradientBuckets(model.parameters(), bucket_bytes=4 * 1024 * 1024, exclude=model.unused_parameters())
    sample = make_batch(4, torch.device("cuda", 0), seed=20 + rank, image_size=64)
    total, results, losses = model.forward(sample)
    buckets.zero_grad(); total.backward(); buckets.finish()
    eager = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    del total, results, losses
    step = GraphedTrainStep(model, opt, sample, warmup=1, restore_state=True, buckets=buckets)
    for k, v in sample.items():
        if torch.is_tensor(v):
            step.static[k].copy_(v)
    step.graph.replay()
    if sync:
        torch.cuda.synchronize()
    names = {p: k for k, p in model.named_parameters()}
    local_bad = [names[p] for p, g in step._grads.items() if not bool(torch.isfinite(g).all()) or float(g.abs().max()) > 1e8]
    buckets.exchange()
    if sync:
        torch.cuda.synchronize()
    bad = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        err = float((p.grad.detach().cpu() - eager[k]).abs().max() / eager[k].abs().max().clamp_min(1e-30))
        if not err < 1e-3:
            bad.append(k)
    print("rank %d mode %s sync %d: local garbage %d %s | after exchange wrong %d" % (rank, mode, sync, len(local_bad), local_bad[:3], len(bad)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    for sync in (0, 1, 0, 1, 0, 1):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        try:
            mp.start_processes(worker, args=(2, port, tempfile.mkdtemp(), sync), nprocs=2, start_method="spawn")
        except Exception as e:
            print("run failed", type(e).__name__, str(e)[:200], flush=True)
