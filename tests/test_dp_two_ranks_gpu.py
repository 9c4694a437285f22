"""Two data-parallel ranks running the REAL HandNet (BASELINE configs[3] semantics) - on the one GPU of the
test box: two processes share cuda:0, the process group is gloo (RCCL refuses two ranks on one device), and when this
torch build's gloo cannot reduce device tensors the test harness stages the collective through host memory (harness only:
the product path `dp.GradientBuckets` is unchanged and issues the same `dist.all_reduce` calls it issues on RCCL).

What is held: ranks get different shards; the gradient every rank ends the step with == the mean over ranks of the gradients
two independent single-process runs produce on those shards (``nn.DataParallel``-replica semantics, `traineval.py:130`);
BatchNorm running statistics stay rank-local; ``base_net.fc`` keeps ``grad = None``; a shard WITHOUT object points (atlas +
contact branches inactive on that rank only: a rank-divergent autograd graph) neither deadlocks nor mismatches the
collectives; after two Adam steps both ranks hold bit-identical weights."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

IMAGE, BATCH = 64, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build():
    import warnings

    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.synthetic import CONFIGS

    warnings.simplefilter("ignore")
    # the same convolution solutions in every process, and none that accumulate with atomics: the two workers and the
    # reference runs below then differ by round-off of the averaging only, and the comparison can be tight
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    torch.manual_seed(0)
    return HandNet(**CONFIGS["c3p1"]).to("cuda:0").train()


def _shard(rank):
    from obman_train_amd.queries import TransQueries
    from obman_train_amd.synthetic import make_batch

    sample = make_batch(BATCH, torch.device("cuda", 0), seed=20 + rank, image_size=IMAGE)
    if rank == 1:  # no object annotation in this shard: HandNet skips the atlas and contact branches (handnet.py:244-246)
        del sample[TransQueries.objpoints3d]
    return sample


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OBMAN_MANO_SYNTHETIC="1")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from obman_train_amd.dp import GradientBuckets, broadcast_parameters
    from obman_train_amd.dp_selftest import stage_collectives_through_host_if_needed

    mode = stage_collectives_through_host_if_needed(torch.device("cuda", 0))
    from obman_train_amd.trainer import make_optimizer, train_step

    model = _build()
    if rank == 1:  # desynchronise on purpose: the broadcast must repair it
        with torch.no_grad():
            model.mano_branch.pose_reg.weight.add_(1.0)
    broadcast_parameters(model)
    opt = make_optimizer(model, "adam", lr=1e-4)
    buckets = GradientBuckets(model.parameters(), bucket_bytes=4 * 1024 * 1024, exclude=model.unused_parameters())
    assert buckets.enabled and len(buckets.buckets) >= 3
    sample = _shard(rank)
    # step 1 without the optimizer: the averaged gradients and this rank's BatchNorm statistics
    total, _, _ = model.forward(sample)
    buckets.zero_grad()
    total.backward()
    buckets.finish()
    out = {"mode": mode, "loss": float(total),
           "grads": {k: (p.grad.detach().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()},
           "bn": {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.endswith("running_mean")}}
    # then two full train steps: every rank must apply the same update
    for _ in range(2):
        train_step(model, opt, sample, buckets)
    out["weights"] = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    torch.save(out, os.path.join(out_dir, "rank_%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_handnet_average_per_shard_gradients(tmp_path):
    import torch.multiprocessing as mp

    from tests.conftest import record_measurement

    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, start_method="spawn")
    got = [torch.load(os.path.join(str(tmp_path), "rank_%d.pt" % r)) for r in range(world)]

    # independent single-process runs of the same shards, same initial weights
    want_grads, want_bn, losses = {}, [], []
    for rank in range(world):
        model = _build()
        total, _, _ = model.forward(_shard(rank))
        total.backward()
        losses.append(float(total))
        for k, p in model.named_parameters():
            if p.grad is not None:
                want_grads[k] = want_grads.get(k, 0) + p.grad.detach().cpu() / world
            elif k not in want_grads:
                want_grads[k] = None
        want_bn.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.endswith("running_mean")})
        del model

    atlas_only = [k for k in want_grads if k.startswith("atlas_branch.decoder")]
    assert atlas_only and all(want_grads[k] is not None for k in atlas_only)
    worst = 0.0
    for r in range(world):
        assert abs(got[r]["loss"] - losses[r]) <= 1e-4 * abs(losses[r]), (r, got[r]["loss"], losses[r])  # each rank computed ITS shard
        for k, w in want_grads.items():
            g = got[r]["grads"][k]
            if w is None:
                assert g is None, k  # base_net.fc (excluded) and anything no rank produced a gradient for
                continue
            err = (g - w).abs().max().item() / max(w.abs().max().item(), 1e-30)
            worst = max(worst, err)
            # deterministic convolution solutions on both sides (see _build): what is left is the fp32 rounding of (a + b) / 2
            # against a / 2 + b / 2, plus at most one flipped ReLU / contact-mask element from a last-bit difference
            assert err <= 2e-4, (r, k, err)
        for k, v in want_bn[r].items():  # BatchNorm statistics are those of the rank's own shard ...
            torch.testing.assert_close(got[r]["bn"][k], v, rtol=1e-4, atol=1e-6)
    assert got[0]["grads"]["base_net.fc.weight"] is None and got[1]["grads"]["base_net.fc.weight"] is None
    k = "base_net.bn1.running_mean"  # ... and differ between the ranks
    assert not torch.allclose(got[0]["bn"][k], got[1]["bn"][k], rtol=1e-3, atol=1e-7)
    for k in want_grads:  # the all-reduce result is the same bits everywhere, so two Adam steps keep the replicas identical
        if want_grads[k] is not None:
            assert torch.equal(got[0]["grads"][k], got[1]["grads"][k]), k
        assert torch.equal(got[0]["weights"][k], got[1]["weights"][k]), k
    record_measurement("dp_two_ranks_real_handnet", {"collectives": got[0]["mode"], "worst_grad_err_of_max": worst})
