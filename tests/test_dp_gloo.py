"""Multi-process data-parallel path on CPU (gloo, world_size 2): bucketed gradient all-reduce gives the
mean over ranks of the per-shard gradients (= the DataParallel-replica semantics DESIGN.md defines),
including parameters that never receive a gradient (like ``base_net.fc``) and several small buckets."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(12, 32), nn.ReLU(), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 3))
        self.unused = nn.Linear(5, 5)  # never used in forward: no gradient, like base_net.fc (passed in `exclude`)
        self.dormant = nn.Linear(4, 4)  # in the buckets, but no rank ever produces a gradient: must end with grad = None

    def forward(self, x):
        return self.body(x)


def _model():
    torch.manual_seed(0)
    return _Net()


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obman_train_amd.dp import GradientBuckets, broadcast_parameters

    net = _model()
    if rank == 1:  # desynchronise on purpose: broadcast must repair it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    broadcast_parameters(net)
    # direct_bytes=1024: the two larger weight matrices (1536 and 2048 bytes) are all-reduced in place, the rest is packed
    buckets = GradientBuckets(net.parameters(), bucket_bytes=256, exclude=net.unused.parameters(), direct_bytes=1024)
    plan = buckets.describe()
    assert plan["in_place_tensor_bytes"] == [2048, 1536] and "P" in plan["order"] and plan["order"].endswith("P"), plan
    assert len(buckets.buckets) >= 3
    # bucket targets shrink once less than one full bucket remains: the last (exposed) bucket is the smallest
    probe = GradientBuckets(nn.ModuleList([nn.Linear(8, 8) for _ in range(23)]).parameters(), bucket_bytes=1024, tail_bytes=64)
    sizes = [flat.numel() * 4 for flat, _ in probe.buckets]  # 23 x (256 + 32) bytes, walked from the last layer
    assert sizes == [1152] * 5 + [576, 288 + 46 * 4], sizes  # + one "some rank had a gradient" flag per parameter in the last one
    info = probe.describe()
    assert info["world_size"] == 2 and info["backend"] == "gloo" and info["packed_bucket_bytes"] == sizes and info["parameters"] == 46
    assert info["in_place_tensor_bytes"] == [] and info["order"] == "P" * 7
    x, y = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    for step in range(2):  # twice: bucket state must reset between steps
        if step == 0:
            buckets.zero_grad()
        else:
            net.zero_grad(set_to_none=True)  # a caller that detaches the views: zero_grad() must re-attach them
            buckets.zero_grad()
        loss = ((net(x[shard]) - y[shard]) ** 2).mean()
        loss.backward()
        buckets.finish()
    flat = buckets.buckets[buckets._where[net.body[0].bias]][0]
    assert flat.data_ptr() <= net.body[0].bias.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4  # a packed gradient lives in its bucket
    assert buckets.buckets[buckets._where[net.body[0].weight]][0] is None  # a big one was reduced where autograd left it
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    torch.save(grads, os.path.join(out_dir, "grads_%d.pt" % rank))
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_full_batch(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, start_method="spawn")
    net = _model()
    x, y = _data()
    ((net(x) - y) ** 2).mean().backward()
    want = {k: p.grad for k, p in net.named_parameters()}
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "grads_%d.pt" % rank))
        for k, g in want.items():
            if g is None:
                assert got[k] is None
            else:
                torch.testing.assert_close(got[k], g, rtol=1e-5, atol=1e-6)


class _Branchy(nn.Module):
    """``side`` only contributes when asked: a data-dependent branch (like the atlas / contact branches)."""

    def __init__(self):
        super().__init__()
        self.a, self.side, self.b = nn.Linear(6, 6), nn.Linear(6, 6), nn.Linear(6, 2)

    def forward(self, x, use_side):
        h = torch.relu(self.a(x))
        if use_side:
            h = h + self.side(h)
        return self.b(h)


def _branchy_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obman_train_amd.dp import GradientBuckets

    torch.manual_seed(0)
    net = _Branchy()
    buckets = GradientBuckets(net.parameters(), bucket_bytes=64)  # several buckets; `side` sits in the middle
    x = torch.ones(3, 6) * (rank + 1)
    buckets.zero_grad()
    net(x, use_side=(rank == 0)).sum().backward()  # rank 1 never touches `side`: its hooks never fire there
    buckets.finish()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    # the verdict a recorded data-parallel step is refused on must be the SAME on every rank: only rank 1 lacked
    verdict = {"lacked_somewhere": buckets.some_rank_lacked_a_gradient(), "lacked_here": buckets.last_missing}
    buckets.zero_grad()
    net(x, use_side=True).sum().backward()  # a step in which every rank produces every gradient
    buckets.finish()
    verdict["clean_step"] = buckets.some_rank_lacked_a_gradient()
    torch.save({"grads": grads, "verdict": verdict}, os.path.join(out_dir, "branchy_%d.pt" % rank))
    dist.destroy_process_group()


def test_rank_divergent_graphs_issue_identical_collectives(tmp_path):
    """A branch active on one rank only: the static bucket order keeps the collectives matched (no hang, no mismatched
    sizes) and every rank ends with the same averaged gradient for every bucketed parameter."""
    world, port = 2, _free_port()
    mp.start_processes(_branchy_worker, args=(world, port, str(tmp_path)), nprocs=world, start_method="spawn")
    torch.manual_seed(0)
    net = _Branchy()
    want = {}
    for rank in range(world):
        net.zero_grad()
        net(torch.ones(3, 6) * (rank + 1), use_side=(rank == 0)).sum().backward()
        for k, p in net.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            want[k] = want.get(k, 0) + g / world
    got0 = torch.load(os.path.join(str(tmp_path), "branchy_0.pt"))
    got1 = torch.load(os.path.join(str(tmp_path), "branchy_1.pt"))
    v0, v1 = got0["verdict"], got1["verdict"]
    assert v0["lacked_here"] == 0 and v1["lacked_here"] > 0  # the rank-local count differs ...
    assert v0["lacked_somewhere"] is True and v1["lacked_somewhere"] is True  # ... the reduced verdict does not
    assert v0["clean_step"] is False and v1["clean_step"] is False
    got0, got1 = got0["grads"], got1["grads"]
    for k in want:
        torch.testing.assert_close(got0[k], want[k], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got1[k], got0[k], rtol=0, atol=0)


def _in_place_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obman_train_amd.dp import GradientBuckets

    net = _model()
    buckets = GradientBuckets(net.parameters(), bucket_bytes=256, exclude=net.unused.parameters(), accumulate_in_place=True)
    x, y = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    for step in range(2):
        buckets.zero_grad()  # zero fills of the flat buffers; .grad stays the bucket view and autograd accumulates into it
        ((net(x[shard]) - y[shard]) ** 2).mean().backward()
        buckets.finish()
    torch.save({k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()},
               os.path.join(out_dir, "inplace_%d.pt" % rank))
    dist.destroy_process_group()


def test_accumulate_in_place_mode_gives_the_same_average(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_in_place_worker, args=(world, port, str(tmp_path)), nprocs=world, start_method="spawn")
    net = _model()
    x, y = _data()
    ((net(x) - y) ** 2).mean().backward()
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "inplace_%d.pt" % rank))
        for k, p in net.named_parameters():
            if p.grad is None:
                assert got[k] is None, k
            else:
                torch.testing.assert_close(got[k], p.grad, rtol=1e-5, atol=1e-6)


def test_single_process_is_a_noop():
    from obman_train_amd.dp import GradientBuckets

    net = _model()
    b = GradientBuckets(net.parameters())
    assert not b.enabled
    b.finish()


def _layout_worker(rank, world, port, out_dir):
    """A directly reduced channels_last filter whose gradient arrives CONTIGUOUS on one rank (a backward kernel that chose
    another dense layout) while the other rank has no gradient at all and contributes zeros laid out like the parameter."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obman_train_amd.dp import GradientBuckets

    torch.manual_seed(0)
    w = nn.Parameter(torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last))
    small = nn.Parameter(torch.randn(5))
    buckets = GradientBuckets([w, small], direct_bytes=512)  # w (1152 bytes) is all-reduced in place
    assert buckets.buckets[buckets._where[w]][0] is None
    buckets.zero_grad()
    g = torch.arange(8 * 4 * 3 * 3, dtype=torch.float32).reshape(8, 4, 3, 3)  # logical values; contiguous (NOT channels_last) memory
    if rank == 0:
        w.grad = g.clone()
        buckets._on_grad(w)  # what the post-accumulate hook does
        small.grad = torch.ones(5)
        buckets._on_grad(small)
    else:
        small.grad = torch.ones(5) * 3
        buckets._on_grad(small)  # w receives no gradient on this rank
    buckets.finish()
    torch.save({"w": w.grad.clone(), "stride": w.grad.stride(), "small": small.grad.clone()}, os.path.join(out_dir, "layout_%d.pt" % rank))
    dist.destroy_process_group()


def test_direct_gradient_is_normalised_to_the_parameter_layout(tmp_path):
    """the collective reduces raw memory, so every rank must hand over the same element order."""
    world, port = 2, _free_port()
    mp.start_processes(_layout_worker, args=(world, port, str(tmp_path)), nprocs=world, start_method="spawn")
    want = torch.arange(8 * 4 * 3 * 3, dtype=torch.float32).reshape(8, 4, 3, 3) / 2  # mean of (g, zeros), element by element
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "layout_%d.pt" % rank))
        torch.testing.assert_close(got["w"], want, rtol=0, atol=0)
        torch.testing.assert_close(got["small"], torch.full((5,), 2.0), rtol=0, atol=0)
