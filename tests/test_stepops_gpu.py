"""GPU: the round-6 step operators (csrc/stepops.hip) against the stock torch operations they replace.

* K11 ``optim.ObmanAdam`` == ``torch.optim.Adam`` (traineval.py:112-116) over several steps, tensors of awkward sizes, a
  channels_last filter, weight decay, a parameter that skips steps, checkpoints that move between the two optimizers, bf16 shadows
  bit-equal to ``p.bfloat16()``;
* K12 ``ops.affine_points`` == ``scale.unsqueeze(1) * verts + trans.unsqueeze(1)`` (atlasbranch.py:133-138), bit-exact forward;
* K13 ``ops.mse_terms`` == ``torch_f.mse_loss`` per term (manobranch.py:251-318, atlasbranch.py:211-228);
* K14 ``ops.gt_object_stats`` == ``gt.mean(1)``, ``gt - centroids``, ``norm(centred, 2, 2).max(1)`` (atlasbranch.py:211-222);
* ``ops.shadow_conv2d`` == the autocast convolution it replaces, and notices a filter somebody else has written.
The torch side runs on the HOST in fp32 (the oracle convention of this suite: the reference's CPU path)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _params(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 7, 7), (128, 64, 3, 3), (257,), (1, 1), (1031, 17), (5,), (512, 515, 1), (3, 128, 1)]
    ps = []
    for i, s in enumerate(shapes):
        t = torch.randn(s, generator=g) * (0.1 + i)
        ps.append(t)
    return ps


@pytest.mark.parametrize("weight_decay", [0.0, 1e-2])
def test_adam_matches_torch_adam(weight_decay):
    from obman_train_amd.optim import ObmanAdam

    dev = torch.device("cuda", 0)
    host = [torch.nn.Parameter(t.clone()) for t in _params(0, dev)]
    gpu = [torch.nn.Parameter(t.clone().to(dev)) for t in _params(0, dev)]
    gpu[1].data = gpu[1].data.contiguous(memory_format=torch.channels_last)  # a channels_last filter (the encoder's layout)
    ref = torch.optim.Adam(host, lr=3e-3, weight_decay=weight_decay)
    opt = ObmanAdam(gpu, lr=3e-3, weight_decay=weight_decay)
    g = torch.Generator().manual_seed(7)
    for step in range(12):
        for i, (h, d) in enumerate(zip(host, gpu)):
            if i == 3 and step % 3 == 1:  # this parameter skips a step: its own counter must not advance
                h.grad, d.grad = None, None
                continue
            gr = torch.randn(h.shape, generator=g) * (1.0 + step)
            h.grad = gr.clone()
            d.grad = gr.to(dev).contiguous(memory_format=torch.channels_last) if i == 1 else gr.to(dev)
        if step == 5:
            for grp in ref.param_groups + opt.param_groups:
                grp["lr"] = 1e-3  # a scheduler step
        ref.step()
        opt.step()
    torch.cuda.synchronize()
    for h, d in zip(host, gpu):
        torch.testing.assert_close(d.detach().cpu(), h.detach(), rtol=1e-5, atol=2e-6)
        # the moments are sums with cancellation (m + (1 - b1)(g - m)): one ulp of the LARGEST addend (gradients up to ~50 here)
        ma, va = float(ref.state[h]["exp_avg"].abs().max()), float(ref.state[h]["exp_avg_sq"].abs().max())
        torch.testing.assert_close(opt.state[d]["exp_avg"].cpu(), ref.state[h]["exp_avg"], rtol=1e-5, atol=1e-5 * max(ma, 1.0))
        torch.testing.assert_close(opt.state[d]["exp_avg_sq"].cpu(), ref.state[h]["exp_avg_sq"], rtol=1e-5, atol=1e-6 * max(va, 1.0))
        assert float(opt.state[d]["step"]) == float(ref.state[h]["step"])
    assert float(opt.state[gpu[3]]["step"]) == 8.0


def test_adam_checkpoints_move_between_the_two_optimizers():
    from obman_train_amd.optim import ObmanAdam

    dev = torch.device("cuda", 0)
    a = [torch.nn.Parameter(t.clone().to(dev)) for t in _params(1, dev)]
    b = [torch.nn.Parameter(t.clone().to(dev)) for t in _params(1, dev)]
    ours, theirs = ObmanAdam(a, lr=1e-3), torch.optim.Adam(b, lr=1e-3)
    g = torch.Generator().manual_seed(3)

    def grads():
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(dev)
            x.grad, y.grad = gr.clone(), gr.clone()

    for _ in range(3):
        grads()
        ours.step()
        theirs.step()
    # swap the optimizer states through state_dict() and keep stepping: trajectories must stay together
    sd_ours, sd_theirs = copy.deepcopy(ours.state_dict()), copy.deepcopy(theirs.state_dict())
    ours2, theirs2 = ObmanAdam(a, lr=1e-3), torch.optim.Adam(b, lr=1e-3)
    ours2.load_state_dict(sd_theirs)
    theirs2.load_state_dict(sd_ours)
    for _ in range(3):
        grads()
        ours2.step()
        theirs2.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(x.detach(), y.detach(), rtol=3e-6, atol=1e-7)
        assert float(ours2.state[x]["step"]) == 6.0


def test_adam_writes_the_bf16_shadow_and_records_into_a_graph():
    from obman_train_amd import ops
    from obman_train_amd.optim import ObmanAdam

    dev = torch.device("cuda", 0)
    p = torch.nn.Parameter(torch.randn(64, 32, 3, 3, device=dev).contiguous(memory_format=torch.channels_last))
    q = torch.nn.Parameter(torch.randn(1001, device=dev))
    p._obman_shadow = ops.bf16_shadow(p.detach())
    assert p._obman_shadow.stride() == p.stride()
    assert torch.equal(p._obman_shadow, p.detach().bfloat16())
    opt = ObmanAdam([p, q], lr=1e-2)
    gp, gq = torch.randn_like(p), torch.randn_like(q)
    p.grad, q.grad = gp, gq
    opt.step()  # eager: creates the state
    assert torch.equal(p._obman_shadow, p.detach().bfloat16())
    ref_p, ref_q = p.detach().clone(), q.detach().clone()
    ref = torch.optim.Adam([torch.nn.Parameter(ref_p.cpu()), torch.nn.Parameter(ref_q.cpu())], lr=1e-2)
    ref.load_state_dict(copy.deepcopy(opt.state_dict()))
    for grp in ref.param_groups:
        grp["capturable"] = False  # ObmanAdam's groups say capturable (device-side step counters); the host reference cannot be
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    for k in range(3):
        gp.normal_()
        gq.normal_()
        graph.replay()
        for rp, gg in zip(ref.param_groups[0]["params"], (gp, gq)):
            rp.grad = gg.cpu()
        ref.step()
    torch.cuda.synchronize()
    torch.testing.assert_close(p.detach().cpu(), ref.param_groups[0]["params"][0].detach(), rtol=3e-6, atol=1e-7)
    torch.testing.assert_close(q.detach().cpu(), ref.param_groups[0]["params"][1].detach(), rtol=3e-6, atol=1e-7)
    assert torch.equal(p._obman_shadow, p.detach().bfloat16())
    assert float(opt.state[p]["step"]) == 4.0


@pytest.mark.parametrize("B,N,with_scale", [(1, 1, True), (3, 642, True), (64, 16050, True), (5, 777, False)])
def test_affine_points(B, N, with_scale):
    from obman_train_amd import ops

    g = torch.Generator().manual_seed(B * 7 + N)
    verts = torch.randn(B, N, 3, generator=g) * 40
    scale = (torch.rand(B, 1, generator=g) + 0.5) if with_scale else None
    trans = torch.randn(B, 3, generator=g) * 100
    cot = torch.randn(B, N, 3, generator=g)
    hv, ht = verts.clone().requires_grad_(), trans.clone().requires_grad_()
    hs = scale.clone().requires_grad_() if with_scale else None
    want = (hs.unsqueeze(1) * hv if with_scale else hv) + ht.unsqueeze(1)
    (want * cot).sum().backward()
    dv, dt = verts.cuda().requires_grad_(), trans.cuda().requires_grad_()
    ds = scale.cuda().requires_grad_() if with_scale else None
    got = ops.affine_points(dv, ds, dt)
    (got * cot.cuda()).sum().backward()
    assert torch.equal(got.detach().cpu(), want.detach())  # two IEEE roundings on both sides
    torch.testing.assert_close(dv.grad.cpu(), hv.grad, rtol=1e-6, atol=0)
    torch.testing.assert_close(dt.grad.cpu(), ht.grad, rtol=2e-5, atol=2e-5 * float(ht.grad.abs().max()))
    if with_scale:
        assert ds.grad.shape == scale.shape
        torch.testing.assert_close(ds.grad.cpu(), hs.grad, rtol=2e-5, atol=2e-5 * float(hs.grad.abs().max()))
    # run to run identical (fixed-order partial sums)
    dv2, dt2 = verts.cuda().requires_grad_(), trans.cuda().requires_grad_()
    ds2 = scale.cuda().requires_grad_() if with_scale else None
    (ops.affine_points(dv2, ds2, dt2) * cot.cuda()).sum().backward()
    assert torch.equal(dt2.grad, dt.grad)


def test_mse_terms_match_mse_loss():
    from obman_train_amd import ops

    g = torch.Generator().manual_seed(11)
    shapes = [(64, 778, 3), (64, 21, 3), (64, 10), (64, 30), (64, 3), (64, 1), (1,), (7, 5)]
    preds = [torch.randn(s, generator=g) * 10 for s in shapes]
    targets = [torch.randn(s, generator=g) * 10 if i not in (2, 3) else None for i, s in enumerate(shapes)]
    lam = torch.rand(len(shapes), generator=g) + 0.1
    hp = [p.clone().requires_grad_() for p in preds]
    want = [F.mse_loss(p, torch.zeros_like(p) if t is None else t) for p, t in zip(hp, targets)]
    sum(l * w for l, w in zip(lam, want)).backward()
    dp = [p.cuda().requires_grad_() for p in preds]
    pose = torch.randn(64, 33, generator=g)  # a strided prediction (preds["pose"][:, 3:]): made contiguous inside
    dpose = pose.cuda().requires_grad_()
    got = ops.mse_terms([(p, None if t is None else t.cuda()) for p, t in zip(dp, targets)])
    extra = ops.mse_terms([(dpose[:, 3:], None)])[0]
    (sum(float(l) * w for l, w in zip(lam, got)) + extra).backward()
    for w, gt_ in zip(want, got):
        assert gt_.dim() == 0
        torch.testing.assert_close(gt_.detach().cpu(), w.detach(), rtol=2e-6, atol=0)
    for h, d in zip(hp, dp):
        torch.testing.assert_close(d.grad.cpu(), h.grad, rtol=2e-6, atol=1e-9)
    hpose = pose.clone().requires_grad_()
    F.mse_loss(hpose[:, 3:], torch.zeros(64, 30)).backward()
    torch.testing.assert_close(extra.detach().cpu(), F.mse_loss(pose[:, 3:], torch.zeros(64, 30)), rtol=2e-6, atol=0)
    torch.testing.assert_close(dpose.grad.cpu(), hpose.grad, rtol=2e-6, atol=1e-9)
    assert ops.mse_terms([]) == []
    with pytest.raises(ValueError):
        ops.mse_terms([(dp[0], dp[1])])


@pytest.mark.parametrize("B,N", [(1, 1), (4, 600), (64, 600), (3, 2049)])
def test_gt_object_stats(B, N):
    from obman_train_amd import ops

    gt = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(N)) * 50 + 200
    cen, centred, rad = ops.gt_object_stats(gt.cuda())
    want_c = gt.mean(1)
    want_d = gt - want_c.unsqueeze(1)
    want_r = torch.norm(want_d, 2, 2).max(1)[0].unsqueeze(1)
    assert rad.shape == (B, 1)
    torch.testing.assert_close(cen.cpu(), want_c, rtol=2e-6, atol=1e-4)
    torch.testing.assert_close(centred.cpu(), want_d, rtol=0, atol=2e-4)
    torch.testing.assert_close(rad.cpu(), want_r, rtol=3e-6, atol=2e-4)


def test_shadow_conv_equals_the_autocast_convolution_and_notices_foreign_writes():
    from obman_train_amd import ops
    from obman_train_amd.optim import attach_bf16_shadows

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(64, 128, 3, stride=2, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(8, 64, 32, 32, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    cot = torch.randn(8, 128, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        want = conv(x)
    (want.float() * cot).sum().backward()
    gw, gx = conv.weight.grad.clone(), x.grad.clone()
    conv.weight.grad, x.grad = None, None
    assert attach_bf16_shadows(conv) == 1
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got = ops.shadow_conv2d(conv, x)
    (got.float() * cot).sum().backward()
    assert got.dtype == torch.bfloat16 and conv.weight.grad.dtype == torch.float32
    # the same MIOpen problem with bit-identical operands; solver choice may differ between the two calls, so bf16-level bounds
    torch.testing.assert_close(got.float(), want.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(conv.weight.grad, gw, rtol=2e-2, atol=2e-2 * float(gw.abs().max()))
    torch.testing.assert_close(x.grad, gx, rtol=2e-2, atol=2e-2 * float(gx.abs().max()))
    # somebody else writes the filter (load_state_dict, another optimizer): the stale shadow must not be used
    with torch.no_grad():
        conv.weight.mul_(-3.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got2 = ops.shadow_conv2d(conv, x)
        want2 = conv(x)
    torch.testing.assert_close(got2.float(), want2.float(), rtol=2e-2, atol=2e-2)
    assert torch.equal(conv.weight._obman_shadow, conv.weight.detach().bfloat16())
    # outside autocast, or without a shadow: the module itself
    out = ops.shadow_conv2d(conv, x)
    assert out.dtype == torch.float32
