"""GPU parity: fused fp32-MFMA PointGenCon decoder (C-ABI obman_pointgen_fwd/bwd) vs the oracle restatement of
atlasbranch.py:117-132 + atlasutils.py:65-75 (materialised concat, conv1d, batch_norm).  fp32 MFMA is an exact fp32 fma
chain; differences come from summation order only: outputs rtol 2e-4 of the output scale, gradients 2e-3 of the largest entry."""
import numpy as np
import pytest
import torch

from oracle import atlas as oatlas
from obman_train_amd.icosphere import icosphere
from tests.golden.common import load_seeded

pytestmark = pytest.mark.gpu


def _decoder(c1, seed):
    from obman_train_amd.networks.branches.atlasutils import PointGenCon

    dec = load_seeded(PointGenCon(bottleneck_size=c1, out_factor=200), seed)
    with torch.no_grad():
        dec.conv4.weight.mul_(0.3)
    return dec


def _oracle(dec, feats, grid, training, mfma_round=None):
    params = {"decoder." + k: v.detach().clone() for k, v in list(dec.named_parameters()) + list(dec.named_buffers())}
    for k, v in params.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_()
    f = feats.clone().requires_grad_()
    B, N = f.shape[0], grid.shape[0]
    x = torch.cat((grid.t().unsqueeze(0).expand(B, -1, -1), f.unsqueeze(2).expand(-1, -1, N)), 1)
    out = oatlas.pointgen(params, x, training=training, out_factor=dec.out_factor, mfma_round=mfma_round).transpose(2, 1)
    return out, f, params


@pytest.mark.parametrize("c1,B,subdiv,training,patches", [(35, 3, 1, True, 1), (35, 2, 1, False, 1), (515, 4, 3, True, 1),
                                                          (515, 2, 2, False, 1), (131, 5, 2, True, 1), (515, 2, 1, True, 25),
                                                          # 259 -> 129 -> 64 channels: one / three leftover columns and one leftover row beyond whole
                                                          # 64 / 128 tiles in every GEMM (side products of the fp32 kernels, incl. the corner elements)
                                                          (259, 3, 1, True, 1), (259, 2, 2, False, 1),
                                                          # 4 050 template vertices: the segmented layer-1 finalize of large templates (N > 2048)
                                                          (35, 2, 2, True, 25), (131, 3, 2, False, 25),
                                                          # 68 850 rows = 538 row blocks: the pre-reduction of the per-block BatchNorm partials (> 512 blocks)
                                                          (35, 17, 2, True, 25)])
def test_decoder_forward_backward_matches_oracle(c1, B, subdiv, training, patches):
    from obman_train_amd import ops

    dec = _decoder(c1, 7)
    dec.train(training)
    from obman_train_amd.icosphere import multi_patch

    grid = torch.from_numpy(multi_patch(subdiv, patches)[0].astype(np.float32))  # patches=25: the configs[2] template layout
    # Seed note: a ReLU derivative is discontinuous at 0, so an input whose pre-activation lands within fp32 rounding of
    # zero flips one mask element between ANY two fp32 evaluation orders and moves one channel's gradient by ~1/sqrt(rows)
    # of its magnitude (seed 8 does this for the 25-patch case: fp64 and fp32 runs of the oracle itself disagree there).
    # Seeds 100.. are free of such ties for every case below (tools/decoder_err.py prints the per-parameter errors against
    # an fp64 run of the oracle: 1e-6 .. 1e-5), so the gradient tolerance is 2e-4 of each tensor's max.
    rng = np.random.RandomState(100)
    feats = torch.from_numpy(rng.normal(0, 1, size=(B, c1 - 3)).astype(np.float32))
    cot = torch.from_numpy(rng.normal(0, 1, size=(B, grid.shape[0], 3)).astype(np.float32))
    want, f_o, params = _oracle(dec, feats, grid, training)
    (want * cot).sum().backward()

    dec_g = _decoder(c1, 7).cuda()
    dec_g.train(training)
    f_g = feats.cuda().requires_grad_()
    got = ops.pointgen_decode(dec_g, f_g, grid.cuda())
    (got * cot.cuda()).sum().backward()
    scale = want.abs().max().item()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-4, atol=2e-4 * scale)

    def check(name, g, w):
        err = (g.cpu() - w).abs().max().item()
        ref = w.abs().max().item()
        assert err <= 2e-4 * ref + 1e-5, (name, err, ref)

    check("features", f_g.grad, f_o.grad)
    for name, prm in dec_g.named_parameters():
        w = params["decoder." + name].grad
        if name.startswith("conv") and name.endswith("bias") and training and name != "conv4.bias":
            # bias before a train-mode BatchNorm has an exactly-zero gradient; autograd returns round-off noise
            assert prm.grad.abs().max().item() <= 1e-3 * (dec_g.conv4.weight.grad.abs().max().item())
            continue
        check(name, prm.grad, w.reshape(prm.grad.shape))
    if training:  # running statistics follow torch's momentum update (unbiased variance)
        for k in (1, 2, 3):
            bn = getattr(dec_g, "bn%d" % k)
            np.testing.assert_allclose(bn.running_mean.cpu().numpy(), params["decoder.bn%d.running_mean" % k].numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(bn.running_var.cpu().numpy(), params["decoder.bn%d.running_var" % k].numpy(), rtol=1e-4, atol=1e-5)
            assert int(bn.num_batches_tracked) == 1


def test_decoder_matches_reference_golden_pointgen(golden):
    """`PointGenCon.forward(x)`, the reference's generic entry point (atlasutils.py:65-75).  (1) The golden input of
    tests/golden/pointgen.npz is an arbitrary x (its feature rows vary along N, it cannot be factorised): the mirror runs the
    stock-op path, says so once, and matches the reference's output.  (2) The tensor the reference actually passes - the
    [grid ; broadcast feature] concatenation of atlasbranch.py:117-132 - is recognised and routed to the fused HIP decoder:
    same result as decode(), as the stock ops on the same x, and the same gradient for the feature leaf."""
    import torch.nn.functional as F

    from obman_train_amd.networks.branches.atlasutils import PointGenCon

    g = golden("pointgen")
    dec = load_seeded(PointGenCon(bottleneck_size=35, out_factor=200), int(g["seed"])).cuda().train()
    PointGenCon._warned_generic = False
    with pytest.warns(UserWarning, match="generic stock-op path"):
        y = dec(torch.from_numpy(g["x"]).cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y_train"], rtol=1e-4, atol=1e-3)

    mk = lambda: load_seeded(PointGenCon(bottleneck_size=35, out_factor=200), int(g["seed"])).cuda().train()  # noqa: E731
    grid = torch.from_numpy(icosphere(1)[0].astype(np.float32)).cuda()
    feats = torch.randn(3, 32, device="cuda")
    cot = torch.randn(3, 3, 42, device="cuda")
    outs = {}
    for how in ("forward", "decode", "stock"):
        dec_k, f = mk(), feats.clone().requires_grad_()
        x = torch.cat((grid.t().unsqueeze(0).expand(3, -1, -1), f.unsqueeze(2).expand(-1, -1, 42)), 1)
        if how == "forward":
            import warnings

            with warnings.catch_warnings():
                warnings.filterwarnings("error", message=".*generic stock-op path.*")  # the concatenation must NOT take the generic path
                out = dec_k(x)
        elif how == "decode":
            out = dec_k.decode(f, grid).transpose(1, 2)
        else:
            out = dec_k._tail(F.relu(dec_k.bn1(dec_k.conv1(x))))
        (out * cot).sum().backward()
        outs[how] = (out.detach().cpu().numpy(), f.grad.cpu().numpy(), dec_k.conv2.weight.grad.cpu().numpy())
    for other in ("decode", "stock"):
        for a, b in zip(outs["forward"], outs[other]):
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4 * np.abs(b).max())
    # (3) point rows that NEED a gradient (a learned / refined point set, or a leaf x): the fused decoder has no gradient for
    # them, so forward() must keep the stock ops - silently dropping that gradient was ADVICE r03's finding
    grads = {}
    for how in ("forward", "stock"):
        dec_k, f, gl = mk(), feats.clone().requires_grad_(), grid.clone().requires_grad_()
        x = torch.cat((gl.t().unsqueeze(0).expand(3, -1, -1), f.unsqueeze(2).expand(-1, -1, 42)), 1)
        out = dec_k(x) if how == "forward" else dec_k._tail(F.relu(dec_k.bn1(dec_k.conv1(x))))
        (out * cot).sum().backward()
        assert gl.grad is not None and float(gl.grad.abs().max()) > 0
        grads[how] = (gl.grad.cpu().numpy(), f.grad.cpu().numpy())
    for a, b in zip(grads["forward"], grads["stock"]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6 * np.abs(b).max())
    leaf = torch.cat((grid.t().unsqueeze(0).expand(3, -1, -1), feats.unsqueeze(2).expand(-1, -1, 42)), 1).clone().requires_grad_()
    (mk()(leaf) * cot).sum().backward()
    assert float(leaf.grad[:, :3].abs().max()) > 0 and float(leaf.grad[:, 3:, 1:].abs().max()) > 0  # per-point gradients, not column 0 only


def test_decoder_multi_patch_and_determinism():
    from obman_train_amd import ops
    from obman_train_amd.icosphere import multi_patch

    dec = _decoder(515, 9).cuda().train()
    grid = torch.from_numpy(multi_patch(2, 3)[0].astype(np.float32)).cuda()  # 3 patches x 162 vertices
    feats = torch.randn(6, 512, device="cuda").requires_grad_()
    outs = []
    for _ in range(2):
        dec.zero_grad()
        feats.grad = None
        out = ops.pointgen_decode(dec, feats, grid)
        out.square().mean().backward()
        outs.append((out.detach().clone(), feats.grad.clone(), dec.conv2.weight.grad.clone()))
    assert outs[0][0].shape == (6, 486, 3)
    # BN running stats moved between the two calls but batch statistics are used in train mode: identical results
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("c1,B,subdiv,training,patches", [(35, 3, 1, False, 1), (515, 4, 3, True, 1), (515, 2, 2, False, 1),
                                                          (131, 5, 2, False, 1), (515, 8, 1, True, 25),
                                                          # a narrow decoder in train mode (weight gradients of <= 128 columns: WN = 2 tiles of
                                                          # decoder_tn2.h since round 6).  (c1 = 35 in train mode - 8 / 17 channels, 126 rows -
                                                          # sits at 0.08 - 0.13 on four tensors: BatchNorm statistics of 126 bf16 rows; not a case)
                                                          (131, 4, 1, True, 1)])
def test_decoder_bf16_mfma_flavour(c1, B, subdiv, training, patches):
    """mfma_dtype="bf16" (BASELINE configs[2]): layer-2/3 operands rounded to bf16, fp32 accumulation and statistics.

    Forward: against the oracle with the same roundings - contraction operands AND the stored layer outputs h2 / h3
    (oracle/atlas.py:pointgen mfma_round) - summation order differs, so values that sit on a bf16 rounding boundary land on
    either side (one step = 2^-8 of that activation): 1e-2 of the output scale (measured 2e-3 .. 6e-3).
    Backward rounds the gradient operands too, which autograd of that model does not do, so gradients are compared with the
    fp32 oracle's in relative L2 norm.  The cotangent is positive, so the gradient sums are coherent: with a random-sign
    cotangent every BatchNorm gradient is a noise-like sum and the ~0.4 % of ReLU masks that bf16 noise flips show up as
    10-30 % relative error of a quantity that is itself noise.  Eval-mode BatchNorm: 3e-2 (measured <= 1e-2); train mode
    subtracts the batch means again (cancellation): 8e-2 (measured 2.7e-2)."""
    from obman_train_amd import ops
    from obman_train_amd.icosphere import multi_patch

    dec = _decoder(c1, 7)
    dec.train(training)
    grid = torch.from_numpy(multi_patch(subdiv, patches)[0].astype(np.float32))
    rng = np.random.RandomState(100)
    feats = torch.from_numpy(rng.normal(0, 1, size=(B, c1 - 3)).astype(np.float32))
    cot = torch.from_numpy((np.abs(rng.normal(0, 1, size=(B, grid.shape[0], 3))) + 0.5).astype(np.float32))
    want_bf, _, _ = _oracle(dec, feats, grid, training, mfma_round=lambda t: t.bfloat16().float())
    want32, f_o, params = _oracle(dec, feats, grid, training)
    (want32 * cot).sum().backward()

    dec_g = _decoder(c1, 7).cuda()
    dec_g.train(training)
    dec_g.mfma_dtype = "bf16"
    f_g = feats.cuda().requires_grad_()
    got = ops.pointgen_decode(dec_g, f_g, grid.cuda())
    (got * cot.cuda()).sum().backward()
    scale = want_bf.abs().max().item()
    err = (got.detach().cpu() - want_bf.detach()).abs().max().item()
    assert err <= 1e-2 * scale, (err, scale)
    assert (got.detach().cpu() - want32.detach()).abs().max().item() <= 5e-2 * scale  # and it is a bf16-accurate decoder

    def rel_l2(g, w):
        return ((g.cpu().double() - w.double()).norm() / w.double().norm().clamp_min(1e-30)).item()

    worst = {"features": rel_l2(f_g.grad, f_o.grad)}
    for name, prm in dec_g.named_parameters():
        if name.startswith("conv") and name.endswith("bias") and training and name != "conv4.bias":
            continue
        worst[name] = rel_l2(prm.grad, params["decoder." + name].grad.reshape(prm.grad.shape))
    tol = 8e-2 if training else 3e-2
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, (bad, worst)
    with pytest.raises(ValueError):
        ops.pointgen_decode(dec_g, f_g, grid.cuda(), mfma_dtype="fp8")
