"""SURVEY §8 a8 - ``AtlasBranch.forward`` (random points on the sphere, one set per sample; atlasbranch.py:78-108).

``tests/golden/atlas_random.npz`` holds outputs of the REFERENCE's own ``AtlasBranch.forward`` run on injected normal draws
(``tests/golden/make_golden_atlas_random.py``).  CPU: the oracle restatement and the product's host logic (test-only oracle
backend) against it.  GPU: the fused decoder with a per-sample grid (``obman_pointgen_fwd/bwd``, ``grid_per_sample``) against
the golden (1e-4 of the output scale, gradients 1e-3 of the largest entry) and against the oracle at the real size
(c1 = 515, 600 points, B = 4)."""
import numpy as np
import pytest
import torch

from oracle import atlas as oatlas
from tests.golden.common import load_seeded

T = torch.from_numpy
PROBES = ("decoder.conv1.weight", "decoder.bn1.weight", "decoder.bn1.bias", "decoder.conv2.weight", "decoder.conv4.bias")
STATS = ("decoder.bn1.running_mean", "decoder.bn1.running_var", "decoder.bn3.running_var")


def _branch(g, trans):
    from obman_train_amd.networks.branches.atlasbranch import AtlasBranch

    return load_seeded(AtlasBranch(points_nb=int(g["points_nb"]), bottleneck_size=g["feats"].shape[1], predict_trans=trans,
                                   inference_ico_divisions=1, out_factor=200), int(g["seed"]))


def _check(g, tag, res, feats, br, tol_o, tol_g):
    def close(got, want, tol, what):
        got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
        err = np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30)
        assert err <= tol, "%s%s: error %.3g of the largest entry (tolerance %.1g)" % (tag, what, err, tol)

    for key in ("objpoints3d", "objtrans", "objpointscentered3d"):
        if tag + key in g.files:
            close(res[key], g[tag + key], tol_o, key)
        else:
            assert key not in res
    close(feats.grad, g[tag + "grad_feats"], tol_g, "grad_feats")
    params = dict(br.named_parameters())
    for name in PROBES:
        if name == "decoder.conv4.bias" or not tag.endswith("train_") or not name.endswith("conv1.bias"):
            close(params[name].grad, g[tag + "g:" + name], tol_g, name)
    sd = br.state_dict()
    for name in STATS:
        close(sd[name], g[tag + "s:" + name], tol_o, name)


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("trans", [False, True])
def test_oracle_matches_reference_golden(golden, trans, mode):
    g = golden("atlas_random")
    br = _branch(g, trans).train(mode == "train")
    params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k)
              for k, v in br.state_dict().items()}
    feats = T(g["feats"]).requires_grad_()
    res = oatlas.forward_random(params, feats, T(g["rand_grid"]), predict_trans=trans, training=(mode == "train"))
    (res["objpoints3d"] * T(g["cot"])).sum().backward()
    tag = "t%d_%s_" % (int(trans), mode)
    np.testing.assert_allclose(res["objpoints3d"].detach().numpy(), g[tag + "objpoints3d"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(feats.grad.numpy(), g[tag + "grad_feats"], rtol=1e-3, atol=1e-4 * np.abs(g[tag + "grad_feats"]).max())
    np.testing.assert_allclose(params["decoder.conv2.weight"].grad.numpy(), g[tag + "g:decoder.conv2.weight"], rtol=1e-3,
                               atol=1e-4 * np.abs(g[tag + "g:decoder.conv2.weight"]).max())


@pytest.mark.parametrize("trans", [False, True])
def test_host_logic_matches_reference_golden(golden, monkeypatch, trans):
    from tests import fake_ops

    fake_ops.install(monkeypatch)
    g = golden("atlas_random")
    br = _branch(g, trans).train()
    feats = T(g["feats"]).requires_grad_()
    res = br(feats, rand_grid=T(g["rand_grid"]))
    (res["objpoints3d"] * T(g["cot"])).sum().backward()
    assert "objfaces" not in res and "objscale" not in res
    _check(g, "t%d_train_" % int(trans), res, feats, br, tol_o=1e-4, tol_g=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("trans", [False, True])
def test_hip_per_sample_grid_matches_reference_golden(golden, trans, mode):
    g = golden("atlas_random")
    br = _branch(g, trans).cuda().train(mode == "train")
    feats = T(g["feats"]).cuda().requires_grad_()
    res = br(feats, rand_grid=T(g["rand_grid"]).cuda())
    (res["objpoints3d"] * T(g["cot"]).cuda()).sum().backward()
    _check(g, "t%d_%s_" % (int(trans), mode), res, feats, br, tol_o=1e-4, tol_g=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
def test_hip_per_sample_grid_full_width_vs_oracle(training):
    """c1 = 515 (ResNet-18 features), 600 points (atlas_points_nb default), B = 4: fused kernels vs the oracle."""
    from obman_train_amd.networks.branches.atlasbranch import AtlasBranch

    rng = np.random.RandomState(5)
    B, P = 4, 600
    br = load_seeded(AtlasBranch(points_nb=P, bottleneck_size=512, predict_trans=True, out_factor=200), 9)
    with torch.no_grad():
        br.decoder.conv4.weight.mul_(0.3)
    br.train(training)
    feats0 = T(rng.normal(0, 1, size=(B, 512)).astype(np.float32))
    draws = T(rng.normal(0, 1, size=(B, 3, P)).astype(np.float32))
    cot = T((np.abs(rng.normal(0, 1, size=(B, P, 3))) + 0.5).astype(np.float32))
    params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k)
              for k, v in br.state_dict().items()}
    f_o = feats0.clone().requires_grad_()
    want = oatlas.forward_random(params, f_o, draws, predict_trans=True, training=training)
    (want["objpoints3d"] * cot).sum().backward()
    br.cuda()
    f_g = feats0.cuda().requires_grad_()
    got = br(f_g, rand_grid=draws.cuda())
    (got["objpoints3d"] * cot.cuda()).sum().backward()
    scale = want["objpoints3d"].abs().max().item()
    assert (got["objpoints3d"].detach().cpu() - want["objpoints3d"].detach()).abs().max().item() <= 2e-4 * scale

    def rel(gv, wv):
        return ((gv.detach().cpu().double() - wv.double()).norm() / wv.double().norm().clamp_min(1e-30)).item()

    worst = {"features": rel(f_g.grad, f_o.grad)}
    for name, prm in br.named_parameters():
        if name.startswith("decoder.conv") and name.endswith("bias") and training and "conv4" not in name:
            continue
        worst[name] = rel(prm.grad, params[name].grad)
    bad = {k: v for k, v in worst.items() if not v <= 1e-3}
    assert not bad, (bad, worst)
    # without the hook the points are random: different every call, still on the decoder's manifold
    a, b = br(f_g)["objpoints3d"], br(f_g)["objpoints3d"]
    assert a.shape == (B, P, 3) and not torch.equal(a, b)
