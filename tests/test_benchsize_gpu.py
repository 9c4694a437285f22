"""GPU: parity AT THE SIZES `bench.py` RUNS (VERDICT r03 "missing" item 3).

* configs[1] - the driver-timed workload itself: whole model, bs 64, 256 x 256, fp32, forward + backward against
  `oracle.handnet_forward` on the same weights and batch (one host step of the materialised formulation, ~8-60 s on the test
  box's cores).  Bounds are frozen at small multiples of what MI355X produced (recorded by the test: profiles/r04_parity_measured.md).
* configs[4] at bs 64: the decoder over R = 64 x 64 050 = 4.1 M rows - its bf16 operands are 2.2 GB (> 2^31 bytes: the 32-bit
  buffer-offset class of bug of commit d6ef277) and its fp32 gy1 is 8.7 GB - in eval-mode BatchNorm, where samples are
  independent: the batch holds 32 copies of the two samples of the oracle-pinned B = 2 run (tests/test_fullsize_gpu.py), so
  every slice of the output must reproduce that run and the weight gradients must be 32 x its weight gradients.
* `compute_contact_loss` at configs[4] size and B = 64 the same way (64 050 object vertices, 128 000 faces in 25 patches).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from obman_train_amd.contactzones import hand_template, load_contacts
from obman_train_amd.icosphere import multi_patch
from tests.golden.common import synth_hand_object
from tests.test_decoder_gpu import _decoder

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_configs1_model_bs64_256_matches_cpu_oracle():
    import warnings

    from oracle import handnet as ohandnet
    from oracle import mano as omano
    from obman_train_amd.mano_params import synthetic_mano
    from obman_train_amd.networks.bases import resnet
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.queries import BaseQueries, TransQueries
    from obman_train_amd.synthetic import CONFIGS, make_batch
    from tests.conftest import record_measurement

    warnings.simplefilter("ignore")
    cfg = dict(CONFIGS["c2"])
    B, res = 64, 256
    torch.manual_seed(0)
    model = HandNet(**cfg).train()
    named = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in named.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_()
    sample = make_batch(B, "cpu", seed=0, image_size=res)  # the batch bench.py times (seed = rank 0)
    sample[BaseQueries.sides] = ["left", "right"] * (B // 2)
    keys = SimpleNamespace(images=TransQueries.images, verts3d=TransQueries.verts3d, joints3d=TransQueries.joints3d,
                           objpoints3d=TransQueries.objpoints3d, sides=BaseQueries.sides)
    packs = {s: omano.pack_to_torch(synthetic_mano(s)) for s in ("right", "left")}
    o_total, o_res, o_losses = ohandnet.handnet_forward(
        named, cfg, dict(sample), keys, packs, model.atlas_branch.test_verts.clone(), model.atlas_branch.test_faces,
        zones=load_contacts()[1], resnet_shell=resnet.resnet18(), training=True)
    o_total.backward()
    model.cuda()
    total, out, losses = model.forward(sample)
    total.backward()

    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)  # noqa: E731
    m = {"total": rel(total, o_total),
         "worst_loss": max(rel(losses[k], v) for k, v in o_losses.items() if v is not None and abs(float(v)) > 1e-5)}
    for k in ("verts", "joints", "objpoints3d"):
        w = o_res[k].detach()
        m[k + "_of_scale"] = float((out[k].detach().cpu() - w).abs().max() / w.abs().max())
    grads, grads_l2 = {}, {}
    got_params = dict(model.named_parameters())
    for name in ("mano_branch.pose_reg.weight", "mano_branch.shape_reg.0.weight", "atlas_branch.decoder.conv1.weight",
                 "atlas_branch.decoder.conv2.weight", "atlas_branch.decoder.conv4.weight", "base_net.layer4.1.conv2.weight",
                 "base_net.layer2.0.conv1.weight", "base_net.conv1.weight", "base_net.bn1.weight"):
        if name not in got_params or named[name].grad is None:
            continue
        g, w = got_params[name].grad.cpu().numpy(), named[name].grad.numpy()
        grads[name] = float(np.abs(g - w).max() / np.abs(w).max())
        grads_l2[name] = float(np.linalg.norm((g - w).astype(np.float64)) / np.linalg.norm(w.astype(np.float64)))
    m["worst_grad_of_max"] = max(grads.values())
    m["worst_grad_l2"] = max(grads_l2.values())
    m["grads"] = grads
    m["grads_l2"] = grads_l2
    record_measurement("configs1_bs64_256_vs_oracle", m)
    assert len(grads) >= 7, grads
    # frozen at ~3x the values measured on MI355X (profiles/r04_parity_measured.md: total 1.3e-6, worst loss term 1.9e-6, object
    # points 1.3e-5 of scale): north_star's 1e-4 on loss scalars and outputs holds with an order of magnitude to spare
    assert m["total"] <= 1e-5, m
    for k, v in o_losses.items():
        if v is None:
            assert losses[k] is None, k
        else:
            np.testing.assert_allclose(float(losses[k]), float(v), rtol=1e-4, atol=1e-5, err_msg=k)
    assert m["verts_of_scale"] <= 1e-4 and m["joints_of_scale"] <= 1e-4 and m["objpoints3d_of_scale"] <= 1e-4, m
    # gradients: measured worst 2.4e-2 of the largest entry (decoder.conv2.weight: single entries move when a Chamfer arg-min or a
    # ReLU mask flips on a 1e-5 difference of the encoder features), 7.2e-3 in relative L2 (base_net.bn1.weight); the decoder
    # by itself at this size is within 1e-4 L2 (test_configs1_decoder_bs64_matches_oracle)
    assert m["worst_grad_of_max"] <= 6e-2 and m["worst_grad_l2"] <= 2e-2, m


def test_configs1_decoder_bs64_matches_oracle():
    """The decoder call of the driver-timed step by itself: c1 = 515, 64 samples x 642 points (R = 41 088 rows: 321 row blocks,
    the split-K chunking and the side products at their bench geometry), train-mode BatchNorm, against the oracle."""
    from obman_train_amd import ops
    from tests.conftest import record_measurement
    from tests.test_decoder_gpu import _oracle

    c1, B = 515, 64
    grid = T(multi_patch(3, 1)[0].astype(np.float32))
    rng = np.random.RandomState(100)
    feats = T(rng.normal(0, 1, size=(B, c1 - 3)).astype(np.float32))
    cot = T((np.abs(rng.normal(0, 1, size=(B, grid.shape[0], 3))) + 0.5).astype(np.float32))
    dec = _decoder(c1, 7).train()
    want, f_o, params = _oracle(dec, feats, grid, True)
    (want * cot).sum().backward()
    dec_g = _decoder(c1, 7).cuda().train()
    f_g = feats.cuda().requires_grad_()
    got = ops.pointgen_decode(dec_g, f_g, grid.cuda())
    (got * cot.cuda()).sum().backward()
    scale = want.abs().max().item()
    out_err = float((got.detach().cpu() - want.detach()).abs().max()) / scale
    l2 = lambda g, w: ((g.detach().cpu().double() - w.double()).norm() / w.double().norm().clamp_min(1e-30)).item()  # noqa: E731
    mx = lambda g, w: float((g.detach().cpu() - w).abs().max() / w.abs().max().clamp_min(1e-30))  # noqa: E731
    m = {"out_of_scale": out_err, "features_l2": l2(f_g.grad, f_o.grad), "features_max": mx(f_g.grad, f_o.grad)}
    for name, prm in dec_g.named_parameters():
        if name.startswith("conv") and name.endswith("bias") and name != "conv4.bias":
            continue
        w = params["decoder." + name].grad.reshape(prm.grad.shape)
        m[name + "_l2"], m[name + "_max"] = l2(prm.grad, w), mx(prm.grad, w)
    record_measurement("configs1_decoder_bs64_vs_oracle", m)
    assert out_err <= 2e-4, m
    bad = {k: v for k, v in m.items() if k.endswith("_l2") and not v <= 1e-3}
    assert not bad, (bad, m)


@pytest.mark.parametrize("flavour", ["f32", "bf16"])
def test_configs4_decoder_bs64_reproduces_the_pinned_b2_run(flavour):
    from obman_train_amd import ops

    c1, B, copies = 515, 64, 32
    grid = T(multi_patch(4, 25)[0].astype(np.float32)).cuda()
    assert grid.shape[0] == 64050
    rng = np.random.RandomState(100)  # the inputs of tests/test_fullsize_gpu.py::test_decoder_full_size (oracle-pinned there)
    feats2 = T(rng.normal(0, 1, size=(2, c1 - 3)).astype(np.float32))
    cot2 = T((np.abs(rng.normal(0, 1, size=(2, grid.shape[0], 3))) + 0.5).astype(np.float32))

    def run(feats, cot):
        dec = _decoder(c1, 7).cuda().eval()  # eval-mode BatchNorm: every sample is independent of the rest of the batch
        dec.mfma_dtype = flavour
        f = feats.cuda().requires_grad_()
        out = ops.pointgen_decode(dec, f, grid)
        (out * cot.cuda()).sum().backward()
        torch.cuda.synchronize()
        return out.detach(), f.grad.detach(), {n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None}

    out2, gf2, gw2 = run(feats2, cot2)
    if flavour == "bf16":  # the operand that crosses 2^31 bytes: gy2 / h2 as bf16 [R, 272]
        assert B * 64050 * 272 * 2 > 2 ** 31
    out, gf, gw = run(feats2.repeat(copies, 1), cot2.repeat(copies, 1, 1))
    assert out.shape == (B, 64050, 3) and bool(torch.isfinite(out).all())
    scale = float(out2.abs().max())
    for b in (0, 1, 2, 31, 33, 62, 63):  # first, last and middle rows of the 4.1 M-row problem
        err = float((out[b] - out2[b % 2]).abs().max())
        assert err <= 1e-6 * scale, (b, err, scale)  # same per-row arithmetic whatever the batch: equal up to nothing
    tol = 1e-4 if flavour == "f32" else 2e-3
    for b in (0, 1, 30, 63):
        err = float((gf[b] - gf2[b % 2]).abs().max() / gf2[b % 2].abs().max())
        assert err <= tol, ("feature gradient", b, err)
    for name, g in gw.items():  # weight gradients sum over the batch: 32 x the pinned run's
        ref = gw2[name] * copies
        if float(ref.abs().max()) == 0.0:
            continue
        err = float((g - ref).abs().max() / ref.abs().max())
        assert err <= (5e-4 if flavour == "f32" else 1e-2), (name, err)


def test_configs4_contact_loss_bs64_reproduces_the_pinned_run():
    from obman_train_amd.networks.branches.contactloss import compute_contact_loss
    from tests.test_fullsize_gpu import _CONTACT_SIZES, _c3_scene

    patches, copies = 25, 64
    B1, seed, n_obj, n_faces = _CONTACT_SIZES[4]  # the oracle-pinned scene of test_contact_loss_full_size (B = 1)
    hand1, obj1, faces = _c3_scene(B1, seed, 4)
    assert obj1.shape[1] == n_obj and faces.shape[0] == n_faces
    kw = dict(contact_thresh=10, contact_mode="dist_tanh", collision_thresh=20, collision_mode="dist_tanh", contact_target="all",
              contact_zones="zones")

    def run(hand, obj):
        h, o = hand.cuda().requires_grad_(), obj.cuda().requires_grad_()
        missed, pen, info, metrics = compute_contact_loss(h, None, o, faces, obj_patches=patches, **kw)
        (missed + 2.0 * pen).sum().backward()
        torch.cuda.synchronize()
        return missed.detach(), pen.detach(), info, metrics, h.grad, o.grad

    m1, p1, info1, met1, gh1, go1 = run(hand1, obj1)
    m, p, info, met, gh, go = run(hand1.repeat(copies, 1, 1), obj1.repeat(copies, 1, 1))
    # batch-global masked means over 64 identical samples = the single sample's
    np.testing.assert_allclose(float(m), float(m1), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(float(p), float(p1), rtol=1e-5, atol=1e-7)
    for b in (0, 1, 31, 32, 63):
        assert torch.equal(info["repulsion_masks"][b], info1["repulsion_masks"][0]), b
        assert torch.equal(info["attraction_masks"][b] != 0, info1["attraction_masks"][0] != 0), b
        assert torch.equal(info["min_dists"][b], info1["min_dists"][0]), b
        # gradients of a mean over 64 x as many elements: 1/64 of the single-sample gradient
        for g, g1, name in ((gh, gh1, "hand"), (go, go1, "obj")):
            if g1 is None or float(g1.abs().max()) == 0.0:
                continue
            err = float((g[b] * copies - g1[0]).abs().max() / g1[0].abs().max())
            assert err <= 1e-4, (name, b, err)
    for k in ("max_penetr", "mean_penetr"):
        np.testing.assert_allclose(float(met[k]), float(met1[k]), rtol=1e-5, atol=1e-7)
