"""GPU: the real HandNet (HIP kernels through the C-ABI) against (1) the golden vectors produced by
the reference's own HandNet.forward (TinyEncoder backbone, tolerance 1e-4 rel on losses/outputs as
north_star states) and (2) the CPU oracle on the real ResNet18 configuration of BASELINE configs 1/2."""
import numpy as np
import pytest
import torch

from tests.handnet_common import assert_matches_fixture, build_fixture_model, fixture_sample

pytestmark = pytest.mark.gpu

# real-ResNet18 whole-model bounds against the CPU oracle (see test_handnet_resnet18_matches_cpu_oracle)
# Measured on MI355X (profiles/r03_parity_measured.md; bs 4, 64 x 64, three configurations): total 1.4e-6, worst loss term 7.8e-5
# (3.0e-4 in the 25-patch model of tests/test_fullsize_gpu.py), vertices 5e-7 / object points 2.5e-5 of their scale, worst
# sampled gradient 4.1e-3 of its largest entry (layer4 under a 16-value BatchNorm).  Bounds = 2.5-7x those, because MIOpen's
# find picks the convolution solutions per process and its weight-gradient kernels accumulate with atomics.
TOL_TOTAL, TOL_LOSS, TOL_POINTS, TOL_GRAD = 1e-5, 8e-4, 1e-4, 1e-2


@pytest.mark.parametrize("tag", ["train", "eval"])
def test_handnet_matches_reference_golden(golden, monkeypatch, tag):
    g = golden("handnet_" + tag)
    model, _ = build_fixture_model(g, monkeypatch, train_mode=(tag == "train"))
    model.cuda()
    total, results, losses = model.forward(fixture_sample(g))  # CPU sample: HandNet moves it
    total.backward()
    assert total.is_cuda
    assert_matches_fixture(g, total, results, losses, model)


@pytest.mark.parametrize("contact,patches", [(False, 1), (True, 1), (True, 3)])
def test_handnet_resnet18_matches_cpu_oracle(contact, patches):
    """Config 1/2 model (ResNet18 + MANO + 1-sphere AtlasNet + Chamfer [+ contact]) at bs 4, 64x64 images:
    GPU product vs oracle.handnet_forward on the same weights.  Tolerance 1e-3 on loss scalars (MIOpen vs
    oneDNN convolutions differ at 1e-5..1e-4 after 18 layers); vertices 1e-3 rel of the hand scale."""
    from types import SimpleNamespace

    from oracle import handnet as ohandnet
    from oracle import mano as omano
    from obman_train_amd.contactzones import hand_template, load_contacts
    from obman_train_amd.mano_params import synthetic_mano
    from obman_train_amd.networks.bases import resnet
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.queries import BaseQueries, TransQueries
    from tests.golden.common import synth_hand_object

    cfg = dict(resnet_version=18, atlas_mesh=True, mano_use_pca=True, mano_comps=30, mano_neurons=[1024, 256],
               atlas_lambda=0.167, atlas_final_lambda=0.167, atlas_predict_trans=True, atlas_predict_scale=True,
               atlas_trans_weight=0.167, atlas_scale_weight=0.167, mano_lambda_verts=0.167, mano_lambda_joints3d=0.167,
               mano_use_shape=True, mano_lambda_shape=0.167, mano_lambda_pose_reg=0.167, mano_center_idx=0)
    if patches > 1:  # configs[2]/[4] extension: P sphere patches (union of closed spheres), smaller spheres to keep the oracle fast
        cfg.update(atlas_patches=patches, atlas_ico_divisions=2)
    if contact:
        cfg.update(contact_lambda=1.0, collision_lambda=1.0, contact_zones="zones", contact_mode="dist_tanh",
                   collision_mode="dist_tanh", contact_thresh=10, collision_thresh=20)
    torch.manual_seed(0)
    model = HandNet(**cfg)
    with torch.no_grad():
        model.atlas_branch.decoder.conv4.weight.mul_(0.2)
    model.train()
    named = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in named.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_()
    B = 4
    gtv, gtj, gto = synth_hand_object(B, 600, 5, hand_template()[0])
    images = torch.rand(B, 3, 64, 64) - 0.5
    keys = SimpleNamespace(images=TransQueries.images, verts3d=TransQueries.verts3d, joints3d=TransQueries.joints3d,
                           objpoints3d=TransQueries.objpoints3d, sides=BaseQueries.sides)
    sample = {TransQueries.images: images, TransQueries.verts3d: gtv, TransQueries.joints3d: gtj,
              TransQueries.objpoints3d: gto, BaseQueries.sides: ["left"] * B, "root": "wrist"}
    packs = {s: omano.pack_to_torch(synthetic_mano(s)) for s in ("right", "left")}
    ocfg = {k: v for k, v in cfg.items() if k not in ("atlas_patches", "atlas_ico_divisions")}
    o_total, o_res, o_losses = ohandnet.handnet_forward(
        named, ocfg, dict(sample), keys, packs, model.atlas_branch.test_verts.clone(), model.atlas_branch.test_faces,
        zones=load_contacts()[1], resnet_shell=resnet.resnet18(), training=True)
    o_total.backward()
    model.cuda()
    total, res, losses = model.forward(sample)
    total.backward()
    from tests.conftest import record_measurement

    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)  # noqa: E731
    measured = {"total": rel(total, o_total)}
    measured["worst_loss"] = max(rel(losses[k], v) for k, v in o_losses.items() if v is not None and abs(float(v)) > 1e-5)
    for k in ("verts", "objpoints3d"):
        w = o_res[k].detach()
        measured[k + "_of_scale"] = float((res[k].detach().cpu() - w).abs().max() / w.abs().max())
    grads = {}
    for name in ("mano_branch.pose_reg.weight", "atlas_branch.decoder.conv2.weight", "base_net.layer4.1.conv2.weight"):
        got = dict(model.named_parameters())[name].grad.cpu().numpy()
        want = named[name].grad.numpy()
        grads[name] = float(np.abs(got - want).max() / np.abs(want).max())
    measured["worst_grad_of_max"] = max(grads.values())
    record_measurement("handnet_resnet18_vs_oracle[contact=%s,patches=%d]" % (contact, patches), measured)
    # Bounds = ~2x the errors measured on MI355X (profiles/r03_parity_measured.md), not a guess: MIOpen's fp32 implicit-GEMM
    # convolutions and oneDNN's differ in summation order, 18 layers deep (north_star's 1e-4 holds where the encoder is the
    # same on both sides: tests/test_fullsize_gpu.py with injected features, the golden-vector tests above).
    assert measured["total"] <= TOL_TOTAL, measured
    for k, v in o_losses.items():
        if v is None:
            assert losses[k] is None
        else:
            np.testing.assert_allclose(float(losses[k]), float(v), rtol=TOL_LOSS, atol=1e-5, err_msg=k)
    assert measured["verts_of_scale"] <= TOL_POINTS and measured["objpoints3d_of_scale"] <= TOL_POINTS, measured
    assert measured["worst_grad_of_max"] <= TOL_GRAD, (grads, measured)


# configs[2] in its stated precision: bounds = ~2x what MI355X produced (profiles/r03_parity_measured.md), frozen.
FLAVOUR_BOUNDS = {
    # smooth loss terms rel. | total rel. | objpoints3d: max |diff| and rms diff over the point scale | repulsion-mask Hamming fraction |
    # five-Adam-step loss trajectory rel.   Measured (bs 16, 256 x 256, profiles/r03_parity_measured.md):
    #   dec_bf16: loss 8.8e-4, total 3.7e-4, points max 7.8e-3, hamming 2.5e-3, track 4.2e-3
    #   all_bf16 (three boxes): loss 3.8e-3 .. 1.4e-2 (final_chamfer_loss), total 1.7e-3 .. 5.0e-3, points max 6.8e-2 .. 8.0e-2 / rms 1.6e-2,
    #             hamming 3.0e-2 .. 3.1e-2, track 1.3e-2 .. 5.4e-2 - MIOpen's bf16 weight-gradient kernels accumulate with atomics, so
    #             the encoder's gradients (hence the Adam trajectory) differ from run to run; the decoder-only flavour is run-to-run stable
    # r06: the dec_bf16 flavour's forward is pinned to the ORACLE with the same roundings at bs 64
    # (tests/test_parity_evidence_gpu.py:test_configs2_dec_bf16_bs64_256_matches_the_bf16_oracle) and the all_bf16 flavour to the
    # autocast oracle; what this self-comparison still holds for it is the five-step Adam trajectory
    "dec_bf16": dict(track=1e-2),
    "all_bf16": dict(loss=3e-2, total=1.5e-2, points=0.16, points_rms=4e-2, hamming=6e-2, track=0.12),
}


def test_configs2_precision_flavours_track_the_fp32_model_at_size():
    """BASELINE configs[2] (25 x 642 points, trans + scale heads, shape, contact + penetration) at bs 16, 256 x 256, same weights
    and batch, three flavours: `f32` (the oracle-pinned path), `dec_bf16` (decoder contractions on the bf16 matrix pipe) and
    `all_bf16` (additionally the ResNet under bf16 autocast with the fused bf16 BatchNorm kernels) = the configuration's
    stated precision, what `bench.py --config c3 --encoder-dtype bf16 --decoder-dtype bf16` times.  Every smooth loss term,
    the total, the predicted object points, the penetration mask and five Adam steps are held to measured-then-frozen bounds."""
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.synthetic import CONFIGS, make_batch
    from obman_train_amd.trainer import make_optimizer, read_losses, train_step
    from tests.conftest import record_measurement

    import warnings
    warnings.simplefilter("ignore")
    dev = torch.device("cuda", 0)
    B = 16
    sample = make_batch(B, dev, seed=3, image_size=256)
    out = {}
    for flavour in ("f32", "dec_bf16", "all_bf16"):
        torch.manual_seed(0)
        model = HandNet(**CONFIGS["c3"]).to(dev).train()
        if flavour != "f32":
            model.atlas_branch.decoder.mfma_dtype = "bf16"
        if flavour == "all_bf16":
            model.base_net.autocast_dtype = torch.bfloat16
        total, results, losses = model.forward(sample)
        rec = dict(total=float(total), losses={k: v for k, v in read_losses(losses).items() if v is not None},
                   points=results["objpoints3d"].detach().float().cpu(),
                   rep=results["contact_info"]["repulsion_masks"].cpu())
        opt = make_optimizer(model)
        rec["track"] = [float(train_step(model, opt, sample)[0]) for _ in range(5)]
        out[flavour] = rec
        del model, opt, total, results, losses
    ref = out["f32"]
    assert ref["points"].shape == (B, 16050, 3)
    scale = ref["points"].abs().max().item()
    measured = {}
    for flavour, bounds in FLAVOUR_BOUNDS.items():
        got = out[flavour]
        m = {"total": abs(got["total"] - ref["total"]) / abs(ref["total"]),
             "points": (got["points"] - ref["points"]).abs().max().item() / scale,
             "points_rms": (got["points"] - ref["points"]).square().mean().sqrt().item() / scale,
             "hamming": (got["rep"] != ref["rep"]).float().mean().item(),
             "track": max(abs(a - b) / abs(b) for a, b in zip(got["track"], ref["track"]))}
        terms = {}
        for k, v in ref["losses"].items():
            # penetration / attraction terms are means over thresholded vertex sets: covered by the mask distance and the total
            if abs(v) > 1e-6 and k in got["losses"] and k.startswith(("mano", "atlas", "final")):
                terms[k] = abs(got["losses"][k] - v) / abs(v)
        m["loss"] = max(terms.values())
        m["terms"] = terms
        assert all(np.isfinite(t) for t in got["track"]), got["track"]
        measured[flavour] = m
    record_measurement("configs2_precision_flavours[bs16,256x256]", measured)
    for flavour, bounds in FLAVOUR_BOUNDS.items():
        for key, bound in bounds.items():
            assert measured[flavour][key] <= bound, (flavour, key, measured[flavour])




def test_weighted_terms_on_device_matches_the_sequential_composition():
    """``ops.weighted_terms`` on the GPU (stack, multiply, sum) against the reference's term-by-term composition evaluated on the
    CPU: value within the summation order of the terms (3e-7 relative), gradients ``lambda_i`` exactly, result usable in place."""
    from obman_train_amd import ops

    torch.manual_seed(5)
    lambdas = [0.167, 0.167, 1e-5, 0.5]
    host = [torch.rand(()) * 5 for _ in lambdas]
    ref = torch.zeros(1)
    for lam, t in zip(lambdas, host):
        ref += lam * t
    terms = [t.cuda().requires_grad_() for t in host]
    out = ops.weighted_terms(list(zip(lambdas, terms)) + [(0.167, 0)], (1,))
    assert out.is_cuda and out.shape == (1,)
    torch.testing.assert_close(out.detach().cpu(), ref, rtol=3e-7, atol=0)
    out += 2.0
    out.backward()
    for lam, t in zip(lambdas, terms):
        torch.testing.assert_close(t.grad.cpu(), torch.tensor(lam), rtol=1e-7, atol=0)
    with pytest.raises(TypeError):
        ops.weighted_terms([(0.167, terms[0]), (None, 0)], ())
