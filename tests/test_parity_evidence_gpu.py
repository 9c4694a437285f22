"""GPU: the parity caveats VERDICT r04 left open, closed with measurements instead of explanations.

(a) configs[2] - the WHOLE model at bs 64 / 256 x 256 in fp32 (25 patches = 16 050 object vertices, 32 000 faces, trans + scale
    heads, shape, contact + penetration) against one host step of `oracle.handnet_forward` on the same weights and batch.  The
    oracle's multi-patch inside test already runs patch by patch and its pairwise tensors are a few GB at this size, so the host
    step needs no chunking by hand (~2 - 4 min on the GPU box's cores); BatchNorm statistics and the batch-global masked means
    (contactloss.py:50-57) are those of the full batch on both sides.
(b) which loss term carries the 3e-4 of the B = 2 run (tests/test_fullsize_gpu.py, real encoder): every term's relative error is
    recorded (profiles/r05_parity_measured.md) by the same helper, at B = 2 / 64 x 64 and at bs 64.
(d) where the whole-model GRADIENT differences of configs[1] at bs 64 come from (2.4e-2 of the largest entry, 7e-3 in L2: bounds
    6e-2 / 2e-2 in tests/test_benchsize_gpu.py).  Two runs split the model at the encoder output:
      downstream - the oracle's features injected on both sides: every gradient below the encoder (decoder, heads, MANO branch,
                   and d loss / d features itself) must agree to 1e-3 in relative L2 - the kernels, the Chamfer arg-mins and the
                   ReLU masks are the oracle's when they see the oracle's features;
      encoder    - the same cotangent d loss / d features pushed through the encoder on the GPU, on the host in fp32 and on the
                   host in fp64: GPU and host-fp32 are EQUALLY far (3 - 6e-3 in relative L2, every layer) from fp64 - ReLU
                   masks that flip on 3e-6 of forward round-off; a fraction f of flipped elements is sqrt(f) in L2.
    So the whole-model difference is the fp32 floor of a ReLU network's backward at this size, not an error of either half.
"""
import warnings
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from obman_train_amd.contactzones import load_contacts

pytestmark = pytest.mark.gpu


def _keys():
    from obman_train_amd.queries import BaseQueries, TransQueries

    return SimpleNamespace(images=TransQueries.images, verts3d=TransQueries.verts3d, joints3d=TransQueries.joints3d,
                           objpoints3d=TransQueries.objpoints3d, sides=BaseQueries.sides)


class _FixedFeatures(torch.nn.Module):
    """Encoder stand-in returning a given feature tensor (a Parameter, so its gradient can be compared)."""

    def __init__(self, feats):
        super().__init__()
        self.feats = torch.nn.Parameter(feats.clone())
        self.fc = torch.nn.Linear(1, 1)  # HandNet.unused_parameters() looks for the classifier head

    def forward(self, image):
        return self.feats, {}


def _l2(g, w):
    g, w = g.detach().cpu().double(), w.detach().double()
    return float((g - w).norm() / w.norm().clamp_min(1e-30))


def _mx(g, w):
    g, w = g.detach().cpu(), w.detach()
    return float((g - w).abs().max() / w.abs().max().clamp_min(1e-30))


def _run_both(cfg_name, B, res, inject, grad_names, flavour="f32"):
    """One train-mode forward + backward of HandNet(**CONFIGS[cfg_name]) on the GPU and of the oracle on the host, same weights and
    batch (the batch bench.py times, seed 0).  inject: the encoder is replaced by fixed features on both sides.
    flavour: "f32" | "dec_bf16" (decoder contractions on the bf16 matrix pipe; oracle: operands and stored layer outputs rounded
    to bf16) | "all_bf16" (additionally the encoder under bf16 autocast on both sides)."""
    from oracle import handnet as ohandnet
    from oracle import mano as omano
    from obman_train_amd.mano_params import synthetic_mano
    from obman_train_amd.networks.bases import resnet
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.queries import BaseQueries
    from obman_train_amd.synthetic import CONFIGS, make_batch

    warnings.simplefilter("ignore")
    cfg = dict(CONFIGS[cfg_name])
    torch.manual_seed(0)
    model = HandNet(**cfg).train()
    named = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in named.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_()
    sample = make_batch(B, "cpu", seed=0, image_size=res)
    sample[BaseQueries.sides] = ["left", "right"] * (B // 2)
    packs = {s: omano.pack_to_torch(synthetic_mano(s)) for s in ("right", "left")}
    feats = f_o = None
    if inject:
        feats = torch.randn(B, 512, generator=torch.Generator().manual_seed(5)) * 0.5
        f_o = feats.clone().requires_grad_()
    o_total, o_res, o_losses = ohandnet.handnet_forward(
        named, cfg, dict(sample), _keys(), packs, model.atlas_branch.test_verts.clone(), model.atlas_branch.test_faces,
        zones=load_contacts()[1], resnet_shell=resnet.resnet18(), training=True, features=f_o,
        mfma_round=(lambda t: t.bfloat16().float()) if flavour != "f32" else None,
        encoder_autocast=torch.bfloat16 if flavour == "all_bf16" else None)
    o_total.backward()
    if inject:
        model.base_net = _FixedFeatures(feats)
    if flavour != "f32":
        model.atlas_branch.decoder.mfma_dtype = "bf16"
    if flavour == "all_bf16":
        model.base_net.autocast_dtype = torch.bfloat16
    model.cuda()
    total, out, losses = model.forward(sample)
    total.backward()
    torch.cuda.synchronize()

    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)  # noqa: E731
    m = {"total": rel(total, o_total), "terms": {}}
    for k, v in o_losses.items():
        if v is None:
            assert losses[k] is None, k
        elif abs(float(v)) > 1e-5:
            m["terms"][k] = rel(losses[k], v)
    m["worst_term"] = max(m["terms"], key=m["terms"].get)
    m["worst_loss"] = m["terms"][m["worst_term"]]
    for k in ("verts", "joints", "objpoints3d"):
        w = o_res[k].detach()
        m[k + "_of_scale"] = float((out[k].detach().cpu() - w).abs().max() / w.abs().max())
    if "contact_info" in o_res:
        gi, oi = out["contact_info"], o_res["contact_info"]
        n = oi["repulsion_masks"].numel()
        m["repulsion_hamming"] = int((gi["repulsion_masks"].cpu() != oi["repulsion_masks"]).sum()) / n
        m["attraction_hamming"] = int(((gi["attraction_masks"].cpu() != 0) != (oi["attraction_masks"] != 0)).sum()) / n
        m["min_dists_rel"] = float(((gi["min_dists"].detach().cpu() - oi["min_dists"].detach()).abs()
                                    / oi["min_dists"].detach().abs().clamp_min(1e-3)).max())
    got = dict(model.named_parameters())
    m["grads_l2"], m["grads_max"] = {}, {}
    for name in grad_names:
        if name in got and got[name].grad is not None and named[name].grad is not None:
            m["grads_l2"][name], m["grads_max"][name] = _l2(got[name].grad, named[name].grad), _mx(got[name].grad, named[name].grad)
    if inject:
        m["grads_l2"]["features"], m["grads_max"]["features"] = _l2(model.base_net.feats.grad, f_o.grad), _mx(model.base_net.feats.grad, f_o.grad)
    return m, (total, out, losses), (o_total, o_res, o_losses)


_DOWNSTREAM = ("mano_branch.pose_reg.weight", "mano_branch.shape_reg.0.weight", "mano_branch.base_layer.0.weight",
               "atlas_branch.decoder.conv1.weight", "atlas_branch.decoder.conv2.weight", "atlas_branch.decoder.conv3.weight",
               "atlas_branch.decoder.conv4.weight", "atlas_branch.decoder.bn2.weight", "atlas_branch.decode_scale.2.weight",
               "atlas_branch.decode_trans.2.weight")
_ENCODER = ("base_net.layer4.1.conv2.weight", "base_net.layer2.0.conv1.weight", "base_net.conv1.weight", "base_net.bn1.weight")


def test_configs2_model_bs64_256_matches_cpu_oracle():
    """(a) + (b) at bs 64."""
    from tests.conftest import record_measurement

    m, _, _ = _run_both("c3", 64, 256, False, _DOWNSTREAM + _ENCODER)
    record_measurement("configs2_bs64_256_vs_oracle", m)
    # north_star: loss scalars and outputs within 1e-4 relative.  Frozen at small multiples of the MI355X measurement
    # (profiles/r05_parity_measured.md); the contact terms sit on hard thresholds (a vertex entering / leaving a mask moves a
    # batch-global masked mean by 1 / count), so their bound is the mask Hamming distance, asserted separately
    assert m["total"] <= 2e-5, m
    assert m["verts_of_scale"] <= 1e-4 and m["joints_of_scale"] <= 1e-4 and m["objpoints3d_of_scale"] <= 1e-4, m
    assert m["repulsion_hamming"] <= 2e-4 and m["attraction_hamming"] <= 2e-4, m
    soft = {k: v for k, v in m["terms"].items() if k not in ("penetration_loss", "attraction_loss", "contact_loss", "max_penetr",
                                                               "mean_penetr", "contact_auc")}
    assert max(soft.values()) <= 1e-4, m
    # r06: <= 10 x the measured values (worst term 4.3e-6 -> north_star's 1e-4 itself; gradients through 18
    # convolution layers of two libraries 4.9e-3 -> 2e-2), where r05 asserted 1e-3 / 3e-2.  The 1e-4 holds for every SMOOTH term
    # (asserted above); the six threshold quantities move in quanta - contact_auc came out at 1.1e-4 on one box (one vertex across
    # one threshold of the sweep; masks identical, every smooth term <= 1.3e-6), 4.3e-6 on the others - and keep r05's 1e-3
    assert m["worst_loss"] <= 1e-3, m
    assert max(m["grads_l2"].values()) <= 2e-2, m


# configs[2] in its STATED precision against the ORACLE (VERDICT r05 task 2b).  Frozen at ~3x the MI355X measurement
# (profiles/r06_parity_measured.md).  What the numbers mean: the oracle applies the same roundings (contraction operands and the
# stored h2 / h3 in bf16, fp32 accumulation and statistics) but sums in another order, so a value on a bf16 rounding boundary
# lands on the neighbouring bf16 number (2^-8 relative) in a few activations; the chamfer / vertex terms average that out, the
# contact terms sit on hard thresholds (their bound is the mask Hamming distance).
BF16_VS_ORACLE = {
    # measured (r06, MI355X): total 2.9e-5, worst smooth term 4.5e-5 (atlas_objpoints3d), points max 6.4e-3 / rms 3.6e-4 of scale,
    # masks 3.6e-4 / 4.0e-5, hand vertices 4.1e-7 - the decoder flavour meets north_star's 1e-4 on every smooth scalar
    "dec_bf16": dict(total=1e-4, soft=1.5e-4, points=2e-2, points_rms=1.2e-3, hamming=1.2e-3, verts=1e-5),
    # measured: total 7.9e-3, worst smooth term 3.0e-2 (atlas_objpoints3d), points max 9.1e-2 / rms 2.0e-2, masks 3.2e-2 / 1.8e-3,
    # hand vertices 2.3e-3 (two convolution libraries in bf16, 18 layers deep)
    "all_bf16": dict(total=2.5e-2, soft=6e-2, points=0.2, points_rms=5e-2, hamming=6e-2, verts=8e-3),
}
_HARD = ("penetration_loss", "attraction_loss", "contact_loss", "max_penetr", "mean_penetr", "contact_auc")


def _soft(m):
    soft = {k: v for k, v in m["terms"].items() if k not in _HARD}
    m["worst_soft_term"] = max(soft, key=soft.get)
    m["worst_soft"] = soft[m["worst_soft_term"]]


def _check_flavour(name, m, b):
    assert m["total"] <= b["total"], (name, m)
    assert m["worst_soft"] <= b["soft"], (name, m)
    assert m["objpoints3d_of_scale"] <= b["points"] and m["objpoints3d_rms_of_scale"] <= b["points_rms"], (name, m)
    assert m["verts_of_scale"] <= b["verts"] and m["joints_of_scale"] <= b["verts"], (name, m)
    assert m["repulsion_hamming"] <= b["hamming"] and m["attraction_hamming"] <= b["hamming"], (name, m)


def test_configs2_dec_bf16_bs64_256_matches_the_bf16_oracle():
    """configs[2] with the decoder's contractions on the bf16 matrix pipe (`decoder.mfma_dtype = "bf16"`), whole model, bs 64,
    256 x 256, against one host step of the oracle WITH the same operand / storage roundings - total, every loss term, object
    points (max and rms), hand vertices, both contact masks.  Until round 5 this flavour was only compared with the build's own
    fp32 run at bs 16 (tests/test_handnet_gpu.py FLAVOUR_BOUNDS)."""
    from tests.conftest import record_measurement

    m, (_, out, _), (_, o_res, _) = _run_both("c3", 64, 256, False, _DOWNSTREAM, flavour="dec_bf16")
    w = o_res["objpoints3d"].detach()
    m["objpoints3d_rms_of_scale"] = float((out["objpoints3d"].detach().cpu() - w).square().mean().sqrt() / w.abs().max())
    _soft(m)
    record_measurement("configs2_dec_bf16_bs64_256_vs_bf16_oracle", m)
    _check_flavour("dec_bf16", m, BF16_VS_ORACLE["dec_bf16"])


def test_configs2_all_bf16_bs16_256_matches_the_autocast_bf16_oracle():
    """configs[2] as `bench.py --config c3 --encoder-dtype bf16 --decoder-dtype bf16` runs it (ResNet under bf16 autocast with the
    fused bf16 BatchNorm kernels + bf16 decoder) against the oracle with the ResNet under CPU bf16 autocast (oneDNN's bf16
    convolutions, fp32 accumulation) and the rounded decoder.  Two convolution libraries in bf16, 18 layers deep: the bounds are
    bf16-noise bounds, measured then frozen - what they pin is that the stated-precision build is a bf16-accurate evaluation of
    the REFERENCE's model, not merely of this build's fp32 flavour."""
    from tests.conftest import record_measurement

    m, (_, out, _), (_, o_res, _) = _run_both("c3", 16, 256, False, _DOWNSTREAM, flavour="all_bf16")
    w = o_res["objpoints3d"].detach()
    m["objpoints3d_rms_of_scale"] = float((out["objpoints3d"].detach().cpu() - w).square().mean().sqrt() / w.abs().max())
    _soft(m)
    record_measurement("configs2_all_bf16_bs16_256_vs_autocast_oracle", m)
    _check_flavour("all_bf16", m, BF16_VS_ORACLE["all_bf16"])


def test_configs2_b2_names_the_loose_term():
    """(b) the B = 2 / 64 x 64 run of tests/test_fullsize_gpu.py with every term recorded: the 3e-4 belongs to ONE term."""
    from tests.conftest import record_measurement

    m, _, _ = _run_both("c3", 2, 64, False, _DOWNSTREAM)
    record_measurement("configs2_b2_64_terms", m)
    loose = {k: v for k, v in m["terms"].items() if v > 1e-4}
    # whatever is above north_star's 1e-4 must be a contact-side term (hard thresholds on 2 x 778 vertices), never a smooth one
    assert set(loose) <= {"penetration_loss", "attraction_loss", "contact_loss", "max_penetr", "mean_penetr", "contact_auc"}, m


def test_configs1_gradients_downstream_of_the_encoder_bs64():
    """(d) downstream half: injected features."""
    from tests.conftest import record_measurement

    m, _, _ = _run_both("c2", 64, 256, True, _DOWNSTREAM)
    record_measurement("configs1_bs64_downstream_injected", m)
    assert m["total"] <= 1e-5 and m["worst_loss"] <= 1e-4, m
    assert len(m["grads_l2"]) >= 6, m
    bad = {k: v for k, v in m["grads_l2"].items() if not v <= 1e-3}
    assert not bad, (bad, m)


def test_configs1_gradients_through_the_encoder_bs64():
    """(d) encoder half: one cotangent through the encoder (train-mode BatchNorm, bs 64, 256 x 256) three times - the GPU path
    (MIOpen convolutions + the fused BatchNorm kernels, fp32), the same modules on the host in fp32 (oneDNN) and in fp64.

    Measured (profiles/r05_parity_measured.md): the features agree to 3e-6 of scale, yet the weight gradients of GPU and host-fp32
    differ by 3 - 6e-3 in relative L2 in EVERY layer below layer4 - and so does each of them from the fp64 run.  A ReLU network's
    backward is discontinuous: an activation within round-off of zero takes the other branch, the whole incoming gradient of that
    element appears or disappears, and a fraction f of flipped elements shows as sqrt(f) in L2 (1e-5 of the elements -> 3e-3).
    That is the floor of ANY fp32 evaluation at this size, not an error of these kernels: the assertion is that the GPU is no
    further from fp64 than the host's own fp32 run is (x 2), plus an absolute cap."""
    from obman_train_amd.networks.bases import resnet
    from obman_train_amd.synthetic import make_batch
    from obman_train_amd.queries import TransQueries
    from tests.conftest import record_measurement

    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    enc = resnet.resnet18().train()
    ref32 = resnet.resnet18().train()
    ref32.load_state_dict(enc.state_dict())
    ref64 = resnet.resnet18().double().train()
    ref64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in enc.state_dict().items()})
    images = make_batch(64, "cpu", seed=0, image_size=256)[TransQueries.images]
    cot = torch.randn(64, 512, generator=torch.Generator().manual_seed(9))
    f64, _ = ref64(images.double())
    (f64 * cot.double()).sum().backward()
    f32, _ = ref32(images)
    (f32 * cot).sum().backward()
    enc.cuda()
    f_gpu, _ = enc(images.cuda())
    (f_gpu * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    scale = float(f64.detach().abs().max())
    m = {"features_gpu_vs_f64_of_scale": float((f_gpu.detach().cpu().double() - f64.detach()).abs().max()) / scale,
         "features_host32_vs_f64_of_scale": float((f32.detach().double() - f64.detach()).abs().max()) / scale,
         "gpu_vs_f64_l2": {}, "host32_vs_f64_l2": {}, "gpu_vs_host32_l2": {}}
    w64, w32 = dict(ref64.named_parameters()), dict(ref32.named_parameters())
    for name, p in enc.named_parameters():
        if p.grad is None or w64[name].grad is None or name.startswith("fc."):
            continue
        m["gpu_vs_f64_l2"][name] = _l2(p.grad, w64[name].grad)
        m["host32_vs_f64_l2"][name] = _l2(w32[name].grad, w64[name].grad)
        m["gpu_vs_host32_l2"][name] = _l2(p.grad, w32[name].grad)
    worst = lambda d: max(d.values())  # noqa: E731
    m["worst"] = {k: worst(m[k]) for k in ("gpu_vs_f64_l2", "host32_vs_f64_l2", "gpu_vs_host32_l2")}
    record_measurement("configs1_bs64_encoder_cotangent", m)
    assert len(m["gpu_vs_f64_l2"]) >= 60, len(m["gpu_vs_f64_l2"])
    assert m["features_gpu_vs_f64_of_scale"] <= 2e-5, m["features_gpu_vs_f64_of_scale"]
    assert m["worst"]["gpu_vs_f64_l2"] <= 2.0 * m["worst"]["host32_vs_f64_l2"] + 1e-4, m["worst"]
    assert m["worst"]["gpu_vs_f64_l2"] <= 1.5e-2, m["worst"]
    # the last layers, where few ReLU decisions lie upstream, are tight
    assert m["gpu_vs_f64_l2"]["layer4.1.bn2.weight"] <= 1e-3, m["gpu_vs_f64_l2"]["layer4.1.bn2.weight"]
