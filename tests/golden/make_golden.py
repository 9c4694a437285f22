"""DEV-CONTAINER ONLY - generate golden vectors by importing the *reference itself*.

    python tests/golden/make_golden.py

Imports ``/root/reference`` under the import-time shims of SURVEY Appendix A (identity
``.cuda()``, ``torch.cuda.LongTensor``, stub ``trimesh.creation.icosphere`` backed by this repo's
icosphere generator, empty ``cv2``, a ``manopth.manolayer.ManoLayer`` backed by the oracle MANO
restatement with the synthetic parameter pack, ``resnet18 -> TinyEncoder``), feeds seeded inputs to
the reference's own functions and stores inputs + outputs (+ input gradients) as small ``.npz``.
Nothing from the reference is copied: fixtures are data.  The reference's Python never travels to
the GPU box; tests read only the ``.npz``.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("OBMAN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from obman_train_amd.icosphere import icosphere  # noqa: E402
from obman_train_amd.mano_params import synthetic_mano  # noqa: E402
from obman_train_amd.contactzones import hand_template  # noqa: E402
from oracle import mano as omano  # noqa: E402
from tests.golden.common import (  # noqa: E402
    TinyEncoder, load_seeded, pack_bits, seeded_state, synth_hand_object,
)


def install_shims():
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.LongTensor = torch.LongTensor

    tm = types.ModuleType("trimesh")
    creation = types.ModuleType("trimesh.creation")

    def _ico(subdivisions=3, **_):
        v, f = icosphere(subdivisions)
        return types.SimpleNamespace(vertices=v, faces=f)

    creation.icosphere = _ico
    tm.creation = creation
    sys.modules["trimesh"] = tm
    sys.modules["trimesh.creation"] = creation
    sys.modules["cv2"] = types.ModuleType("cv2")

    class ManoLayer(nn.Module):
        def __init__(self, ncomps=6, center_idx=None, side="right", mano_root=None, use_pca=True, **_):
            super().__init__()
            self.pk = omano.pack_to_torch(synthetic_mano(side))
            self.ncomps, self.center_idx, self.use_pca = ncomps, center_idx, use_pca
            self.th_faces = self.pk["faces"]

        def forward(self, pose, th_betas=None, th_trans=None, root_palm=False):
            return omano.mano_lbs(self.pk, pose, th_betas, ncomps=self.ncomps, center_idx=self.center_idx,
                                  root_palm=bool(root_palm), use_pca=self.use_pca)

    mp = types.ModuleType("manopth")
    ml = types.ModuleType("manopth.manolayer")
    ml.ManoLayer = ManoLayer
    mp.manolayer = ml
    sys.modules["manopth"] = mp
    sys.modules["manopth.manolayer"] = ml


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **conv)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def deformed_sphere(subdiv, batch, seed, radius=45.0, centre=(0.0, 0.0, 0.0)):
    rng = np.random.RandomState(seed)
    v, f = icosphere(subdiv)
    scale = radius * (1.0 + 0.25 * np.sin(3.0 * v[:, :1]) * np.cos(2.0 * v[:, 1:2]))
    pts = v[None] * scale[None] * rng.uniform(0.7, 1.3, size=(batch, 1, 3)) + np.asarray(centre)[None, None]
    pts = pts + rng.normal(0, 0.5, size=pts.shape)
    return torch.from_numpy(pts.astype(np.float32)), f


def gen_chamfer():
    from mano_train.networks.branches.atlasutils import ChamferLoss

    rng = np.random.RandomState(1)
    preds = torch.from_numpy((rng.normal(0, 40, size=(3, 50, 3)) + 20).astype(np.float32)).requires_grad_()
    gts = torch.from_numpy((rng.normal(0, 40, size=(3, 37, 3)) + 20).astype(np.float32)).requires_grad_()
    cl = ChamferLoss()
    l1, l2 = cl(preds, gts)
    P = cl.batch_pairwise_dist(gts, preds)
    torch.mean(l1 + l2).backward()
    save("chamfer", preds=preds, gts=gts, loss_1=l1, loss_2=l2, P=P, grad_preds=preds.grad, grad_gts=gts.grad)


def gen_contains():
    from mano_train.networks.branches.contactutils import batch_mesh_contains_points

    obj, faces = deformed_sphere(1, 2, 3, radius=30.0)
    rng = np.random.RandomState(4)
    origins = torch.from_numpy(rng.normal(0, 28, size=(2, 90, 3)).astype(np.float32))
    tri = obj[:, torch.from_numpy(faces)]
    ext = batch_mesh_contains_points(origins, tri)
    save("contains", origins=origins, obj_verts=obj, faces=faces.astype(np.int32), exterior=ext.numpy())


def contact_inputs():
    tv, _ = hand_template()
    hand, _, _ = synth_hand_object(2, 8, 11, tv)
    # object overlapping the palm so both interior and exterior hand vertices exist
    obj, faces = deformed_sphere(2, 2, 12, radius=32.0, centre=(5.0, -25.0, 5.0))
    return hand, obj, faces


def gen_contact():
    from mano_train.networks.branches import contactloss

    hand0, obj0, faces = contact_inputs()
    hand_faces = torch.from_numpy(hand_template()[1].astype(np.int64))
    out = dict(hand=hand0, obj=obj0, faces=faces.astype(np.int32))
    combos = []
    for zones in ("all", "tips", "zones"):
        for mode in ("dist_sq", "dist", "dist_tanh"):
            for target in ("all", "obj", "hand"):
                combos.append((zones, mode, mode, target))
    combos.append(("zones", "dist_tanh", "dist_sq", "all"))
    combos.append(("all", "dist", "dist_tanh", "all"))
    cwd = os.getcwd()
    os.chdir(REF)  # contactloss.py:263 reads the relative path assets/contact_zones.pkl
    try:
        for ci, (zones, cmode, kmode, target) in enumerate(combos):
            hand = hand0.clone().requires_grad_()
            obj = obj0.clone().requires_grad_()
            missed, penetr, info, metrics = contactloss.compute_contact_loss(
                hand, hand_faces, obj, faces, contact_thresh=10, contact_mode=cmode, collision_thresh=20,
                collision_mode=kmode, contact_target=target, contact_zones=zones,
            )
            loss = missed.sum() + 2.0 * penetr.sum()
            if loss.requires_grad:
                loss.backward()
            tag = "c%02d_" % ci
            out[tag + "missed"] = missed.detach().reshape(-1)
            out[tag + "penetr"] = penetr.detach().reshape(-1)
            out[tag + "max_penetr"] = metrics["max_penetr"]
            out[tag + "mean_penetr"] = metrics["mean_penetr"]
            out[tag + "attr_mask"] = pack_bits(info["attraction_masks"].numpy() != 0)
            out[tag + "rep_mask"] = pack_bits(info["repulsion_masks"].numpy())
            out[tag + "attr_dtype"] = str(info["attraction_masks"].dtype)
            out[tag + "grad_hand"] = hand.grad if hand.grad is not None else torch.zeros_like(hand)
            out[tag + "grad_obj"] = obj.grad if obj.grad is not None else torch.zeros_like(obj)
            if ci == 0:
                out["min_dists"] = info["min_dists"]
                out["contact_points"] = info["contact_points"]
        out["combos"] = np.array(["|".join(c) for c in combos])
        gt_d = contactloss.batch_pairwise_dist(hand0, obj0 + 3.0).min(2)[0]
        ious, auc = contactloss.meshiou(gt_d, out["min_dists"])
        out["iou_gt_dists"] = gt_d
        out["iou_batch"] = ious
        out["iou_auc"] = auc
    finally:
        os.chdir(cwd)
    save("contact", **out)


def gen_pointgen():
    from mano_train.networks.branches.atlasutils import PointGenCon

    dec = load_seeded(PointGenCon(bottleneck_size=35, out_factor=200), 21)
    rng = np.random.RandomState(22)
    x = torch.from_numpy(rng.normal(0, 1, size=(3, 35, 42)).astype(np.float32)).requires_grad_()
    dec.train()
    y = dec(x)
    (y * torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))).sum().backward()
    grads = {"grad_" + k.replace(".", "_"): p.grad for k, p in dec.named_parameters()}
    rm = {"after_" + k.replace(".", "_"): v.clone() for k, v in dec.state_dict().items() if "running" in k}
    dec.eval()
    y_eval = dec(x)
    save("pointgen", x=x, y_train=y, y_eval=y_eval, grad_x=x.grad, seed=21, wseed=22, **grads, **rm)


def gen_atlas():
    from mano_train.networks.branches.atlasbranch import AtlasBranch, AtlasLoss, edge_loss
    from handobjectdatasets.queries import TransQueries

    br = AtlasBranch(use_residual=False, bottleneck_size=32, inference_ico_divisions=1,
                     predict_trans=True, predict_scale=True, out_factor=200)
    load_seeded(br, 31)
    br.train()
    rng = np.random.RandomState(32)
    feats = torch.from_numpy(rng.normal(0, 1, size=(3, 32)).astype(np.float32)).requires_grad_()
    gt = torch.from_numpy((rng.normal(0, 30, size=(3, 40, 3)) + np.array([10, -20, 5])).astype(np.float32))
    res = br.forward_inference(feats)
    out = dict(feats=feats, gt=gt, objpoints3d=res["objpoints3d"], objtrans=res["objtrans"],
               objscale=res["objscale"], centered=res["objpointscentered3d"], faces=np.asarray(res["objfaces"]))
    loss_mod = AtlasLoss(lambda_atlas=0.5, final_lambda_atlas=0.167, trans_weight=0.167, scale_weight=0.167,
                         edge_regul_lambda=0.3)
    total, parts = loss_mod.compute_loss(res, {TransQueries.objpoints3d: gt})
    total.backward()
    out.update(total=total.detach().reshape(-1), grad_feats=feats.grad,
               grad_conv4=br.decoder.conv4.weight.grad, grad_trans_bias=br.decode_trans[2].bias.grad)
    for k, v in parts.items():
        if v is not None:
            out["loss_" + k] = v.detach().reshape(-1)
    out["edge"] = edge_loss(res["objpointscentered3d"], res["objfaces"]).detach().reshape(-1)
    # second branch flavour: no trans/scale head (atlasbranch.py:255-265)
    br2 = AtlasBranch(use_residual=False, bottleneck_size=32, inference_ico_divisions=1, out_factor=200)
    load_seeded(br2, 33)
    br2.eval()
    res2 = br2.forward_inference(feats.detach())
    tot2, parts2 = AtlasLoss(lambda_atlas=0.167, final_lambda_atlas=None).compute_loss(res2, {TransQueries.objpoints3d: gt})
    out.update(plain_points=res2["objpoints3d"], plain_total=tot2.detach().reshape(-1),
               plain_sym=parts2["atlas_objpoints3d"].detach().reshape(-1))
    save("atlas", **out)


def gen_manobranch():
    from mano_train.networks.branches.manobranch import ManoBranch, ManoLoss
    from handobjectdatasets.queries import TransQueries

    br = ManoBranch(ncomps=30, base_neurons=[512, 64, 32], center_idx=0, use_shape=True, use_pca=True,
                    adapt_skeleton=False)
    sd = {k: v for k, v in br.state_dict().items()}
    new = seeded_state({k: v.shape for k, v in sd.items()}, 41)
    for k in ("pose_reg.weight", "pose_reg.bias", "shape_reg.0.weight", "shape_reg.0.bias"):
        new[k] = new[k] * 0.3
    br.load_state_dict(new)
    rng = np.random.RandomState(42)
    feats = torch.from_numpy(rng.normal(0, 1, size=(4, 512)).astype(np.float32)).requires_grad_()
    sides = ["left", "right", "left", "left"]
    res = br(feats, sides=sides, root_palm=False)
    tv, _ = hand_template()
    gtv, gtj, _ = synth_hand_object(4, 8, 43, tv)
    total, parts = ManoLoss(lambda_verts=0.167, lambda_joints3d=0.167, lambda_shape=0.167,
                            lambda_pose_reg=0.167).compute_loss(
        res, {TransQueries.verts3d: gtv, TransQueries.joints3d: gtj})
    total.backward()
    out = dict(feats=feats, sides=np.array(sides), gt_verts=gtv, gt_joints=gtj, verts=res["verts"],
               joints=res["joints"], shape=res["shape"], pose=res["pose"], total=total.detach(),
               grad_feats=feats.grad)
    for k, v in parts.items():
        if v is not None:
            out["loss_" + k] = v.detach().reshape(-1)
    save("manobranch", **out)


def gen_handnet():
    from mano_train.networks.bases import resnet as ref_resnet
    from handobjectdatasets.queries import TransQueries, BaseQueries

    ref_resnet.resnet18 = lambda pretrained=False, **kw: TinyEncoder()
    from mano_train.networks.handnet import HandNet

    cfg = dict(
        atlas_lambda=0.167, atlas_final_lambda=0.167, atlas_mesh=True, atlas_ico_divisions=2,
        atlas_predict_trans=True, atlas_trans_weight=0.167, atlas_predict_scale=True, atlas_scale_weight=0.167,
        atlas_lambda_regul_edges=0.1, contact_target="all", contact_zones="zones", contact_lambda=1.0,
        contact_thresh=10, contact_mode="dist_tanh", collision_thresh=20, collision_mode="dist_tanh",
        collision_lambda=1.0, resnet_version=18, mano_neurons=[64, 32], mano_comps=30, mano_use_shape=True,
        mano_lambda_pose_reg=0.167, mano_use_pca=True, mano_center_idx=0, mano_lambda_joints3d=0.167,
        mano_lambda_verts=0.167, mano_lambda_shape=0.167,
    )
    tv, _ = hand_template()
    gtv, gtj, gto = synth_hand_object(3, 50, 52, tv)
    rng = np.random.RandomState(53)
    images = torch.from_numpy(rng.uniform(-0.5, 0.5, size=(3, 3, 32, 32)).astype(np.float32))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        for tag, train_mode in (("train", True), ("eval", False)):
            torch.manual_seed(0)
            model = HandNet(**cfg)
            sd = model.state_dict()
            new = seeded_state({k: v.shape for k, v in sd.items() if "mano_layer" not in k}, 51)
            for k in list(new):
                if k.startswith("mano_branch.pose_reg") or k.startswith("mano_branch.shape_reg"):
                    new[k] = new[k] * 0.3
                if k.startswith("atlas_branch.decoder.conv4"):
                    new[k] = new[k] * 0.2
            model.load_state_dict(new, strict=False)
            model.train(train_mode)
            sample = {
                TransQueries.images: images, TransQueries.verts3d: gtv, TransQueries.joints3d: gtj,
                TransQueries.objpoints3d: gto, BaseQueries.sides: ["left", "left", "right"], "root": "wrist",
            }
            total, results, losses = model.forward(sample)
            total.backward()
            out = dict(images=images, gt_verts=gtv, gt_joints=gtj, gt_obj=gto, total=total.detach().reshape(-1),
                       verts=results["verts"], joints=results["joints"], objpoints3d=results["objpoints3d"],
                       objtrans=results["objtrans"], objscale=results["objscale"],
                       attr_mask=pack_bits(results["contact_info"]["attraction_masks"].numpy() != 0),
                       rep_mask=pack_bits(results["contact_info"]["repulsion_masks"].numpy()),
                       batch_ious=results["contact_info"]["batch_ious"],
                       grad_pose_bias=model.mano_branch.pose_reg.bias.grad,
                       grad_conv4=model.atlas_branch.decoder.conv4.weight.grad,
                       grad_enc_bias=model.base_net.proj.bias.grad,
                       cfg=np.array(repr(cfg)))
            for k, v in losses.items():
                if v is None:
                    continue
                out["loss_" + k] = v.detach().reshape(-1) if isinstance(v, torch.Tensor) else np.asarray(v).reshape(-1)
            save("handnet_" + tag, **out)
    finally:
        os.chdir(cwd)


def gen_laplacian():
    from mano_train.networks.branches.laplacianloss import Laplacian

    v, f = icosphere(2)
    tmpl = torch.from_numpy(v.astype(np.float32))
    rng = np.random.RandomState(61)
    V = torch.from_numpy((v[None] * rng.uniform(20, 60, size=(2, 1, 3)) + rng.normal(0, 3.0, size=(2,) + v.shape)).astype(np.float32))
    lap = Laplacian(f, tmpl)          # legacy autograd.Function: call its methods directly
    Lx = lap.forward(V)
    loss = torch.norm(Lx.view(-1, 3), p=2, dim=1).mean()
    g_Lx = Lx / torch.norm(Lx, p=2, dim=2, keepdim=True) / (Lx.shape[0] * Lx.shape[1])
    grad = lap.backward(g_Lx)
    save("laplacian", template=tmpl, faces=f.astype(np.int32), V=V, Lx=Lx, loss=loss.reshape(1), grad=grad)


def gen_zimeval():
    from mano_train.evaluation.zimeval import EvalUtil

    rng = np.random.RandomState(71)
    gt = rng.normal(0, 40, size=(12, 21, 3)).astype(np.float32)
    pred = (gt + rng.normal(0, 12, size=gt.shape)).astype(np.float32)
    vis = rng.uniform(size=(12, 21)) > 0.2
    vis[:, 5] = False  # a keypoint that is never visible
    ev = EvalUtil()
    for g, p, v in zip(gt, pred, vis):
        ev.feed(torch.from_numpy(g), torch.from_numpy(p), keypoint_vis=v)
    epe_mean, epe_joint, epe_median, auc, curve, thr = ev.get_measures(0, 50, 20)
    save("zimeval", gt=gt, pred=pred, vis=vis, epe_mean=epe_mean, epe_joint=np.array(epe_joint), epe_median=epe_median, auc=auc,
         curve=curve, thresholds=thr)


if __name__ == "__main__":
    torch.set_num_threads(4)
    install_shims()
    gen_chamfer()
    gen_contains()
    gen_contact()
    gen_pointgen()
    gen_atlas()
    gen_manobranch()
    gen_handnet()
    gen_laplacian()
    gen_zimeval()
