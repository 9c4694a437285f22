"""Oracle: the whole ``HandNet.forward`` on CPU as a function of named tensors.

Follows ``mano_train/networks/handnet.py:198-392``: encoder (:207-210) -> MANO branch +
loss (:253-288) -> AtlasNet forward_inference (:310-329) -> contact loss (:330-373) ->
atlas loss (:376-386) -> ``(total_loss, results, losses)`` (:387-392), including the aliased
``mano_total_loss`` accumulator.  Branches that are dead in the
reference (absolute branch, 2-D joints; App. C #14) are not restated.

``named`` maps reference state-dict names (``base_net.*``, ``atlas_base_net.*``,
``mano_branch.*``, ``atlas_branch.*``) to CPU tensors (leaf tensors with
``requires_grad`` when gradients are wanted).  The ResNet is the stock torch module run
through ``torch.func.functional_call``.
"""
import numpy as np
import torch
from torch.func import functional_call

from . import atlas as _atlas
from . import contact as _contact
from . import mano as _mano
from .chamfer import batch_pairwise_dist

DEFAULTS = dict(  # HandNet.__init__ defaults, handnet.py:20-63
    atlas_lambda=None, atlas_final_lambda=None, atlas_mesh=True, atlas_lambda_regul_edges=0,
    atlas_predict_trans=False, atlas_trans_weight=1, atlas_predict_scale=False, atlas_scale_weight=1,
    atlas_separate_encoder=False, atlas_out_factor=200, contact_target="all", contact_zones="all",
    contact_lambda=0, contact_thresh=25, contact_mode="dist_sq", collision_thresh=25, collision_mode="dist_sq",
    collision_lambda=0, resnet_version=50, mano_comps=6, mano_use_shape=False, mano_lambda_pose_reg=0,
    mano_use_pca=True, mano_center_idx=9, mano_lambda_joints3d=None, mano_lambda_verts=None,
    mano_lambda_shape=None, atlas_patches=1,
)


def _sub(named, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in named.items() if k.startswith(prefix)}


def handnet_forward(named, cfg, sample, keys, packs, template_verts, template_faces, zones=None,
                    resnet_shell=None, training=True, bn_training=None, no_loss=False, features=None, mfma_round=None,
                    encoder_autocast=None):
    """``keys`` = namespace with images/verts3d/joints3d/objpoints3d/sides keys of ``sample``.
    ``bn_training``: BN mode (defaults to ``training``; False = --freeze_batchnorm, epochpass3d.py:48-52).
    ``features`` (test hook): encoder output [B,C] used instead of running the ResNet, so everything downstream of the encoder
    can be compared at tight tolerance, free of convolution-library round-off.
    ``mfma_round`` / ``encoder_autocast`` (not in the reference; BASELINE configs[2]'s stated precision): the decoder's layer-2/3
    contraction operands and stored outputs rounded by ``mfma_round`` (``oracle/atlas.py:pointgen``), the ResNet under
    ``torch.autocast("cpu", dtype=encoder_autocast)`` with fp32 features handed to the heads - the model of the build's
    ``decoder.mfma_dtype = "bf16"`` / ``base_net.autocast_dtype`` flavours, so that those are compared with the oracle and not
    with the build's own fp32 run."""
    c = dict(DEFAULTS)
    c.update(cfg)
    bn_train = training if bn_training is None else bn_training
    results, losses = {}, {}
    total = None
    image = sample[keys.images]
    if features is not None:
        feats = features
    else:
        resnet_shell.train(bn_train)
        if encoder_autocast is not None:
            with torch.autocast("cpu", dtype=encoder_autocast):
                feats, _ = functional_call(resnet_shell, _sub(named, "base_net."), (image,))
            feats = feats.float()
        else:
            feats, _ = functional_call(resnet_shell, _sub(named, "base_net."), (image,))
    if c["atlas_separate_encoder"]:
        atlas_feats, _ = functional_call(resnet_shell, _sub(named, "atlas_base_net."), (image,))
    mano_lambdas = bool(c["mano_lambda_verts"] or c["mano_lambda_joints3d"])
    has_hand_gt = keys.joints3d in sample or keys.verts3d in sample
    mano_res = None
    if has_hand_gt and keys.sides in sample and mano_lambdas:
        mano_res = _mano.mano_branch(
            _sub(named, "mano_branch."), feats, sample[keys.sides], packs, ncomps=c["mano_comps"],
            center_idx=c["mano_center_idx"], use_shape=c["mano_use_shape"], use_pca=c["mano_use_pca"],
            root_palm=(sample.get("root") == "palm"),
        )
        if not no_loss:
            mano_total, mano_losses = _mano.mano_loss(
                mano_res, sample.get(keys.verts3d), sample.get(keys.joints3d),
                lambda_verts=c["mano_lambda_verts"], lambda_joints3d=c["mano_lambda_joints3d"],
                lambda_shape=c["mano_lambda_shape"], lambda_pose_reg=c["mano_lambda_pose_reg"],
            )
            total = mano_total  # same object: later in-place adds show up in losses["mano_total_loss"]
            losses.update(mano_losses)
        results.update(mano_res)
    predict_atlas = keys.objpoints3d in sample and (c["atlas_lambda"] or c["atlas_final_lambda"])
    if predict_atlas:
        atlas_res = _atlas.forward_inference(
            _sub(named, "atlas_branch."), feats, template_verts, template_faces,
            predict_trans=c["atlas_predict_trans"], predict_scale=c["atlas_predict_scale"],
            separate_features=atlas_feats if c["atlas_separate_encoder"] else None,
            training=bn_train, out_factor=c["atlas_out_factor"], mfma_round=mfma_round,
        )
        if c["contact_lambda"] or c["collision_lambda"]:
            attr, penetr, info, metrics = _contact.compute_contact_loss(
                mano_res["verts"], packs["right"]["faces"], atlas_res["objpoints3d"], template_faces, zones=zones,
                contact_thresh=c["contact_thresh"], contact_mode=c["contact_mode"],
                collision_thresh=c["collision_thresh"], collision_mode=c["collision_mode"],
                contact_target=c["contact_target"], contact_zones=c["contact_zones"],
                obj_patches=int(c.get("atlas_patches", 1)),
            )
            if not no_loss:
                if keys.verts3d in sample and keys.objpoints3d in sample:
                    gt_h2o = batch_pairwise_dist(sample[keys.verts3d], sample[keys.objpoints3d]).min(2)[0]
                    ious, auc = _contact.meshiou(gt_h2o, info["min_dists"])
                    info["batch_ious"] = ious
                    losses["contact_auc"] = auc
                contact_loss = c["contact_lambda"] * attr + c["collision_lambda"] * penetr
                total += contact_loss
                losses["penetration_loss"] = penetr
                losses["attraction_loss"] = attr
                losses["contact_loss"] = contact_loss
                losses.update(metrics)
            results["contact_info"] = info
        results.update(atlas_res)
        if not no_loss:
            a_total, a_losses = _atlas.atlas_loss(
                atlas_res, sample[keys.objpoints3d], lambda_atlas=c["atlas_lambda"],
                final_lambda_atlas=c["atlas_final_lambda"], trans_weight=c["atlas_trans_weight"],
                scale_weight=c["atlas_scale_weight"], edge_regul_lambda=c["atlas_lambda_regul_edges"],
            )
            if total is None:
                total = a_total
            else:
                total += a_total
            losses.update(a_losses)
    losses["total_loss"] = total
    return total, results, losses
