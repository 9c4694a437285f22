"""HandNet - the drop-in boundary of the hot path.

Mirror of ``mano_train/networks/handnet.py:19-392`` (reference): ``HandNet(**kwargs)`` with the
reference's keyword names and defaults (:20-63), ``forward(sample, no_loss, return_features,
force_objects) -> (total_loss [1], results, losses)`` (:198-392), ``decay_regul(gamma)`` (:188-196),
attributes ``base_net / atlas_base_net / mano_branch.faces / atlas_branch.decoder / .test_faces``
and the reference's state-dict key names, so ``traineval.py`` / ``epochpass3d.py`` can drive it
unchanged.  Flow: ResNet (MIOpen) -> ManoBranch (fused LBS kernel) + ManoLoss -> AtlasBranch
(decoder) -> compute_contact_loss (pair-min, inside test, tail kernels) -> AtlasLoss (Chamfer kernel).

Reference quirks kept: ``total_loss`` aliases ``mano_total_loss`` and is accumulated
in place; losses may be ``None``.  Dormant reference branches that crash there (absolute branch
:216-252 needs an undefined attribute; 2-D joints) raise NotImplementedError here.
Extension: ``atlas_patches`` (P sphere patches, BASELINE.json configs 3/5).
"""
from copy import deepcopy

import torch
from torch import nn

from obman_train_amd.networks.bases import resnet
from obman_train_amd.networks.branches.atlasbranch import AtlasBranch, AtlasLoss
from obman_train_amd.networks.branches.contactloss import compute_contact_loss, meshiou
from obman_train_amd.networks.branches.manobranch import ManoBranch, ManoLoss
from obman_train_amd import ops
from obman_train_amd.queries import BaseQueries, TransQueries

_DEFAULTS = dict(
    absolute_lambda=None, atlas_lambda=None, atlas_loss="chamfer", atlas_final_lambda=None, atlas_mesh=True,
    atlas_residual=False, atlas_lambda_regul_edges=0, atlas_lambda_laplacian=0, atlas_points_nb=600,
    atlas_predict_trans=False, atlas_trans_weight=1, atlas_predict_scale=False, atlas_scale_weight=1,
    atlas_use_tanh=False, atlas_ico_divisions=3, atlas_separate_encoder=False, atlas_out_factor=200,
    contact_target="all", contact_zones="all", contact_lambda=0, contact_thresh=25, contact_mode="dist_sq",
    collision_thresh=25, collision_mode="dist_sq", collision_lambda=0, fc_dropout=0, resnet_version=50,
    mano_adapt_skeleton=False, mano_neurons=[512], mano_comps=6, mano_use_shape=False, mano_lambda_pose_reg=0,
    mano_use_pca=True, mano_center_idx=9, mano_root="misc/mano", mano_lambda_joints3d=None,
    mano_lambda_joints2d=None, mano_lambda_verts=None, mano_lambda_shape=None, mano_lambda_pca=None,
    adapt_atlas_decoder=False,
    atlas_patches=1,  # extension (not in the reference)
)


class HandNet(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError("HandNet got unexpected keyword arguments: %s" % sorted(unknown))
        cfg = dict(_DEFAULTS)
        cfg.update(kwargs)
        self.cfg = cfg
        version = int(cfg["resnet_version"])
        if version == 18:
            feat, base_net = 512, resnet.resnet18(pretrained=True)
        elif version == 50:
            feat, base_net = 2048, resnet.resnet50(pretrained=True)
        else:
            raise NotImplementedError("Resnet {} not supported".format(cfg["resnet_version"]))
        if cfg["mano_lambda_joints2d"] or cfg["absolute_lambda"]:
            raise NotImplementedError("2-D joint / absolute-centre supervision is dormant and broken in the reference "
                                      "(handnet.py:242 reads an undefined attribute)")
        self.adapt_atlas_decoder = cfg["adapt_atlas_decoder"]
        self.atlas_separate_encoder = cfg["atlas_separate_encoder"]
        if self.adapt_atlas_decoder:
            self.atlas_adapter = nn.Linear(feat, feat)
        for k in ("contact_target", "contact_zones", "contact_lambda", "contact_thresh", "contact_mode",
                  "collision_lambda", "collision_thresh", "collision_mode", "absolute_lambda", "atlas_mesh",
                  "atlas_lambda", "atlas_final_lambda", "atlas_trans_weight", "atlas_scale_weight",
                  "mano_adapt_skeleton"):
            setattr(self, k, cfg[k])
        self.need_collisions = bool(cfg["contact_lambda"] or cfg["collision_lambda"])
        self.base_net = base_net
        if self.atlas_separate_encoder:
            self.atlas_base_net = deepcopy(base_net)
        self.mano_branch = ManoBranch(
            ncomps=cfg["mano_comps"], base_neurons=[feat] + list(cfg["mano_neurons"]),
            adapt_skeleton=cfg["mano_adapt_skeleton"], dropout=cfg["fc_dropout"], use_trans=False,
            mano_root=cfg["mano_root"], center_idx=cfg["mano_center_idx"], use_shape=cfg["mano_use_shape"],
            use_pca=cfg["mano_use_pca"])
        self.mano_lambdas = bool(cfg["mano_lambda_verts"] or cfg["mano_lambda_joints3d"] or cfg["mano_lambda_pca"])
        self.mano_loss = ManoLoss(
            lambda_verts=cfg["mano_lambda_verts"], lambda_joints3d=cfg["mano_lambda_joints3d"],
            lambda_shape=cfg["mano_lambda_shape"], lambda_pose_reg=cfg["mano_lambda_pose_reg"],
            lambda_pca=cfg["mano_lambda_pca"])
        self.lambda_joints2d = cfg["mano_lambda_joints2d"]
        self.atlas_branch = AtlasBranch(
            mode="sphere", use_residual=cfg["atlas_residual"], points_nb=cfg["atlas_points_nb"],
            predict_trans=cfg["atlas_predict_trans"], predict_scale=cfg["atlas_predict_scale"],
            inference_ico_divisions=cfg["atlas_ico_divisions"], bottleneck_size=feat, use_tanh=cfg["atlas_use_tanh"],
            out_factor=cfg["atlas_out_factor"], separate_encoder=self.atlas_separate_encoder,
            patches=cfg["atlas_patches"])
        self.atlas_loss = AtlasLoss(
            atlas_loss=cfg["atlas_loss"], lambda_atlas=cfg["atlas_lambda"],
            final_lambda_atlas=cfg["atlas_final_lambda"], trans_weight=cfg["atlas_trans_weight"],
            scale_weight=cfg["atlas_scale_weight"], edge_regul_lambda=cfg["atlas_lambda_regul_edges"],
            lambda_laplacian=cfg["atlas_lambda_laplacian"], laplacian_faces=self.atlas_branch.test_faces,
            laplacian_verts=self.atlas_branch.test_verts)

    def decay_regul(self, gamma):
        if self.atlas_loss.edge_regul_lambda is not None:
            self.atlas_loss.edge_regul_lambda = gamma * self.atlas_loss.edge_regul_lambda
        if self.atlas_loss.lambda_laplacian is not None:
            self.atlas_loss.lambda_laplacian = gamma * self.atlas_loss.lambda_laplacian

    def unused_parameters(self):
        """Parameters that can never receive a gradient: the encoders' ImageNet classifier heads (``fc`` is part of the
        checkpoint layout but the feature extractor never calls it, resnet.py:184-186).  ``dp.GradientBuckets(exclude=...)``
        leaves them out of the all-reduce; they keep ``grad = None`` exactly as in the reference."""
        nets = [self.base_net] + ([self.atlas_base_net] if self.atlas_separate_encoder else [])
        return [p for net in nets for p in net.fc.parameters()]

    def _device(self):
        return next(self.base_net.parameters()).device

    def _to_device(self, sample, dev):
        """The reference relies on DataParallel.scatter to move the GT tensors; do it here."""
        moved = dict(sample)
        for key in (TransQueries.images, TransQueries.verts3d, TransQueries.joints3d, TransQueries.objpoints3d):
            val = moved.get(key)
            if torch.is_tensor(val) and val.device != dev:
                moved[key] = val.to(dev, non_blocking=True)
        return moved

    def forward(self, sample, no_loss=False, return_features=False, force_objects=False):
        if force_objects and TransQueries.objpoints3d not in sample:
            sample[TransQueries.objpoints3d] = None
        dev = self._device()
        ops.require_rocm(dev)
        sample = self._to_device(sample, dev)
        total_loss, results, losses = None, {}, {}
        image = sample[TransQueries.images]
        features, _ = self.base_net(image)
        if self.atlas_separate_encoder:
            atlas_infeatures, _ = self.atlas_base_net(image)
            if return_features:
                results["atlas_features"] = atlas_infeatures
        if return_features:
            results["img_features"] = features
        has_hand_gt = TransQueries.joints3d in sample or TransQueries.verts3d in sample
        mano_results = None
        if has_hand_gt and BaseQueries.sides in sample and self.mano_lambdas:
            mano_results = self.mano_branch(features, sides=sample[BaseQueries.sides],
                                            root_palm=(sample.get("root") == "palm"), use_stereoshape=False)
            if not no_loss:
                mano_total_loss, mano_losses = self.mano_loss.compute_loss(mano_results, sample)
                total_loss = mano_total_loss  # alias: later in-place adds are visible in losses["mano_total_loss"]
                losses.update(mano_losses)
            results.update(mano_results)
        predict_atlas = TransQueries.objpoints3d in sample and (self.atlas_lambda or self.atlas_final_lambda)
        if predict_atlas:
            if self.atlas_mesh:
                atlas_features = self.atlas_adapter(features) if self.adapt_atlas_decoder else features
                atlas_results = self.atlas_branch.forward_inference(
                    atlas_features, separate_encoder_features=atlas_infeatures if self.atlas_separate_encoder else None)
            else:
                atlas_results = self.atlas_branch(features)
            if self.need_collisions:
                attr_loss, penetr_loss, contact_infos, contact_metrics = compute_contact_loss(
                    mano_results["verts"], self.mano_branch.faces, atlas_results["objpoints3d"],
                    self.atlas_branch.test_faces_dev, contact_thresh=self.contact_thresh,
                    contact_mode=self.contact_mode, collision_thresh=self.collision_thresh,
                    collision_mode=self.collision_mode, contact_target=self.contact_target,
                    contact_zones=self.contact_zones, obj_patches=self.atlas_branch.patches)
                if not no_loss:
                    if TransQueries.verts3d in sample and TransQueries.objpoints3d in sample:
                        dist_h2o_gt = ops.pairmin(sample[TransQueries.verts3d], sample[TransQueries.objpoints3d],
                                                  want_y=False)[0]
                        contact_ious, contact_auc = meshiou(dist_h2o_gt, contact_infos["min_dists"])
                        contact_infos["batch_ious"] = contact_ious
                        losses["contact_auc"] = contact_auc
                    contact_loss = ops.weighted_terms([(self.contact_lambda, attr_loss), (self.collision_lambda, penetr_loss)],
                                                      attr_loss.shape)
                    total_loss += contact_loss
                    losses["penetration_loss"] = penetr_loss
                    losses["attraction_loss"] = attr_loss
                    losses["contact_loss"] = contact_loss
                    losses.update(contact_metrics)
                results["contact_info"] = contact_infos
            results.update(atlas_results)
            if not no_loss:
                atlas_total_loss, atlas_losses = self.atlas_loss.compute_loss(atlas_results, sample)
                if total_loss is None:
                    total_loss = atlas_total_loss
                else:
                    total_loss += atlas_total_loss
                losses.update(atlas_losses)
        losses["total_loss"] = total_loss
        return total_loss, results, losses
