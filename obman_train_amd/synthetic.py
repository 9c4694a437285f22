"""Synthetic ObMan-shaped batches (no dataset / network here), following SURVEY §8d:
images U(-0.5,0.5) [B,3,256,256] (``handataset.py:391-405`` normalisation); GT hand vertices = hand
template (mm, root-centred) + N(0,5 mm); GT joints N(0,40 mm); GT object cloud = points on a random
ellipsoid (semi-axes U(20,80) mm) offset ~60 mm from the palm; all sides "left" (the CLI default,
``datasetopts.py:28-33``); ``"root": "wrist"``.  Generated directly in HBM with a per-rank seed."""
import torch

from .contactzones import hand_template
from .queries import BaseQueries, TransQueries

CONFIGS = {
    # BASELINE.json configs[0]/[1]: flags --atlas_mesh --mano_use_pca --atlas_lambda 0.167 + CLI defaults
    "c2": dict(
        resnet_version=18, atlas_mesh=True, mano_use_pca=True, mano_comps=30, mano_neurons=[1024, 256],
        mano_root="synthetic",  # no MANO files here (licence-gated): the seeded stand-in model, asked for explicitly
        mano_center_idx=0, atlas_lambda=0.167, atlas_final_lambda=0.167, atlas_trans_weight=0.167,
        atlas_scale_weight=0.167, mano_lambda_verts=0.167, mano_lambda_joints3d=0.167, mano_lambda_pose_reg=0.167,
        contact_thresh=10, collision_thresh=20, contact_mode="dist_tanh", collision_mode="dist_tanh",
        contact_zones="zones",
    ),
}
# configs[2]: full ObMan recipe (README.md:133) + contact losses, 25 patches
CONFIGS["c3"] = dict(
    CONFIGS["c2"], mano_use_shape=True, mano_lambda_shape=0.167, atlas_predict_trans=True, atlas_predict_scale=True,
    contact_lambda=1.0, collision_lambda=1.0, atlas_patches=25,
)
CONFIGS["c3p1"] = dict(CONFIGS["c3"], atlas_patches=1)  # contact config on the reference's single sphere
# configs[4]: high-resolution object decoder, 25 patches x 2562 points (icosphere subdivision 4), same losses as configs[2]
CONFIGS["c5"] = dict(CONFIGS["c3"], atlas_ico_divisions=4)


def make_batch(batch, device, seed=0, n_obj=600, image_size=256, dtype=torch.float32):
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + seed)
    r = lambda *s: torch.rand(*s, device=device, generator=gen, dtype=dtype)
    n = lambda *s: torch.randn(*s, device=device, generator=gen, dtype=dtype)
    tmpl = torch.from_numpy(hand_template()[0]).to(device=device, dtype=dtype) * 1000.0
    tmpl = tmpl - tmpl.mean(0, keepdim=True)
    verts = tmpl.unsqueeze(0) + 5.0 * n(batch, 778, 3)
    joints = 40.0 * n(batch, 21, 3)
    u = n(batch, n_obj, 3)
    u = u / u.norm(dim=2, keepdim=True)
    axes = 20.0 + 60.0 * r(batch, 1, 3)
    centre = 30.0 * n(batch, 1, 3) + torch.tensor([0.0, -60.0, 0.0], device=device, dtype=dtype)
    return {
        TransQueries.images: r(batch, 3, image_size, image_size) - 0.5,
        TransQueries.verts3d: verts,
        TransQueries.joints3d: joints,
        TransQueries.objpoints3d: u * axes + centre,
        BaseQueries.sides: ["left"] * batch,
        "root": "wrist",
    }
