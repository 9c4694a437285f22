"""Oracle: MANO linear-blend-skinning layer, ManoBranch glue and ManoLoss.

The layer itself is NOT in the reference: ``manobranch.py:6`` imports it from
the external package ``manopth`` (github.com/hassony2/manopth, no pinned
version; call sites ``manobranch.py:92-105,170-182``), which needs the
licence-gated MANO pickles.  **Parity unpinned.**  ``mano_lbs`` restates the
published MANO/SMPL algorithm (Romero et al. 2017 / Loper et al. 2015) in the
order manopth evaluates it:

  PCA -> axis-angle -> quaternion Rodrigues -> shape & pose blend shapes ->
  joint regression -> 16-joint kinematic chain -> rest-pose removal ->
  linear-blend skinning -> 21 joints (16 + 5 fingertip vertices, re-ordered) ->
  centring on ``center_idx`` -> metres to millimetres.

``mano_branch`` / ``mano_loss`` follow ``manobranch.py:115-218`` and ``:251-324``.
dtype-generic (fp64 for finite-difference checks).
"""
import torch
import torch.nn.functional as F

JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
LEVELS = ([1, 4, 7, 10, 13], [2, 5, 8, 11, 14], [3, 6, 9, 12, 15])


def axisang_to_rotmat(aa):
    """aa [n,3] -> R [n,3,3] via unit quaternion; theta = ||aa + 1e-8|| (manopth rodrigues_layer)."""
    theta = torch.norm(aa + 1e-8, p=2, dim=1, keepdim=True)
    axis = aa / theta
    half = theta * 0.5
    quat = torch.cat([torch.cos(half), torch.sin(half) * axis], 1)
    quat = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    R = torch.stack(
        [
            w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2,
        ],
        1,
    )
    return R.view(-1, 3, 3)


def pack_to_torch(pack, dtype=torch.float32):
    out = {}
    for k, v in pack.items():
        if k in ("faces", "parents", "tips", "palm_ids"):
            out[k] = torch.as_tensor(v).long()
        elif k == "side":
            out[k] = v
        else:
            out[k] = torch.as_tensor(v).to(dtype)
    return out


def mano_lbs(pk, pose, betas=None, ncomps=30, center_idx=0, root_palm=False, use_pca=True):
    """pose [b,3+ncomps] (axis-angle root + PCA coeffs) , betas [b,10]|None ->
    verts [b,778,3] mm, joints [b,21,3] mm.  A 4-D ``pose`` [b,16,3,3] is taken as rotation matrices."""
    b = pose.shape[0]
    dtype = pose.dtype
    if pose.dim() == 4:  # ManoLayer(use_pca=False) fed rotation matrices (manobranch.py:126-128): used as given
        R = pose
    else:
        if use_pca:
            hand = pk["hands_mean"] + pose[:, 3:3 + ncomps] @ pk["hands_components"][:ncomps]
        else:
            hand = pk["hands_mean"] + pose[:, 3:48]
        full_pose = torch.cat([pose[:, :3], hand], 1)  # [b,48]
        R = axisang_to_rotmat(full_pose.reshape(b * 16, 3)).view(b, 16, 3, 3)
    eye = torch.eye(3, dtype=dtype)
    pose_map = (R[:, 1:] - eye).reshape(b, 135)
    if betas is None:
        betas = torch.zeros(b, 10, dtype=dtype)
    v_shaped = pk["v_template"].unsqueeze(0) + torch.einsum("vck,bk->bvc", pk["shapedirs"], betas)
    J = torch.einsum("jv,bvc->bjc", pk["J_regressor"], v_shaped)  # [b,16,3]
    v_posed = v_shaped + torch.einsum("vck,bk->bvc", pk["posedirs"], pose_map)
    # kinematic chain, root -> tips
    parents = pk["parents"].tolist()
    G_R = [None] * 16
    G_t = [None] * 16
    G_R[0] = R[:, 0]
    G_t[0] = J[:, 0]
    for level in LEVELS:
        for i in level:
            p = parents[i]
            G_R[i] = G_R[p] @ R[:, i]
            G_t[i] = (G_R[p] @ (J[:, i] - J[:, p]).unsqueeze(2)).squeeze(2) + G_t[p]
    GR = torch.stack(G_R, 1)  # [b,16,3,3]
    Gt = torch.stack(G_t, 1)  # [b,16,3]
    # remove the rest pose: t' = t - R J
    t_rel = Gt - (GR @ J.unsqueeze(3)).squeeze(3)
    W = pk["weights"]  # [778,16]
    TR = torch.einsum("vj,bjrc->bvrc", W, GR)
    Tt = torch.einsum("vj,bjr->bvr", W, t_rel)
    verts = (TR @ v_posed.unsqueeze(3)).squeeze(3) + Tt
    jtr = Gt
    tips = verts[:, pk["tips"]]
    if root_palm:
        palm = (verts[:, pk["palm_ids"][0]] + verts[:, pk["palm_ids"][1]]).unsqueeze(1) / 2
        jtr = torch.cat([palm, jtr[:, 1:]], 1)
    jtr = torch.cat([jtr, tips], 1)[:, JOINT_REORDER]
    if center_idx is not None:
        c = jtr[:, center_idx].unsqueeze(1)
        jtr = jtr - c
        verts = verts - c
    return verts * 1000, jtr * 1000


def mano_branch(params, features, sides, packs, ncomps=30, center_idx=0, use_shape=False,
                use_pca=True, root_palm=False):
    """ManoBranch.forward (manobranch.py:115-218).  ``params`` = state-dict-named tensors
    (``base_layer.{0,2}.*``, ``pose_reg.*``, ``shape_reg.0.*``); ``packs`` = {'right','left'} torch packs."""
    h = features
    k = 0
    while "base_layer.%d.weight" % k in params:
        h = F.relu(F.linear(h, params["base_layer.%d.weight" % k], params["base_layer.%d.bias" % k]))
        k += 2
    pose = F.linear(h, params["pose_reg.weight"], params["pose_reg.bias"])
    mano_pose = pose if use_pca else pose.reshape(pose.shape[0], 16, 3, 3)  # manobranch.py:126-130
    shape = F.linear(h, params["shape_reg.0.weight"], params["shape_reg.0.bias"]) if use_shape else None
    B = features.shape[0]
    is_right = torch.tensor([s == "right" for s in sides][:B], dtype=torch.bool)
    verts = features.new_empty((B, 778, 3))
    joints = features.new_empty((B, 21, 3))
    for side, mask in (("right", is_right), ("left", ~is_right)):
        if int(mask.sum()) == 0:
            continue
        v, j = mano_lbs(packs[side], mano_pose[mask], shape[mask] if shape is not None else None,
                        ncomps=ncomps, center_idx=center_idx, root_palm=root_palm, use_pca=use_pca)
        verts[mask] = v
        joints[mask] = j
    return {"verts": verts, "joints": joints, "shape": shape, "pose": pose}


def mano_loss(preds, target_verts, target_joints, lambda_verts=None, lambda_joints3d=None,
              lambda_shape=None, lambda_pose_reg=None):
    """ManoLoss.compute_loss (manobranch.py:251-324) -> (final_loss [1], dict)."""
    final = torch.zeros(1, dtype=preds["verts"].dtype)
    out = {}
    if target_verts is not None and lambda_verts:
        lv = F.mse_loss(preds["verts"], target_verts)
        final = final + lambda_verts * lv
    else:
        lv = None
    out["mano_verts3d"] = lv
    if target_joints is not None and lambda_joints3d:
        lj = F.mse_loss(preds["joints"], target_joints)
        final = final + lambda_joints3d * lj
        out["mano_joints3d"] = lj
    if lambda_shape:
        ls = F.mse_loss(preds["shape"], torch.zeros_like(preds["shape"]))
        final = final + lambda_shape * ls
    else:
        ls = None
    out["mano_shape"] = ls
    if lambda_pose_reg:
        lp = F.mse_loss(preds["pose"][:, 3:], torch.zeros_like(preds["pose"][:, 3:]))
        final = final + lambda_pose_reg * lp
        out["pose_reg"] = lp
    out["mano_pca"] = None
    out["mano_total_loss"] = final
    return final, out
