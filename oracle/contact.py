"""Oracle: ray-parity inside test and the contact / penetration loss.

Follows ``mano_train/networks/branches/contactutils.py:62-159``
(batch_mesh_contains_points) and ``contactloss.py:11-57,149-308``
(batch_index_select, meshiou, masked_mean_loss, compute_contact_loss).
Same tensor formulation as the reference (all point x triangle pairs
materialised), so it costs what the reference costs.  Quirks kept on purpose
(SURVEY App. C): 'dist' mode compares *squared* distances with the unsquared
threshold; 'dist_tanh' yields a uint8 attraction mask; empty masks give a fresh
``tensor([0.])``; zones keep only the closest vertex of each zone per sample.
"""
import numpy as np
import torch

from .chamfer import batch_pairwise_dist

RAY_DIRECTION = (0.4395064455, 0.617598629942, 0.652231566745)  # contactutils.py:65
TOL = 0.0000001  # contactutils.py:78
TIP_IDXS = [745, 317, 444, 556, 673]  # contactloss.py:258


def _rowdot(a, b):
    """Row-wise 3-vector dot through bmm, as the reference does (contactutils.py:97-100,121-126)."""
    n = a.shape[0]
    return torch.bmm(a.reshape(n, 1, 3), b.reshape(n, 3, 1)).reshape(n)


def mesh_contains_points(ray_origins, obj_triangles, direction=None, return_counts=False):
    """origins [B,P,3], triangles [B,T,3,3] -> exterior [B,P] bool (even hit count).
    ``return_counts`` (tests only; not a reference argument): the crossing count [B,P] itself, before the parity.

    Moeller-Trumbore for every (point, triangle) pair along one shared direction."""
    B, T = obj_triangles.shape[:2]
    P = ray_origins.shape[1]
    dtype = ray_origins.dtype
    if direction is None:
        direction = torch.tensor(RAY_DIRECTION, dtype=torch.float32).to(dtype)
    a, b, c = obj_triangles[:, :, 0], obj_triangles[:, :, 1], obj_triangles[:, :, 2]
    e1 = b - a
    e2 = c - a
    dirs = direction.view(1, 1, 3).expand(B, T, 3)
    pvec = torch.cross(dirs, e2, dim=2)
    det = _rowdot(e1.reshape(B * T, 3), pvec.reshape(B * T, 3)).view(B, T)
    parallel = det.abs() < TOL
    inv_det = 1 / (det + 0.1 * TOL)
    # tile triangle quantities once per ray: index [b, p*T + t]
    a_r = a.repeat(1, P, 1)
    e1_r = e1.repeat(1, P, 1)
    e2_r = e2.repeat(1, P, 1)
    pvec_r = pvec.repeat(1, P, 1)
    inv_r = inv_det.repeat(1, P)
    orig_r = ray_origins.view(B, P, 1, 3).repeat(1, 1, T, 1).view(B, P * T, 3)
    tvec = orig_r - a_r
    n = B * P * T
    u = _rowdot(tvec.reshape(n, 3), pvec_r.reshape(n, 3)).view(B, P * T) * inv_r
    u_ok = (u > 0) * (u < 1)
    qvec = torch.cross(tvec, e1_r, dim=2)
    dirs_r = dirs.repeat(1, P, 1)
    v = _rowdot(dirs_r.reshape(n, 3), qvec.reshape(n, 3)).view(B, P * T) * inv_r
    v_ok = (v > 0) * (u + v < 1)
    t = _rowdot(e2_r.reshape(n, 3), qvec.reshape(n, 3)).view(B, P * T) * inv_r
    t_ok = t >= TOL
    hit = v_ok * u_ok * parallel.repeat(1, P).logical_not() * t_ok
    hits = hit.view(B, P, T).sum(2)
    if return_counts:
        return hits
    return hits % 2 == 0


def batch_index_select(inp, dim, index):
    """gather rows ``index[b, :]`` along ``dim`` (contactloss.py:11-19); inp [B,N,3], index [B,V]."""
    assert dim == 1
    return torch.gather(inp, 1, index.unsqueeze(2).expand(-1, -1, inp.shape[2]))


def masked_mean_loss(vals, mask):
    """Batch-global masked mean; fresh zero when the mask is empty (contactloss.py:50-57)."""
    m = mask.float()
    cnt = m.sum()
    if cnt > 0:
        return (m * vals).sum() / cnt
    return torch.zeros(1, dtype=vals.dtype)


def thresh_ious(gt_dists, pred_dists, thresh):
    g = gt_dists <= thresh
    p = pred_dists <= thresh
    inter = (g * p).sum(1).float()
    union = (g | p).sum(1).float()
    iou = torch.zeros_like(union)
    nz = union != 0
    iou[nz] = inter[nz] / union[nz]
    return iou


def meshiou(gt_dists, pred_dists, threshs=(1, 2, 3, 4, 5, 6, 7, 8, 9, 10)):
    """contactloss.py:36-47 -> (batch_ious [len(threshs)], auc float)."""
    stack = torch.stack([thresh_ious(gt_dists, pred_dists, t) for t in threshs])
    trapz = getattr(np, "trapezoid", None) or np.trapz
    auc = np.mean(trapz(stack.cpu().numpy(), axis=0, x=list(threshs)))
    return stack.mean(1), auc


def _penalty(mode, thresh, anchor_dists, sq_vals, what):
    if mode == "dist_sq":
        return sq_vals
    if mode == "dist":
        return anchor_dists
    if mode == "dist_tanh":
        return thresh * torch.tanh(anchor_dists / thresh)
    raise ValueError("{} {} not in [dist_sq|dist|dist_tanh]".format(what, mode))


def compute_contact_loss(
    hand_verts, hand_faces, obj_verts, obj_faces, zones=None,
    contact_thresh=5, contact_mode="dist_sq", collision_thresh=10, collision_mode="dist_sq",
    contact_target="all", contact_sym=False, contact_zones="all", obj_patches=1,
):
    """contactloss.py:149-308.  ``zones`` = {zone: [vertex ids]} (the reference loads it from
    assets/contact_zones.pkl at :262-265).  Returns (missed_loss, penetr_loss, contact_info, metrics).

    ``obj_patches`` > 1 (the build's multi-patch extension, no reference counterpart): the faces are that many equal
    consecutive groups, each a closed surface; the reference's inside test (:169-171) is applied per patch and a hand vertex
    is interior when it is inside ANY patch."""
    dists = batch_pairwise_dist(hand_verts, obj_verts)  # [B,V,N]
    mins12, _ = dists.min(1)       # per obj vertex  [B,N]
    mins21, idx21 = dists.min(2)   # per hand vertex [B,V]
    faces_t = torch.as_tensor(np.asarray(obj_faces).astype(np.int64))
    if obj_patches > 1:
        exterior = None
        for grp in faces_t.chunk(obj_patches, 0):
            ext = mesh_contains_points(hand_verts.detach(), obj_verts[:, grp].detach())
            exterior = ext if exterior is None else (exterior & ext)
    else:
        triangles = obj_verts[:, faces_t]
        exterior = mesh_contains_points(hand_verts.detach(), triangles.detach())
    penetr_mask = ~exterior
    closest = batch_index_select(obj_verts, 1, idx21)
    if contact_target == "all":
        delta = closest - hand_verts
    elif contact_target == "obj":
        delta = closest - hand_verts.detach()
    elif contact_target == "hand":
        delta = closest.detach() - hand_verts
    else:
        raise ValueError("contact_target {} not in [all|obj|hand]".format(contact_target))
    anchor_dists = torch.norm(delta, 2, 2)
    sq_vals = (delta ** 2).sum(2)
    contact_vals = _penalty(contact_mode, contact_thresh, anchor_dists, sq_vals, "contact_mode")
    if contact_mode == "dist_sq":
        below = mins21 < (contact_thresh ** 2)
    elif contact_mode == "dist":
        below = mins21 < contact_thresh  # sic: squared distance vs unsquared threshold
    else:
        below = torch.ones_like(mins21).byte()  # sic: uint8
    collision_vals = _penalty(collision_mode, collision_thresh, anchor_dists, sq_vals, "collision_mode")

    missed_mask = below & exterior
    if contact_zones == "tips":
        tips = torch.zeros_like(missed_mask)
        tips[:, TIP_IDXS] = 1
        missed_mask = missed_mask & tips
    elif contact_zones == "zones":
        matching = torch.zeros_like(missed_mask)
        rows = torch.arange(missed_mask.shape[0])
        for _, ids in zones.items():
            ids_t = torch.as_tensor(list(ids), dtype=torch.long)
            pick = mins21[:, ids_t].min(1)[1]
            matching[rows, ids_t[pick]] = 1
        missed_mask = missed_mask & matching
    elif contact_zones != "all":
        raise ValueError("contact_zones {} not in [tips|zones|all]".format(contact_zones))

    missed_loss = masked_mean_loss(contact_vals, missed_mask)
    penetr_loss = masked_mean_loss(collision_vals, penetr_mask)
    if contact_sym:
        missed_loss = missed_loss + masked_mean_loss(torch.sqrt(mins12), mins12 < contact_thresh)
    depth = anchor_dists.detach() * penetr_mask.float()
    metrics = {"max_penetr": depth.max(1)[0].mean(), "mean_penetr": depth.mean(1).mean()}
    info = {
        "attraction_masks": missed_mask, "repulsion_masks": penetr_mask,
        "contact_points": closest, "min_dists": mins21,
    }
    return missed_loss, penetr_loss, info, metrics
