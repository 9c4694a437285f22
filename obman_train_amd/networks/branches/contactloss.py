"""Contact / penetration losses, HIP-backed.

Mirror of ``mano_train/networks/branches/contactloss.py`` (reference): ``batch_index_select``
(:11-19), ``thresh_ious`` / ``meshiou`` (:22-47), ``masked_mean_loss`` (:50-57),
``batch_pairwise_dist`` (:60-79) and ``compute_contact_loss`` (:149-308) with the same arguments,
return tuple and dict keys.  ``compute_contact_loss`` is three launches (pair-min with arg-min,
inside test, fused tail) and never syncs with the host: the reference's ``if valid_vals > 0`` and
``.cpu().numpy()`` (meshiou) are device-side here, so ``contact_auc`` is a 0-d device tensor rather
than a numpy float (``.item()`` works on both, epochpass3d.py:111-117).
"""
import numpy as np
import torch

from obman_train_amd import ops
from obman_train_amd.contactzones import TIP_IDXS, load_contacts
from obman_train_amd.networks.branches.contactutils import mesh_exterior

_ZONE_CACHE = {}


def batch_index_select(inp, dim, index):
    shape = [inp.shape[0]] + [1 if i != dim else -1 for i in range(1, inp.dim())]
    expanse = list(inp.shape)
    expanse[0] = -1
    expanse[dim] = -1
    return torch.gather(inp, dim, index.long().view(shape).expand(expanse))


def thresh_ious(gt_dists, pred_dists, thresh):
    g, p = gt_dists <= thresh, pred_dists <= thresh
    inter, union = (g & p).sum(1).float(), (g | p).sum(1).float()
    return torch.where(union != 0, inter / union.clamp(min=1), torch.zeros_like(union))


_THRESH_CACHE = {}


def _thresholds(threshs, dtype, device):
    """The threshold vector on the device, uploaded once: a host->device copy per call would break hipGraph capture of the step."""
    key = (tuple(threshs), dtype, str(device))
    t = _THRESH_CACHE.get(key)
    if t is None:
        t = _THRESH_CACHE[key] = torch.tensor(list(threshs), dtype=dtype, device=device).view(-1, 1, 1)
    return t


def meshiou(gt_dists, pred_dists, threshs=(1, 2, 3, 4, 5, 6, 7, 8, 9, 10)):
    th = _thresholds(threshs, gt_dists.dtype, gt_dists.device)
    g, p = gt_dists.unsqueeze(0) <= th, pred_dists.unsqueeze(0) <= th
    inter, union = (g & p).sum(2).float(), (g | p).sum(2).float()
    ious = torch.where(union != 0, inter / union.clamp(min=1), torch.zeros_like(union))  # [T,B]
    auc = torch.trapezoid(ious, x=th.view(-1), dim=0).mean()
    return ious.mean(1), auc


def masked_mean_loss(dists, mask):
    m = mask.float()
    return (m * dists).sum() / m.sum().clamp(min=1)  # empty mask -> 0, without the reference's host sync


def batch_pairwise_dist(x, y, use_cuda=True):
    raise NotImplementedError("the N x M matrix is never materialised on the HIP path; use obman_train_amd.ops.pairmin")


def _zone_tables(contact_zones, device):
    key = (contact_zones, str(device))
    if key not in _ZONE_CACHE:
        if contact_zones == "tips":
            lists = [list(TIP_IDXS)]
        else:
            _, zones = load_contacts("assets/contact_zones.pkl")
            lists = [zones[k] for k in sorted(zones)]
        ids = torch.tensor([i for l in lists for i in l], dtype=torch.int32, device=device)
        off = torch.tensor(np.cumsum([0] + [len(l) for l in lists]), dtype=torch.int32, device=device)
        _ZONE_CACHE[key] = (ids, off, len(lists))
    return _ZONE_CACHE[key]


def _faces_on(obj_faces, device):
    if torch.is_tensor(obj_faces):
        return obj_faces.to(device=device, dtype=torch.int32)
    return torch.as_tensor(np.ascontiguousarray(obj_faces, dtype=np.int32), device=device)


def compute_contact_loss(hand_verts_pt, hand_faces, obj_verts_pt, obj_faces, contact_thresh=5, contact_mode="dist_sq",
                         collision_thresh=10, collision_mode="dist_sq", contact_target="all", contact_sym=False,
                         contact_zones="all", obj_patches=1):
    """``obj_patches`` (extension, default = the reference's single closed mesh): the object mesh is that many closed patch
    surfaces (equal consecutive face groups); penetration = inside any patch."""
    if contact_target not in ops.TARGETS:
        raise ValueError("contact_target {} not in [all|obj|hand]".format(contact_target))
    if contact_mode not in ops.MODES:
        raise ValueError("contact_mode {} not in [dist_sq|dist|dist_tanh]".format(contact_mode))
    if collision_mode not in ops.MODES:
        raise ValueError("collision_mode {} not in [dist_sq|dist|dist_tanh]".format(collision_mode))
    if contact_zones not in ("tips", "zones", "all"):
        raise ValueError("contact_zones {} not in [tips|zones|all]".format(contact_zones))
    if contact_sym:
        raise NotImplementedError("contact_sym is never enabled by HandNet (handnet.py:336-347)")
    dev = hand_verts_pt.device
    mins21, idx21, _, _ = ops.pairmin(hand_verts_pt.detach(), obj_verts_pt.detach(), want_y=False)
    exterior, hits = mesh_exterior(hand_verts_pt, obj_verts_pt, _faces_on(obj_faces, dev), patches=obj_patches)
    if contact_zones == "all":
        ids, off, nz, zmode = None, None, 0, 0
    else:
        ids, off, nz = _zone_tables(contact_zones, dev)
        zmode = 1 if contact_zones == "tips" else 2
    missed, penetr, out, attr, rep, closest = ops.contact_tail(
        hand_verts_pt, obj_verts_pt, idx21, mins21, hits, ids, off, nz, zmode, ops.MODES[contact_mode],
        contact_thresh, ops.MODES[collision_mode], collision_thresh, ops.TARGETS[contact_target])
    contact_info = {
        # dtype quirk of the reference (App. C #5): uint8 for dist_tanh, bool otherwise
        "attraction_masks": attr if contact_mode == "dist_tanh" else attr.bool(),
        "repulsion_masks": rep.bool(),
        "contact_points": closest,
        "min_dists": mins21,
    }
    metrics = {"max_penetr": out[2], "mean_penetr": out[3]}
    return missed, penetr, contact_info, metrics
