"""TEST-ONLY fake backend: the ``obman_train_amd.ops`` API served by the CPU oracle, so the host
logic (HandNet assembly, loss dicts, quirks) can be exercised by ``-m "not gpu"`` tests.  Installed
with ``monkeypatch`` by tests; the product never imports this module."""
import torch

from oracle import chamfer as ocham
from oracle import contact as ocontact
from oracle import mano as omano
from obman_train_amd.mano_params import synthetic_mano

_PACKS = {}


def _pack(side, dtype):
    key = (side, dtype)
    if key not in _PACKS:
        _PACKS[key] = omano.pack_to_torch(synthetic_mano(side), dtype)
    return _PACKS[key]


def pairmin(x, y, want_x=True, want_y=True):
    mx, ix, my, iy = ocham.pairmin_direct(x, y)
    return (mx if want_x else None, ix.int() if want_x else None, my if want_y else None, iy.int() if want_y else None)


def chamfer(preds, gts):
    return ocham.chamfer_direct(preds, gts)


def mano_lbs(pose, betas, blob_right, blob_left=None, side=None, ncomps=30, use_pca=True, center_idx=0, root_palm=False):
    B = pose.shape[0]
    verts = pose.new_empty((B, 778, 3))
    joints = pose.new_empty((B, 21, 3))
    flags = torch.zeros(B, dtype=torch.bool) if side is None else side.bool()
    for name, mask in (("right", ~flags), ("left", flags)):
        if int(mask.sum()) == 0:
            continue
        v, j = omano.mano_lbs(_pack(name, pose.dtype), pose[mask], betas[mask] if betas is not None else None,
                              ncomps=ncomps, center_idx=center_idx, root_palm=root_palm, use_pca=use_pca)
        verts[mask], joints[mask] = v, j
    return verts, joints


def mesh_contains_hits(points, verts, faces, patches=1):
    exterior = None
    for grp in faces.long().chunk(max(int(patches), 1), 0):
        ext = ocontact.mesh_contains_points(points.detach(), verts.detach()[:, grp])
        exterior = ext if exterior is None else (exterior & ext)
    return (~exterior).int()


def contact_tail(hand, obj, idx21, mins21, hits, zone_ids, zone_off, n_zones, zone_mode, contact_mode,
                 contact_thresh, collision_mode, collision_thresh, target):
    exterior = (hits & 1) == 0
    closest = ocontact.batch_index_select(obj, 1, idx21.long())
    if target == 0:
        delta = closest - hand
    elif target == 1:
        delta = closest - hand.detach()
    else:
        delta = closest.detach() - hand
    anchor = torch.norm(delta, 2, 2)
    sq = (delta ** 2).sum(2)
    names = {0: "dist_sq", 1: "dist", 2: "dist_tanh"}
    cvals = ocontact._penalty(names[contact_mode], contact_thresh, anchor, sq, "contact_mode")
    kvals = ocontact._penalty(names[collision_mode], collision_thresh, anchor, sq, "collision_mode")
    if contact_mode == 0:
        below = mins21 < contact_thresh ** 2
    elif contact_mode == 1:
        below = mins21 < contact_thresh
    else:
        below = torch.ones_like(mins21, dtype=torch.bool)
    allow = torch.ones_like(below)
    if zone_mode == 1:
        allow = torch.zeros_like(below)
        allow[:, zone_ids[: int(zone_off[1])].long()] = True
    elif zone_mode == 2:
        allow = torch.zeros_like(below)
        rows = torch.arange(below.shape[0])
        for z in range(n_zones):
            ids = zone_ids[int(zone_off[z]): int(zone_off[z + 1])].long()
            allow[rows, ids[mins21[:, ids].min(1)[1]]] = True
    missed = below & exterior & allow
    penetr = ~exterior
    m, p = missed.float(), penetr.float()
    missed_loss = (m * cvals).sum() / m.sum().clamp(min=1)
    penetr_loss = (p * kvals).sum() / p.sum().clamp(min=1)
    depth = anchor.detach() * p
    out = torch.stack([missed_loss.detach(), penetr_loss.detach(), depth.max(1)[0].mean(), depth.mean(1).mean(),
                       m.sum(), p.sum(), torch.zeros(()), torch.zeros(())])
    return missed_loss, penetr_loss, out, missed.to(torch.uint8), penetr.to(torch.uint8), closest.detach()


def pointgen_decode(decoder, features, grid):
    """Reference formulation: materialise the [B,3+C,N] concat and run PointGenCon (oracle/atlas.py)."""
    from oracle import atlas as oatlas

    B, N = features.shape[0], grid.shape[-2]
    params = {"decoder." + k: v for k, v in list(decoder.named_parameters()) + list(decoder.named_buffers())}
    g3 = grid.transpose(2, 1) if grid.dim() == 3 else grid.t().unsqueeze(0).expand(B, -1, -1)
    x = torch.cat((g3, features.unsqueeze(2).expand(-1, -1, N)), 1)
    out = oatlas.pointgen(params, x, training=decoder.training, out_factor=decoder.out_factor,
                          momentum=decoder.bn1.momentum, eps=decoder.bn1.eps).transpose(2, 1)
    if decoder.training:
        with torch.no_grad():
            for bn in (decoder.bn1, decoder.bn2, decoder.bn3):
                bn.num_batches_tracked += 1
    return out


def edge_loss(verts, faces):
    from oracle import atlas as oatlas

    return oatlas.edge_loss(verts, faces.long())


def laplacian_loss(verts, row_ptr, col, val):
    n = row_ptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n), (row_ptr[1:] - row_ptr[:-1]).long())
    L = torch.zeros(n, n, dtype=verts.dtype)
    L[rows, col.long()] = val.to(verts.dtype)
    Lx = torch.einsum('ij,bjc->bic', L, verts)
    return torch.norm(Lx.reshape(-1, 3), p=2, dim=1).mean()


def affine_points(verts, scale=None, trans=None):
    """atlasbranch.py:133-138, the reference's own operations."""
    pts = verts if scale is None else scale.unsqueeze(1) * verts
    return pts if trans is None else pts + trans.unsqueeze(1)


def mse_terms(pairs):
    import torch.nn.functional as F

    return [F.mse_loss(p, torch.zeros_like(p) if t is None else t) for p, t in pairs]


def gt_object_stats(gt):
    centroids = gt.mean(1)
    centred = gt - centroids.unsqueeze(1)
    return centroids, centred, torch.norm(centred, 2, 2).max(1)[0].unsqueeze(1)


def install(monkeypatch):
    from obman_train_amd import ops

    for name in ("pairmin", "chamfer", "mano_lbs", "mesh_contains_hits", "contact_tail", "pointgen_decode", "edge_loss", "laplacian_loss", "affine_points", "mse_terms",
                 "gt_object_stats"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(ops, "require_rocm", lambda device: None)
