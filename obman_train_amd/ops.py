"""Host-side operators over the HIP C-ABI: thin ``torch.autograd.Function`` wrappers.

PyTorch is plumbing here (device memory, streams, autograd graph); all arithmetic of these ops runs
in ``csrc/*.hip``.  Every op requires fp32 ROCm tensors and raises otherwise - no CPU path.
"""
import ctypes as _ct

import torch

from . import _lib


def require_rocm(device):
    """The product path is HIP-only: refuse to run anywhere else instead of silently falling back."""
    if device.type != "cuda":
        raise _lib.ObmanHipError("obman_train_amd needs a ROCm device (model.cuda()); there is no CPU fallback")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream's handle.  ``torch.cuda.current_stream().cuda_stream`` costs ~11 us of host time per call (device-index
    plumbing in Python) and a step makes ~35 of them: 0.4 ms of an enqueue budget that is within 10 % of the GPU time at configs[2]
    (tools/archive/r06/host_profile.py); the raw accessor is the same query without the wrappers."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.ObmanHipError("%s must be a ROCm device tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _check_pair(x, y):
    if x.dim() != 3 or y.dim() != 3 or x.shape[2] != 3 or y.shape[2] != 3 or x.shape[0] != y.shape[0]:
        raise ValueError("expected x [B,Nx,3], y [B,Ny,3]; got %s %s" % (tuple(x.shape), tuple(y.shape)))
    if x.shape[1] == 0 or y.shape[1] == 0:
        # same failure the reference hits in torch.min over an empty dimension
        raise IndexError("min(): cannot reduce over an empty point set")


def _workspace(B, nx, ny, device):
    nbytes = _lib.lib().obman_pairmin_ws_bytes(B, nx, ny)
    small, large = (nx, ny) if nx < ny else (ny, nx)
    if large < 8192 or B * ((small + 639) // 640) >= 512:
        return None, 0  # the launcher would not split the reference set: skip the allocation
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


class _PairMin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, want_x, want_y):
        x, y = _dev(x, "x"), _dev(y, "y")
        _check_pair(x, y)
        B, nx, ny = x.shape[0], x.shape[1], y.shape[1]
        o = dict(device=x.device)
        min_x = torch.empty((B, nx), dtype=torch.float32, **o) if want_x else None
        idx_x = torch.empty((B, nx), dtype=torch.int32, **o) if want_x else None
        min_y = torch.empty((B, ny), dtype=torch.float32, **o) if want_y else None
        idx_y = torch.empty((B, ny), dtype=torch.int32, **o) if want_y else None
        ws, ws_bytes = _workspace(B, nx, ny, x.device)
        _lib.check(_lib.lib().obman_pairmin_fwd(
            x.data_ptr(), y.data_ptr(), B, nx, ny, _ptr(min_x), _ptr(idx_x), _ptr(min_y), _ptr(idx_y),
            _ptr(ws), ws_bytes, _stream()), "obman_pairmin_fwd")
        ctx.save_for_backward(x, y, idx_x, idx_y)
        ctx.mark_non_differentiable(*[t for t in (idx_x, idx_y) if t is not None])
        return min_x, idx_x, min_y, idx_y

    @staticmethod
    def backward(ctx, g_min_x, _gi, g_min_y, _gj):
        x, y, idx_x, idx_y = ctx.saved_tensors
        B, nx, ny = x.shape[0], x.shape[1], y.shape[1]
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_min_x = g_min_x.contiguous() if g_min_x is not None else None
        g_min_y = g_min_y.contiguous() if g_min_y is not None else None
        grad_x = torch.empty_like(x) if need_x else None
        grad_y = torch.empty_like(y) if need_y else None
        _lib.check(_lib.lib().obman_pairmin_bwd(
            x.data_ptr(), y.data_ptr(), B, nx, ny, _ptr(idx_x), _ptr(idx_y), _ptr(g_min_x), _ptr(g_min_y),
            _ptr(grad_x), _ptr(grad_y), _stream()), "obman_pairmin_bwd")
        return grad_x, grad_y, None, None


def pairmin(x, y, want_x=True, want_y=True):
    """x [B,Nx,3], y [B,Ny,3] -> (min_x [B,Nx], idx_x int32, min_y [B,Ny], idx_y int32).

    min_x[b,i] = min_j |x_i-y_j|^2 (squared, like the reference's batch_pairwise_dist + torch.min,
    contactloss.py:164-166); a direction not wanted returns None."""
    return _PairMin.apply(x, y, bool(want_x), bool(want_y))


_SYNC = {}  # (device index, stream id) -> zero-initialised arrival counters + partial sums of the single-launch Chamfer


def _chamfer_sync(B, n_p, n_g, device):
    """The `sync` buffer of obman_chamfer_fwd (include/obman_hip.h): zeroed ONCE here, left zero by every call, one per stream
    (stream-ordered calls may share it, concurrent ones may not)."""
    nbytes = _lib.lib().obman_chamfer_sync_bytes(B, n_p, n_g)
    if nbytes <= 0:
        return None, 0
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SYNC.get(key)
    if buf is None or buf.numel() < nbytes:
        # a fresh allocation rather than a resize: kernels already queued on this stream may still be using the old one
        buf = _SYNC[key] = torch.zeros(max(nbytes, 4096), dtype=torch.uint8, device=device)
    return buf, buf.numel()


class _Chamfer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, gts):
        preds, gts = _dev(preds, "preds"), _dev(gts, "gts")
        _check_pair(preds, gts)
        B, n_p, n_g = preds.shape[0], preds.shape[1], gts.shape[1]
        o = dict(device=preds.device)
        loss = torch.empty((2, B), dtype=torch.float32, **o)
        mins = torch.empty((B, n_p + n_g), dtype=torch.float32, **o)
        idx = torch.empty((B * (n_p + n_g),), dtype=torch.int32, **o)
        min_pred, min_gt = mins.view(-1)[: B * n_p], mins.view(-1)[B * n_p:]
        idx_pred, idx_gt = idx[: B * n_p], idx[B * n_p:]
        ws, ws_bytes = _workspace(B, n_p, n_g, preds.device)
        sync, sync_bytes = _chamfer_sync(B, n_p, n_g, preds.device)
        _lib.check(_lib.lib().obman_chamfer_fwd(
            preds.data_ptr(), gts.data_ptr(), B, n_p, n_g, loss[0].data_ptr(), loss[1].data_ptr(),
            min_pred.data_ptr(), idx_pred.data_ptr(), min_gt.data_ptr(), idx_gt.data_ptr(),
            _ptr(ws), ws_bytes, _ptr(sync), sync_bytes, _stream()), "obman_chamfer_fwd")
        ctx.save_for_backward(preds, gts, idx)
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g1, g2):
        preds, gts, idx = ctx.saved_tensors
        B, n_p, n_g = preds.shape[0], preds.shape[1], gts.shape[1]
        idx_pred, idx_gt = idx[: B * n_p], idx[B * n_p:]
        g1 = g1.contiguous() if g1 is not None else None
        g2 = g2.contiguous() if g2 is not None else None
        grad_p = torch.empty_like(preds) if ctx.needs_input_grad[0] else None
        grad_g = torch.empty_like(gts) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.lib().obman_chamfer_bwd(
            preds.data_ptr(), gts.data_ptr(), B, n_p, n_g, idx_pred.data_ptr(), idx_gt.data_ptr(),
            _ptr(g1), _ptr(g2), _ptr(grad_p), _ptr(grad_g), _stream()), "obman_chamfer_bwd")
        return grad_p, grad_g


def chamfer(preds, gts):
    """ChamferLoss.forward (atlasutils.py:11-18): -> (loss_1 [B] mean over preds of nearest-gt
    squared distance, loss_2 [B] mean over gts of nearest-pred squared distance)."""
    return _Chamfer.apply(preds, gts)


class _ManoLBS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, betas, blob_right, blob_left, side, ncomps, pose_mode, center_idx, root_palm):
        pose = _dev(pose, "pose")
        B = pose.shape[0]
        in_shape = tuple(pose.shape)
        if pose_mode == 2:
            if in_shape[1:] not in ((16, 3, 3), (144,)):
                raise ValueError("rotation-matrix pose must be [B,16,3,3], got %s" % (in_shape,))
            npose = 144
        else:
            npose = 3 + (ncomps if pose_mode == 1 else 45)
            if pose.dim() != 2 or pose.shape[1] != npose:
                raise ValueError("pose must be [B,%d], got %s" % (npose, in_shape))
        if betas is not None:
            betas = _dev(betas, "betas")
            if tuple(betas.shape) != (B, 10):
                raise ValueError("betas must be [B,10]")
        if side is not None:
            side = _dev(side, "side", torch.int32)
        o = dict(device=pose.device, dtype=torch.float32)
        verts = torch.empty((B, 778, 3), **o)
        joints = torch.empty((B, 21, 3), **o)
        need_bwd = pose.requires_grad or (betas is not None and betas.requires_grad)
        state = torch.empty((B, _lib.lib().obman_mano_state_floats()), **o) if need_bwd else None
        cidx = -1 if center_idx is None else int(center_idx)
        _lib.check(_lib.lib().obman_mano_lbs_fwd(
            blob_right.data_ptr(), _ptr(blob_left), _ptr(side), pose.data_ptr(), _ptr(betas), B, int(ncomps),
            int(pose_mode), cidx, int(bool(root_palm)), verts.data_ptr(), joints.data_ptr(), _ptr(state),
            _stream()), "obman_mano_lbs_fwd")
        ctx.save_for_backward(state, blob_right, blob_left, side)
        ctx.cfg = (B, int(ncomps), int(pose_mode), cidx, int(bool(root_palm)), npose, betas is not None, in_shape)
        return verts, joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        state, blob_right, blob_left, side = ctx.saved_tensors
        B, ncomps, pose_mode, cidx, root_palm, npose, has_betas, in_shape = ctx.cfg
        g_verts = g_verts.contiguous() if g_verts is not None else None
        g_joints = g_joints.contiguous() if g_joints is not None else None
        o = dict(device=state.device, dtype=torch.float32)
        g_pose = torch.empty((B, npose), **o)
        g_betas = torch.empty((B, 10), **o) if (has_betas and ctx.needs_input_grad[1]) else None
        scratch = torch.empty(_lib.lib().obman_mano_bwd_scratch_floats(B), **o)
        _lib.check(_lib.lib().obman_mano_lbs_bwd(
            blob_right.data_ptr(), _ptr(blob_left), _ptr(side), state.data_ptr(), _ptr(g_verts), _ptr(g_joints), B,
            ncomps, pose_mode, cidx, root_palm, g_pose.data_ptr(), _ptr(g_betas), scratch.data_ptr(), _stream()),
            "obman_mano_lbs_bwd")
        return g_pose.view(in_shape), g_betas, None, None, None, None, None, None, None


def mano_lbs(pose, betas, blob_right, blob_left=None, side=None, ncomps=30, use_pca=True, center_idx=0,
             root_palm=False):
    """ManoLayer.forward replacement -> (verts [B,778,3] mm, joints [B,21,3] mm).

    pose [B,3+ncomps] with ``use_pca`` (root axis-angle + PCA coefficients); without it either [B,48] axis-angle or - the
    reference's ``mano_use_pca=False`` path (manobranch.py:126-128) - [B,16,3,3] rotation matrices, used as given.
    betas [B,10]|None.  ``side`` int32 [B] picks right(0)/left(1) model blob per sample."""
    pose_mode = 1 if use_pca else (2 if pose.dim() == 4 else 0)
    return _ManoLBS.apply(pose, betas, blob_right, blob_left, side, ncomps, pose_mode, center_idx, root_palm)


def project_rotations(mats):
    """[...,3,3] -> closest rotation matrices (orthogonal Procrustes via SVD, det = +1): manopth's optional ``robust_rot``
    pre-processing of rotation-matrix poses.  A 3x3 SVD per joint on the device (rocSOLVER through torch.linalg), autograd
    supplies the backward; the skinning itself stays in ``csrc/mano_lbs.hip``."""
    u, _, vh = torch.linalg.svd(mats)
    det = torch.det(u @ vh)
    fix = torch.ones_like(mats[..., 0])
    fix[..., 2] = det
    return (u * fix.unsqueeze(-2)) @ vh


def mesh_contains_hits(points, verts, faces, patches=1, all_pairs=False, raw_bits=False):
    """points [B,P,3], verts [B,Nv,3], faces [F,3] int32 (device) -> hits [B,P] int32 whose PARITY is the inside test
    (exterior <=> even).  ``patches == 1`` (the reference's single closed mesh): ray/triangle crossings along the
    reference's fixed direction.  ``patches > 1`` (multi-patch template, faces = ``patches`` equal consecutive groups, each a
    closed surface): 1 where the point is inside ANY patch (OR of the per-patch parities), else 0 (``raw_bits``: the
    per-patch parity word itself).  No gradient.  ``all_pairs`` runs the all-pairs checker kernel
    (``obman_mesh_contains_bruteforce_fwd``) instead of the grid-culled product kernel: same words, ~100x the pair tests -
    for tests and A/B measurements only."""
    points, verts = _dev(points.detach(), "points"), _dev(verts.detach(), "verts")
    faces = _dev(faces, "faces", torch.int32)
    if points.dim() != 3 or verts.dim() != 3 or faces.dim() != 2 or faces.shape[1] != 3:
        raise ValueError("expected points [B,P,3], verts [B,Nv,3], faces [F,3]")
    B, P, Nv, F = points.shape[0], points.shape[1], verts.shape[1], faces.shape[0]
    hits = torch.empty((B, P), dtype=torch.int32, device=points.device)
    patches = int(patches)
    if patches > 1 and (F % patches != 0 or patches > 32):
        raise ValueError("multi-patch inside test needs <= 32 equal face groups, got F=%d patches=%d" % (F, patches))
    group_faces = F // patches if patches > 1 else 0
    if all_pairs:
        _lib.check(_lib.lib().obman_mesh_contains_bruteforce_fwd(points.data_ptr(), verts.data_ptr(), faces.data_ptr(), B, P,
                                                                 Nv, F, group_faces, hits.data_ptr(), _stream()),
                   "obman_mesh_contains_bruteforce_fwd")
    elif patches <= 1:
        _lib.check(_lib.lib().obman_mesh_contains_fwd(points.data_ptr(), verts.data_ptr(), faces.data_ptr(), B, P, Nv, F,
                                                      hits.data_ptr(), _stream()), "obman_mesh_contains_fwd")
    else:
        _lib.check(_lib.lib().obman_mesh_contains_groups_fwd(points.data_ptr(), verts.data_ptr(), faces.data_ptr(), B, P, Nv,
                                                             F, group_faces, hits.data_ptr(), _stream()),
                   "obman_mesh_contains_groups_fwd")
    if patches <= 1 or raw_bits:
        return hits
    return (hits != 0).to(torch.int32)


MODES = {"dist_sq": 0, "dist": 1, "dist_tanh": 2}
TARGETS = {"all": 0, "obj": 1, "hand": 2}


class _ContactTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hand, obj, idx21, mins21, hits, zone_ids, zone_off, n_zones, zone_mode, contact_mode,
                contact_thresh, collision_mode, collision_thresh, target):
        hand, obj = _dev(hand, "hand"), _dev(obj, "obj")
        B, V, N = hand.shape[0], hand.shape[1], obj.shape[1]
        dev = hand.device
        attr = torch.empty((B, V), dtype=torch.uint8, device=dev)
        rep = torch.empty((B, V), dtype=torch.uint8, device=dev)
        cpts = torch.empty((B, V, 3), dtype=torch.float32, device=dev)
        partials = torch.empty((B, 8), dtype=torch.float32, device=dev)
        out = torch.empty((8,), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().obman_contact_fwd(
            hand.data_ptr(), obj.data_ptr(), idx21.data_ptr(), mins21.data_ptr(), hits.data_ptr(), B, V, N,
            _ptr(zone_ids), _ptr(zone_off), int(n_zones), int(zone_mode), int(contact_mode), float(contact_thresh),
            int(collision_mode), float(collision_thresh), attr.data_ptr(), rep.data_ptr(), cpts.data_ptr(),
            partials.data_ptr(), out.data_ptr(), _stream()), "obman_contact_fwd")
        ctx.save_for_backward(hand, obj, idx21, attr, rep, out)
        ctx.cfg = (B, V, N, int(contact_mode), float(contact_thresh), int(collision_mode), float(collision_thresh),
                   int(target))
        ctx.mark_non_differentiable(attr, rep, cpts, out)
        return out[0], out[1], out, attr, rep, cpts

    @staticmethod
    def backward(ctx, g_missed, g_penetr, *_unused):
        hand, obj, idx21, attr, rep, out = ctx.saved_tensors
        B, V, N, cmode, cth, kmode, kth, target = ctx.cfg
        g_missed = g_missed.contiguous() if g_missed is not None else None
        g_penetr = g_penetr.contiguous() if g_penetr is not None else None
        grad_hand = torch.empty_like(hand) if ctx.needs_input_grad[0] else None
        grad_obj = torch.empty_like(obj) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.lib().obman_contact_bwd(
            hand.data_ptr(), obj.data_ptr(), idx21.data_ptr(), attr.data_ptr(), rep.data_ptr(), out.data_ptr(),
            _ptr(g_missed), _ptr(g_penetr), B, V, N, cmode, cth, kmode, kth, target, _ptr(grad_hand), _ptr(grad_obj),
            _stream()), "obman_contact_bwd")
        return (grad_hand, grad_obj) + (None,) * 12


def contact_tail(hand, obj, idx21, mins21, hits, zone_ids, zone_off, n_zones, zone_mode, contact_mode,
                 contact_thresh, collision_mode, collision_thresh, target):
    """-> (missed_loss, penetr_loss, out[8], attraction_mask u8, repulsion_mask u8, contact_points)."""
    return _ContactTail.apply(hand, obj, idx21, mins21, hits, zone_ids, zone_off, n_zones, zone_mode, contact_mode,
                              contact_thresh, collision_mode, collision_thresh, target)


def _pointgen_params(feat, grid, tensors, running, cfg):
    """obman_pointgen_params over tensors the caller keeps alive (raw pointers: valid only while those tensors live)."""
    training, eps, momentum, out_factor, mfma_bf16 = cfg
    p = _lib.PointGenParams()
    p.B, p.N, p.C1, p.training = feat.shape[0], grid.shape[-2], tensors[0].shape[0], int(training)
    p.eps, p.momentum, p.out_factor = float(eps), float(momentum), float(out_factor)
    p.grid_per_sample = int(grid.dim() == 3)
    p.mfma_bf16 = int(bool(mfma_bf16) and grid.dim() == 2)  # the per-sample-grid path is exact fp32 only
    p.grid, p.feat = grid.data_ptr(), feat.data_ptr()
    for name, t in zip(("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4"), tensors[:8]):
        setattr(p, name, t.data_ptr())
    for k in range(3):
        p.bn_w[k], p.bn_b[k] = tensors[8 + 2 * k].data_ptr(), tensors[9 + 2 * k].data_ptr()
        rm, rv = running[k] if running is not None else (None, None)
        p.bn_rm[k] = rm.data_ptr() if rm is not None else None
        p.bn_rv[k] = rv.data_ptr() if rv is not None else None
    return p


class _PointGen(torch.autograd.Function):
    """PointGenCon over (template grid x per-sample feature) without the [B,C,N] concat (csrc/decoder.hip)."""

    @staticmethod
    def forward(ctx, feat, grid, w1, b1, w2, b2, w3, b3, w4, b4, g1, be1, g2, be2, g3, be3, running, cfg):
        feat, grid = _dev(feat, "features"), _dev(grid, "grid")
        tensors = [_dev(t, "decoder parameter") for t in (w1, b1, w2, b2, w3, b3, w4, b4, g1, be1, g2, be2, g3, be3)]
        B, N, C1 = feat.shape[0], grid.shape[-2], w1.shape[0]
        if feat.dim() != 2 or feat.shape[1] != C1 - 3 or grid.shape[-1] != 3 or grid.dim() not in (2, 3) or (
                grid.dim() == 3 and grid.shape[0] != B):
            raise ValueError("features must be [B,%d] and grid [N,3] (shared template) or [B,N,3] (one point set per sample)" % (C1 - 3))
        p = _pointgen_params(feat, grid, tensors, running, cfg)
        lib = _lib.lib()
        n_ws = lib.obman_pointgen_ws_floats(_ct.addressof(p), 0)
        if n_ws <= 0:
            raise _lib.ObmanHipError("obman_pointgen_ws_floats: unsupported decoder shape (C1=%d)" % C1)
        ws = torch.empty(n_ws, dtype=torch.float32, device=feat.device)
        out = torch.empty((B, N, 3), dtype=torch.float32, device=feat.device)
        _lib.check(lib.obman_pointgen_fwd(_ct.addressof(p), out.data_ptr(), ws.data_ptr(), _stream()), "obman_pointgen_fwd")
        # Everything the backward dereferences goes through save_for_backward: the tensors stay alive AND autograd's version
        # check refuses a backward after an in-place change (optimizer step, load_state_dict) of a weight between the two
        # calls.  No raw pointer outlives this call; the backward rebuilds its parameter block from the saved tensors and
        # never touches the BatchNorm running buffers (a .to() / re-allocation of those in between is harmless).
        ctx.save_for_backward(feat, grid, ws, *tensors)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, g_out):
        feat, grid, ws = ctx.saved_tensors[:3]
        tensors = ctx.saved_tensors[3:]
        p = _pointgen_params(feat, grid, tensors, None, ctx.cfg)
        lib = _lib.lib()
        g_out = g_out.contiguous()
        ws2 = torch.empty(lib.obman_pointgen_ws_floats(_ct.addressof(p), 1), dtype=torch.float32, device=feat.device)
        grads = [torch.empty_like(t) for t in tensors]
        g_feat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        g = _lib.PointGenGrads()
        for name, t in zip(("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4"), grads[:8]):
            setattr(g, name, t.data_ptr())
        for k in range(3):
            g.bn_w[k], g.bn_b[k] = grads[8 + 2 * k].data_ptr(), grads[9 + 2 * k].data_ptr()
        g.feat = _ptr(g_feat)
        _lib.check(lib.obman_pointgen_bwd(_ct.addressof(p), g_out.data_ptr(), ws.data_ptr(), ws2.data_ptr(),
                                          _ct.addressof(g), _stream()), "obman_pointgen_bwd")
        return (g_feat, None, *grads, None, None)


def pointgen_decode(decoder, features, grid, mfma_dtype=None):
    """decoder: PointGenCon-like module (conv1..4, bn1..3, out_factor); features [B,C], grid [N,3] -> [B,N,3].
    ``grid`` may also be [B,N,3] - one point set per sample, as ``AtlasBranch.forward`` draws them (atlasbranch.py:78-108);
    that path always contracts in exact fp32.

    mfma_dtype (default: the module's ``mfma_dtype`` attribute, else "f32"): "f32" = exact fp32 MFMA contraction;
    "bf16" = GEMM operands rounded to bf16 on the fly (fp32 master weights, fp32 accumulation, fp32 BatchNorm
    statistics) on ``v_mfma_f32_32x32x16_bf16`` - the BASELINE configs[2] flavour."""
    mfma_dtype = mfma_dtype or getattr(decoder, "mfma_dtype", "f32")
    if mfma_dtype not in ("f32", "bf16"):
        raise ValueError("mfma_dtype must be 'f32' or 'bf16', got %r" % (mfma_dtype,))
    bns = (decoder.bn1, decoder.bn2, decoder.bn3)
    training = decoder.training or any(bn.running_mean is None for bn in bns)
    running = []
    momentum = 0.0
    for bn in bns:
        running.append((bn.running_mean, bn.running_var) if bn.running_mean is not None else (None, None))
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked + 1)
        else:
            momentum = bn.momentum
    if getattr(decoder, "use_tanh", False):
        raise NotImplementedError("use_tanh: traineval.py:54 always passes atlas_use_tanh=False")
    convs = (decoder.conv1, decoder.conv2, decoder.conv3, decoder.conv4)
    args = []
    for c in convs:
        args += [c.weight.view(c.weight.shape[0], c.weight.shape[1]), c.bias]
    for bn in bns:
        args += [bn.weight, bn.bias]
    out = _PointGen.apply(features, grid, *args, tuple(running), (training, bns[0].eps, momentum, decoder.out_factor, mfma_dtype == "bf16"))
    if decoder.training:
        with torch.no_grad():
            for bn in bns:
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
    return out


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


_BN_DTYPES = (torch.float32, torch.bfloat16)


def _bn_entry(name, dtype):
    """C-ABI entry point of the fused BN family for fp32 / bf16 activations."""
    return getattr(_lib.lib(), name + ("_bf16" if dtype == torch.bfloat16 else ""))


class _BNAct(torch.autograd.Function):
    """BatchNorm2d (+ skip) (+ ReLU) on a channels_last fp32 or bf16 tensor (csrc/bnact.hip; parameters and statistics fp32)."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, rmean, rvar, training, eps, momentum, relu, dual=False):
        B, C, H, W = x.shape
        R = B * H * W
        lib = _lib.lib()
        y = torch.empty_like(x)  # preserves channels_last
        stats = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.obman_bnact_ws_floats(R, C), dtype=torch.float32, device=x.device)
        _lib.check(_bn_entry("obman_bnact_fwd", x.dtype)(x.data_ptr(), _ptr(skip), weight.data_ptr(), bias.data_ptr(), _ptr(rmean),
                                                         _ptr(rvar), R, C, int(training), float(eps), float(momentum), int(relu),
                                                         y.data_ptr(), stats.data_ptr(), ws.data_ptr(), _stream()), "obman_bnact_fwd")
        ctx.save_for_backward(x, y if (relu and skip is not None) else None, weight, stats)
        ctx.cfg = (R, C, int(training), int(relu), skip is not None)
        if dual:
            # the output twice (one storage): a residual block's output has two consumers - the next block's first convolution
            # and its skip branch - and with one output object per consumer autograd hands backward() their gradients
            # separately instead of adding them with a kernel of its own; obman_bnact_bwd2 adds them while it reads
            ctx.set_materialize_grads(False)
            return y, y.detach()
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, y, weight, stats = ctx.saved_tensors
        R, C, training, relu, has_skip = ctx.cfg
        lib = _lib.lib()
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:  # neither output was used
            return (None,) * 11

        def _norm(d):
            if d.dtype != x.dtype:
                d = d.to(x.dtype)
            return d if _is_nhwc(d) else d.contiguous(memory_format=torch.channels_last)

        dy = _norm(dy)
        if dy2 is not None:
            dy2 = _norm(dy2)
            if not (has_skip and relu):  # the kernel adds two gradients only in the form the residual blocks use
                dy, dy2 = dy + dy2, None
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(weight)
        dbeta = torch.empty_like(weight)
        dskip = torch.empty_like(x) if (has_skip and relu) else None
        ws = torch.empty(lib.obman_bnact_ws_floats(R, C), dtype=torch.float32, device=x.device)
        _lib.check(_bn_entry("obman_bnact_bwd2", x.dtype)(x.data_ptr(), _ptr(y), dy.data_ptr(), _ptr(dy2), weight.data_ptr(), stats.data_ptr(),
                                                          R, C, training, relu, int(has_skip), dx.data_ptr(), dgamma.data_ptr(),
                                                          dbeta.data_ptr(), _ptr(dskip), ws.data_ptr(), _stream()), "obman_bnact_bwd2")
        if has_skip and not relu:
            dskip = dy
        return dx, dskip, dgamma, dbeta, None, None, None, None, None, None, None


def bn_act(bn, x, skip=None, relu=True, count=True, dual=False):
    """``relu(bn(x) [+ skip])`` for an ``nn.BatchNorm2d`` module ``bn``.  Fused HIP path for channels_last fp32 or bf16 (autocast
    encoder) ROCm tensors whose channel count is a multiple of 64 (every ResNet-18/50 stage); anything else takes the stock ops.
    ``dual=True`` (fused path only; the stock path returns one tensor): the result as a pair of tensors on one storage, one per
    consumer, so that their gradients reach the fused backward un-added (``_BNAct.forward``)."""
    fused = (x.is_cuda and x.dtype in _BN_DTYPES and _is_nhwc(x) and x.shape[1] % 64 == 0 and bn.affine
             and bn.weight.dtype == torch.float32
             and (skip is None or (_is_nhwc(skip) and skip.dtype == x.dtype and skip.shape == x.shape)))
    training = bn.training or bn.running_mean is None
    if count and bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1  # callers with many layers pass count=False and bump all counters in one launch
    if bn.momentum is not None:
        momentum = bn.momentum
    elif bn.training and bn.running_mean is not None and bn.num_batches_tracked is not None:
        # cumulative moving average: only evaluated where nn.BatchNorm2d.forward evaluates it (training with tracked statistics;
        # it costs a device->host read of the counter), and a counter the caller has not bumped yet (count=False) reads as 1
        momentum = 1.0 / max(float(bn.num_batches_tracked), 1.0)
    else:
        momentum = 0.0
    if not fused:  # stock ops, same bookkeeping as nn.BatchNorm2d.forward (counter bumped above, cumulative momentum)
        y = torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, momentum, bn.eps)
        if skip is not None:
            y = y + skip
        return torch.relu(y) if relu else y
    return _BNAct.apply(x, skip, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, momentum, relu,
                        bool(dual and torch.is_grad_enabled()))


_TERM_WEIGHTS = {}  # (device, weights) -> fp32 tensor on the device, LRU-bounded (a lambda schedule that changes every step
# must not leak one tensor per step).  A weight tensor used WHILE A STREAM IS BEING CAPTURED has its address baked into that graph:
# ownership MOVES from the cache to _CAPTURE_OWNED (trainer.GraphedTrainStep collects its entries, keeps them alive and lets the
# caller rewrite them in place for a changed lambda - which is why a captured tensor must no longer serve eager compositions).
_TERM_WEIGHTS_MAX = 64
_CAPTURE_OWNED = {}  # (device, weights) -> tensor owned by a capture (in progress, or finished and not collected by a GraphedTrainStep)


def _term_weight_tensor(dev, values):
    key = (dev, values)
    if torch.cuda.is_current_stream_capturing():
        w = _CAPTURE_OWNED.get(key)
        if w is None:
            w = _TERM_WEIGHTS.pop(key, None)
            if w is None:  # a tensor made now would be filled by a host copy recorded into the graph
                raise RuntimeError("weighted_terms: loss weights %r first appear while a stream is being captured; run one eager step "
                                   "with them before the capture (trainer.GraphedTrainStep's warm-up does)" % (values,))
            _CAPTURE_OWNED[key] = w
        return w
    w = _TERM_WEIGHTS.pop(key, None)
    if w is None:
        w = torch.tensor(values, dtype=torch.float32, device=dev)
        while len(_TERM_WEIGHTS) >= _TERM_WEIGHTS_MAX:
            _TERM_WEIGHTS.pop(next(iter(_TERM_WEIGHTS)))  # least recently used first (dicts keep insertion order)
    _TERM_WEIGHTS[key] = w
    return w


class _WeightedTerms(torch.autograd.Function):
    """``sum_i w[i] * terms[i]`` of scalar loss terms: three kernels forward, one backward, whatever the number of terms."""

    @staticmethod
    def forward(ctx, w, *terms):
        ctx.save_for_backward(w)
        ctx.shapes = [t.shape for t in terms]
        return (torch.stack([t.reshape(()) for t in terms]) * w).sum()

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        gw = g * w
        return (None,) + tuple(gw[i].reshape(shape) for i, shape in enumerate(ctx.shapes))


def weighted_terms(pairs, shape=()):
    """The loss compositions of the reference - ``final = Tensor([0]); final += lambda_i * term_i`` (``manobranch.py:252-318``),
    ``lambda_a * a + lambda_b * b + ...`` (``atlasbranch.py:247-280``, ``handnet.py:363-366``) - for ``pairs = [(lambda_i, term_i), ...]``.

    Off the GPU: exactly those operations in that order (``0 + x`` is ``x``).  On a ROCm device the 2 - 3 one-element kernels per term
    and direction (scalar multiply, broadcast add, the ``sum`` that undoes the broadcast in the backward: ~90 us of a 10.9 ms
    configs[1] step, profiles/r04_kernels.md section 4) become one stack + multiply + sum, and one multiply in the backward; the
    value differs from the sequential form by the summation order of <= 6 fp32 terms."""
    tensors = [t for _, t in pairs if torch.is_tensor(t)]
    if tensors and tensors[0].is_cuda and all(t.numel() == 1 and t.dtype == torch.float32 for t in tensors):
        const = sum(w * t for w, t in pairs if not torch.is_tensor(t))  # ``scale_weight * 0`` without a scale head (TypeError for a None weight, as in the reference)
        if const != 0:
            raise ValueError("weighted_terms: python-number terms other than 0 are not part of any reference composition")
        pairs = [(w, t) for w, t in pairs if torch.is_tensor(t)]
        dev = tensors[0].device
        w = _term_weight_tensor(dev, tuple(float(w) for w, _ in pairs))
        return _WeightedTerms.apply(w, *tensors).reshape(shape)
    acc = None
    for w, t in pairs:
        acc = w * t if acc is None else acc + w * t
    return acc.reshape(shape) if torch.is_tensor(acc) else acc


class _EdgeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces):
        verts = _dev(verts, "verts")
        faces = _dev(faces, "faces", torch.int32)
        B, N, F = verts.shape[0], verts.shape[1], faces.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=verts.device)
        stats = torch.empty(3 * B, dtype=torch.float32, device=verts.device)
        _lib.check(_lib.lib().obman_edge_loss_fwd(verts.data_ptr(), faces.data_ptr(), B, N, F, loss.data_ptr(), stats.data_ptr(),
                                                  _stream()), "obman_edge_loss_fwd")
        ctx.save_for_backward(verts, faces, stats)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        verts, faces, stats = ctx.saved_tensors
        B, N, F = verts.shape[0], verts.shape[1], faces.shape[0]
        grad = torch.empty_like(verts)
        g = g.contiguous().view(1)
        _lib.check(_lib.lib().obman_edge_loss_bwd(verts.data_ptr(), faces.data_ptr(), B, N, F, stats.data_ptr(), g.data_ptr(),
                                                  grad.data_ptr(), _stream()), "obman_edge_loss_bwd")
        return grad, None


def edge_loss(verts, faces):
    """edge_loss(edges, faces) of atlasbranch.py:153-167: verts [B,N,3], faces [F,3] int32 (device) -> scalar."""
    return _EdgeLoss.apply(verts, faces)


class _LaplacianLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, row_ptr, col, val):
        verts = _dev(verts, "verts")
        B, N = verts.shape[0], verts.shape[1]
        if row_ptr.numel() != N + 1:
            raise ValueError("Laplacian built for %d vertices, got %d" % (row_ptr.numel() - 1, N))
        dev = verts.device
        Lx = torch.empty_like(verts)
        partial = torch.empty(B * ((N + 255) // 256), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().obman_laplacian_fwd(row_ptr.data_ptr(), col.data_ptr(), val.data_ptr(), verts.data_ptr(), B, N,
                                                  Lx.data_ptr(), partial.data_ptr(), loss.data_ptr(), _stream()), "obman_laplacian_fwd")
        ctx.save_for_backward(Lx, row_ptr, col, val)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        Lx, row_ptr, col, val = ctx.saved_tensors
        B, N = Lx.shape[0], Lx.shape[1]
        scratch, grad = torch.empty_like(Lx), torch.empty_like(Lx)
        g = g.contiguous().view(1)
        _lib.check(_lib.lib().obman_laplacian_bwd(row_ptr.data_ptr(), col.data_ptr(), val.data_ptr(), Lx.data_ptr(), g.data_ptr(), B, N,
                                                  scratch.data_ptr(), grad.data_ptr(), _stream()), "obman_laplacian_bwd")
        return grad, None, None, None


def laplacian_loss(verts, row_ptr, col, val):
    """mean row norm of (L . verts) for the fixed template Laplacian given as CSR device tensors (int32, int32, float32)."""
    return _LaplacianLoss.apply(verts, row_ptr, col, val)


class _BNReLUPool(torch.autograd.Function):
    """maxpool3x3/s2/p1(relu(bn(x))) for the ResNet stem without the full-resolution activation (csrc/bnact.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rmean, rvar, training, eps, momentum):
        B, C, H, W = x.shape
        lib = _lib.lib()
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)  # (not .contiguous(...): that is a 67 MB layout copy of uninitialised memory per step)
        stats = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.obman_bnact_ws_floats(B * H * W, C), dtype=torch.float32, device=x.device)
        if x.dtype == torch.bfloat16:  # arg-max taps for the backward: equality routing would hit every tie of the bf16 inputs
            y32 = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
            _lib.check(lib.obman_bnpool_fwd_bf16(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(rmean), _ptr(rvar), B, H, W, C,
                                                 int(training), float(eps), float(momentum), y.data_ptr(), y32.data_ptr(),
                                                 stats.data_ptr(), ws.data_ptr(), _stream()), "obman_bnpool_fwd_bf16")
        else:
            y32 = y
            _lib.check(lib.obman_bnpool_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(rmean), _ptr(rvar), B, H, W, C,
                                            int(training), float(eps), float(momentum), y.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                            _stream()), "obman_bnpool_fwd")
        ctx.save_for_backward(x, y32, weight, stats)
        ctx.cfg = (B, H, W, C, int(training))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        B, H, W, C, training = ctx.cfg
        lib = _lib.lib()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if not _is_nhwc(dy):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(weight), torch.empty_like(weight)
        ws = torch.empty(lib.obman_bnact_ws_floats(B * H * W, C), dtype=torch.float32, device=x.device)
        _lib.check(_bn_entry("obman_bnpool_bwd", x.dtype)(x.data_ptr(), y.data_ptr(), dy.data_ptr(), weight.data_ptr(), stats.data_ptr(),
                                                          B, H, W, C, training, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                          ws.data_ptr(), _stream()), "obman_bnpool_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None


def bn_relu_maxpool(bn, x, pool, count=True):
    """``pool(relu(bn(x)))`` for ``nn.MaxPool2d(3, stride=2, padding=1)``; fused for channels_last fp32 / bf16 ROCm tensors with
    C % 64 == 0, stock ops otherwise."""
    def _t(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

    fused = (x.is_cuda and x.dtype in _BN_DTYPES and _is_nhwc(x) and x.shape[1] % 64 == 0 and bn.affine
             and bn.weight.dtype == torch.float32 and _t(pool.kernel_size) == (3, 3) and _t(pool.stride) == (2, 2) and _t(pool.padding) == (1, 1)
             and _t(pool.dilation) == (1, 1) and not pool.ceil_mode and not pool.return_indices)
    if not fused:
        return pool(bn_act(bn, x, relu=True, count=count))
    training = bn.training or bn.running_mean is None
    if count and bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    if bn.momentum is not None:
        momentum = bn.momentum
    elif bn.training and bn.running_mean is not None and bn.num_batches_tracked is not None:
        # cumulative moving average: only evaluated where nn.BatchNorm2d.forward evaluates it (training with tracked statistics;
        # it costs a device->host read of the counter), and a counter the caller has not bumped yet (count=False) reads as 1
        momentum = 1.0 / max(float(bn.num_batches_tracked), 1.0)
    else:
        momentum = 0.0
    return _BNReLUPool.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, momentum)


def image_stream(src_rgb, params_words, max_blur_r, any_contrast, out_res, channels_last=False, black_pad=0,
                 mean=(0.5, 0.5, 0.5), std=(1.0, 1.0, 1.0)):
    """K10 (csrc/imgstream.hip): a batch of source images -> network inputs, the pixel pipeline of
    HandDataset.get_sample (handataset.py:373-405) bit for bit.

    src_rgb [B,Hp,Wp,3] uint8 device tensor (PIL raw "RGB", every image in the top-left corner of its slot),
    params_words [B,24] int32 device tensor (one ``obman_img_params`` per sample, see include/obman_hip.h).
    -> [B,3,out_res,out_res] fp32; with channels_last the same logical shape in torch.channels_last memory format."""
    if not (isinstance(src_rgb, torch.Tensor) and src_rgb.is_cuda and src_rgb.dtype == torch.uint8 and src_rgb.dim() == 4
            and src_rgb.shape[3] == 3):
        raise _lib.ObmanHipError("src_rgb must be a [B,H,W,3] uint8 ROCm tensor (the HIP path has no CPU fallback)")
    params_words = _dev(params_words, "params_words", torch.int32)
    src_rgb = src_rgb.contiguous()
    B, Hp, Wp = int(src_rgb.shape[0]), int(src_rgb.shape[1]), int(src_rgb.shape[2])
    if tuple(params_words.shape) != (B, 24):
        raise ValueError("params_words must be [B,24] int32, got %s" % (tuple(params_words.shape),))
    lib = _lib.lib()
    ws = torch.empty(lib.obman_imgstream_ws_bytes(B, Hp, Wp), dtype=torch.uint8, device=src_rgb.device)
    if channels_last:
        out = torch.empty((B, out_res, out_res, 3), dtype=torch.float32, device=src_rgb.device)
    else:
        out = torch.empty((B, 3, out_res, out_res), dtype=torch.float32, device=src_rgb.device)
    mean3 = (_ct.c_float * 3)(*[float(m) for m in mean])
    std3 = (_ct.c_float * 3)(*[float(s) for s in std])
    _lib.check(lib.obman_imgstream_fwd(_ptr(src_rgb), B, Hp, Wp, _ptr(params_words), int(max_blur_r), int(bool(any_contrast)),
                                       int(out_res), int(bool(channels_last)), int(black_pad), _ct.addressof(mean3),
                                       _ct.addressof(std3), _ptr(ws), _ptr(out), _stream()), "obman_imgstream_fwd")
    return out.permute(0, 3, 1, 2) if channels_last else out


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: small operators of the step as single launches (csrc/stepops.hip)
_SCRATCH = {}  # (name, device, stream) -> persistent fp32 scratch (launch-ordered on that stream; graph replays reuse it)


def _scratch(name, floats, device):
    key = (name, device, _stream())
    t = _SCRATCH.get(key)
    if t is None or t.numel() < floats:
        t = _SCRATCH[key] = torch.empty(int(floats), dtype=torch.float32, device=device)
    return t


class _AffinePoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, scale, trans):
        verts = _dev(verts, "verts")
        B, N = verts.shape[0], verts.shape[1]
        s = _dev(scale, "scale").reshape(B) if scale is not None else None
        t = _dev(trans, "trans").reshape(B, 3) if trans is not None else None
        out = torch.empty_like(verts)
        _lib.check(_lib.lib().obman_affine_points_fwd(verts.data_ptr(), _ptr(s), _ptr(t), B, N, out.data_ptr(), _stream()),
                   "obman_affine_points_fwd")
        ctx.save_for_backward(verts, s)
        ctx.shapes = (None if scale is None else scale.shape, None if trans is None else trans.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        verts, s = ctx.saved_tensors
        B, N = verts.shape[0], verts.shape[1]
        g = _dev(g, "grad")
        need_v, need_s, need_t = ctx.needs_input_grad
        gv = torch.empty_like(verts) if need_v else None
        gs = torch.empty(B, dtype=torch.float32, device=verts.device) if (need_s and ctx.shapes[0] is not None) else None
        gt = torch.empty((B, 3), dtype=torch.float32, device=verts.device) if (need_t and ctx.shapes[1] is not None) else None
        ws = _scratch("affine", _lib.lib().obman_affine_points_ws_floats(B), verts.device)
        _lib.check(_lib.lib().obman_affine_points_bwd(g.data_ptr(), verts.data_ptr(), _ptr(s), B, N, _ptr(gv), _ptr(gs), _ptr(gt),
                                                      ws.data_ptr(), _stream()), "obman_affine_points_bwd")
        return gv, (gs.reshape(ctx.shapes[0]) if gs is not None else None), (gt.reshape(ctx.shapes[1]) if gt is not None else None)


def affine_points(verts, scale=None, trans=None):
    """``scale.unsqueeze(1) * verts + trans.unsqueeze(1)`` (atlasbranch.py:133-138; verts [B,N,3], scale [B,1], trans [B,3]) as
    one launch, and a backward of two launches where autograd runs two broadcast products and two [B,N,3] reductions."""
    return _AffinePoints.apply(verts, scale, trans)


class _MseTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n_pairs, *tensors):
        preds, targets = tensors[:n_pairs], tensors[n_pairs:]
        preds = [_dev(p, "pred") for p in preds]
        targets = [None if t is None else _dev(t, "target") for t in targets]
        for p, t in zip(preds, targets):
            if t is not None and t.shape != p.shape:
                raise ValueError("mse_terms: target %s does not match prediction %s" % (tuple(t.shape), tuple(p.shape)))
        dev = preds[0].device
        arr = (_lib.MseTerm * n_pairs)()
        for i, (p, t) in enumerate(zip(preds, targets)):
            arr[i].pred, arr[i].target, arr[i].grad, arr[i].n = p.data_ptr(), _ptr(t), None, p.numel()
        out = torch.empty(n_pairs, dtype=torch.float32, device=dev)
        ws = _scratch("mse", _lib.lib().obman_mse_terms_ws_floats(), dev)
        _lib.check(_lib.lib().obman_mse_terms_fwd(_ct.addressof(arr), n_pairs, ws.data_ptr(), out.data_ptr(), _stream()),
                   "obman_mse_terms_fwd")
        ctx.n_pairs = n_pairs
        ctx.has_target = [t is not None for t in targets]
        ctx.save_for_backward(*preds, *[t for t in targets if t is not None])
        return out

    @staticmethod
    def backward(ctx, g):
        n = ctx.n_pairs
        saved = ctx.saved_tensors
        preds, rest = saved[:n], list(saved[n:])
        targets = [rest.pop(0) if has else None for has in ctx.has_target]
        g = _dev(g, "grad")
        arr = (_lib.MseTerm * n)()
        grads = []
        for i, (p, t) in enumerate(zip(preds, targets)):
            gr = torch.empty_like(p) if ctx.needs_input_grad[1 + i] else None
            grads.append(gr)
            arr[i].pred, arr[i].target, arr[i].grad, arr[i].n = p.data_ptr(), _ptr(t), _ptr(gr), p.numel()
        _lib.check(_lib.lib().obman_mse_terms_bwd(_ct.addressof(arr), n, g.data_ptr(), _stream()), "obman_mse_terms_bwd")
        return (None,) + tuple(grads) + (None,) * n


def mse_terms(pairs):
    """``[torch_f.mse_loss(pred, target) for pred, target in pairs]`` (``target=None``: zeros, the reference's
    ``mse_loss(x, zeros_like(x))`` regularisers) as ONE forward launch (+ a one-block finalize) and ONE backward launch for up to
    8 differently sized tensors: the MSE heads of ManoLoss / AtlasLoss (manobranch.py:251-318, atlasbranch.py:211-228).  Returns a
    list of 0-dim tensors (views of one [k] tensor).  Targets get no gradient (they are data in the reference)."""
    if not pairs:
        return []
    if len(pairs) > 8:
        return mse_terms(pairs[:8]) + mse_terms(pairs[8:])
    out = _MseTerms.apply(len(pairs), *[p for p, _ in pairs], *[t for _, t in pairs])
    return [out[i] for i in range(len(pairs))]


def gt_object_stats(gt):
    """Ground-truth object cloud [B,N,3] -> (centroids [B,3], centred [B,N,3], max point norm [B,1]): the target preparation of
    AtlasLoss.compute_loss (atlasbranch.py:211-222: ``gt.mean(1)``, ``gt - centroids.unsqueeze(1)``,
    ``torch.norm(centred, 2, 2).max(1)[0].unsqueeze(1)``) in one launch instead of five.  No gradient."""
    gt = _dev(gt.detach(), "gt")
    B, N = gt.shape[0], gt.shape[1]
    cen = torch.empty((B, 3), dtype=torch.float32, device=gt.device)
    centred = torch.empty_like(gt)
    mx = torch.empty((B, 1), dtype=torch.float32, device=gt.device)
    _lib.check(_lib.lib().obman_gt_object_stats(gt.data_ptr(), B, N, cen.data_ptr(), centred.data_ptr(), mx.data_ptr(), _stream()),
               "obman_gt_object_stats")
    return cen, centred, mx


def bf16_shadow(param, out=None):
    """bf16 copy of an fp32 tensor in the SAME memory order (strides kept: a channels_last filter stays channels_last), written by
    ``obman_bf16_shadow``; ``optim.ObmanAdam`` keeps it current from then on."""
    if out is None:
        out = torch.empty_like(param, dtype=torch.bfloat16, memory_format=torch.preserve_format)
    if out.stride() != param.stride() or not _is_dense(param):
        raise ValueError("bf16_shadow: needs a dense tensor and a shadow with the same strides")
    _lib.check(_lib.lib().obman_bf16_shadow(param.data_ptr(), out.data_ptr(), param.numel(), _stream()), "obman_bf16_shadow")
    return out


def _is_dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class _ShadowConv2d(torch.autograd.Function):
    """conv2d(x, weight) of a bf16-autocast encoder reading the weight's bf16 SHADOW (kept current by the optimizer kernel) instead
    of casting the fp32 filter every step; the gradient goes to the fp32 parameter.  MIOpen does the convolutions (north_star)."""

    @staticmethod
    def forward(ctx, x, weight, shadow, stride, padding, dilation, groups):
        ctx.save_for_backward(x, shadow)
        ctx.conf = (stride, padding, dilation, groups)
        return torch.ops.aten.convolution(x, shadow, None, stride, padding, dilation, False, [0, 0], groups)

    @staticmethod
    def backward(ctx, gy):
        x, shadow = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conf
        gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, shadow, None, stride, padding, dilation, False, [0, 0], groups,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return gx, gw, None, None, None, None, None


def shadow_conv2d(conv, x):
    """``conv(x)`` for an ``nn.Conv2d`` without bias inside a bf16-autocast region when ``conv.weight`` carries a bf16 shadow
    (``conv.weight._obman_shadow``, attached by ``optim.attach_bf16_shadows`` and rewritten by ``optim.ObmanAdam``'s kernel): same
    MIOpen kernels, no weight-cast launch.  Anything else takes ``conv(x)`` itself.

    Staleness: the optimizer kernel writes filter and shadow together through raw pointers, which leaves the filter's autograd
    version counter alone; any torch-side in-place write (``load_state_dict``, another optimizer, ``copy_``) bumps it.  A shadow whose
    recorded version differs from the filter's - or whose strides differ (the encoder re-lays its filters out as channels_last in
    its first forward) - is rewritten here before it is used.  (Writes through ``.data`` are invisible to the counter:
    ``optim.refresh_bf16_shadows`` after those.)"""
    w = conv.weight
    sh = getattr(w, "_obman_shadow", None)
    if (sh is None or conv.bias is not None or not x.is_cuda or not torch.is_autocast_enabled()
            or torch.get_autocast_dtype("cuda") != torch.bfloat16 or conv.padding_mode != "zeros" or not _is_dense(w)):
        return conv(x)
    if sh.stride() != w.stride():
        sh = w._obman_shadow = bf16_shadow(w.detach())
        w._obman_shadow_version = w._version
    elif getattr(w, "_obman_shadow_version", None) != w._version:
        bf16_shadow(w.detach(), out=sh)
        w._obman_shadow_version = w._version
    with torch.autocast("cuda", enabled=False):
        return _ShadowConv2d.apply(x.to(torch.bfloat16), w, sh, list(conv.stride), list(conv.padding), list(conv.dilation),
                                   conv.groups)
