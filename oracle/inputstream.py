"""CPU oracle of the image / annotation input stream.  TEST INFRASTRUCTURE ONLY.

Restates, in plain numpy (no PIL, no torchvision), what ``HandDataset.get_sample``
(``handobjectdatasets/handataset.py:103-411``) does to one sample: side flip, centre / scale / rotation
jitter, the affine crop (``handutils.py:48-101``), Gaussian blur + colour jitter of the source image
(``handataset.py:373-383``, ``imgtrans.py:5-53``), nearest-neighbour warp, tensorisation and normalisation
(``handataset.py:384-405``) and the matching 2-D / 3-D annotation transforms.

The pixel arithmetic of the reference lives in two third-party packages that are not under ``/root/reference``:

* **Pillow** (un-pinned in ``environment.yml:11``; Pillow 12.2.0 is installed in the dev container): ``Image.transform``
  (AFFINE, NEAREST), ``ImageFilter.GaussianBlur``, ``Image.blend`` / ``ImageEnhance``, ``convert("L"|"HSV"|"RGB")``.
  Each restatement below is pinned **bit-exactly against the installed Pillow** by
  ``tests/golden/make_golden_inputstream.py`` (exhaustively over all 2^24 colours for the colour-space conversions,
  all 2^16 operand pairs for ``blend``, random sizes / radii / matrices for blur and warp).
* **torchvision** (absent here, un-pinned: it is not even listed in ``environment.yml``): ``adjust_brightness /
  adjust_contrast / adjust_saturation / adjust_hue``, ``to_tensor``, ``normalize`` on PIL images.  Restated from the
  published PIL backend (``torchvision/transforms/_functional_pil.py``): the first three are ``ImageEnhance.{Brightness,
  Contrast,Color}(img).enhance(f)``; ``adjust_hue`` adds ``uint8(hue*255)`` (wrapping) to the H channel of the HSV
  image; ``to_tensor`` is ``uint8 / 255`` in fp32.  **torchvision parity unpinned** (no reference test pins it);
  everything those functions call is Pillow and is pinned.

The golden fixture ``tests/golden/inputstream.npz`` holds outputs of the reference's own ``HandDataset.get_sample``
run here (on Pillow, with the torchvision functions above as stand-ins) for seeded synthetic pose datasets.
"""
import math
import random as _random

import numpy as np

F32 = np.float32
F64 = np.float64

OP_BRIGHTNESS, OP_SATURATION, OP_HUE, OP_CONTRAST = 1, 2, 3, 4


# ------------------------------------------------------------------------------------------------ Pillow primitives
def blend(a, b, alpha):
    """``Image.blend(im1=a, im2=b, alpha)`` on uint8 arrays (Pillow ``Blend.c``: C ``float`` arithmetic; plain
    truncation inside [0,1], clipping outside)."""
    al = F32(alpha)
    a32 = a.astype(np.int32)
    d = (b.astype(np.int32) - a32).astype(F32)
    t = (a32.astype(F32) + (al * d).astype(F32)).astype(F32)
    if 0.0 <= al <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def luma(rgb):
    """``convert("L")`` of an RGB image (ITU-R 601-2, 16.16 fixed point, Pillow ``Convert.c`` ``L24``)."""
    r, g, b = [rgb[..., i].astype(np.uint32) for i in range(3)]
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def rgb2hsv(rgb):
    """``convert("HSV")`` (Pillow ``Convert.c`` ``rgb2hsv_row``): float32 variables, double-typed literals promote."""
    r, g, b = [rgb[..., i] for i in range(3)]
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    mi = maxc.astype(np.int32)
    cr = (mi - minc.astype(np.int32)).astype(F32)
    crs = np.where(cr == 0, F32(1), cr)
    s = (cr / np.where(maxc == 0, 1, maxc).astype(F32)).astype(F32)
    rc = ((mi - r).astype(F32) / crs).astype(F32).astype(F64)
    gc = ((mi - g).astype(F32) / crs).astype(F32).astype(F64)
    bc = ((mi - b).astype(F32) / crs).astype(F32).astype(F64)
    h = np.where(r == maxc, (bc - gc).astype(F32).astype(F64), np.where(g == maxc, 2.0 + rc - bc, 4.0 + gc - rc)).astype(F32)
    h = np.fmod(h.astype(F64) / 6.0 + 1.0, 1.0).astype(F32)
    uh = np.clip((h.astype(F64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(F64) * 255.0).astype(np.int32), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb(hsv):
    """``convert("RGB")`` of an HSV image (Pillow ``Convert.c`` ``hsv2rgb_row``)."""
    h, s, v = [hsv[..., i] for i in range(3)]
    hf = h.astype(F32).astype(F64) * 6.0 / 255.0
    i = np.floor(hf).astype(F32)
    f = (hf - i.astype(F64)).astype(F32).astype(F64)
    fs = (s.astype(F32).astype(F64) / 255.0).astype(F32).astype(F64)
    vf = v.astype(F64)

    def rnd(x):  # C round() on non-negative values
        return np.clip(np.floor(x + 0.5), 0, 255).astype(np.uint8)

    p = rnd(vf * (1.0 - fs))
    q = rnd(vf * (1.0 - fs * f))
    t = rnd(vf * (1.0 - fs * (1.0 - f)))
    k = i.astype(np.int32) % 6
    r = np.choose(k, [v, q, p, p, t, v])
    g = np.choose(k, [t, v, v, q, p, p])
    b = np.choose(k, [p, p, t, v, v, q])
    gray = s == 0
    return np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], -1).astype(np.uint8)


def gaussian_box_radius(sigma, passes=3):
    """Pillow ``BoxBlur.c`` ``_gaussian_blur_radius``: radius of the extended box filter (all C ``float``)."""
    f = F32
    radius = f(sigma)
    sigma2 = f(f(radius * radius) / f(passes))
    L = f(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(math.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    return f(l + a)


def box_weights(fr):
    """(integer radius, ww, fw) of one extended-box pass: 8.24 fixed-point weights of the inner taps and of the two
    fractional far taps (Pillow ``BoxBlur.c`` ``ImagingHorizontalBoxBlur``)."""
    r = int(fr)
    ww = int(np.uint32(F32(1 << 24) / F32(F32(fr) * F32(2) + F32(1))))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def _box_pass_h(img, r, ww, fw):
    H, W, C = img.shape
    x = np.arange(W)
    acc = np.zeros((H, W, C), np.uint64)
    for d in range(-r, r + 1):
        acc += img[:, np.clip(x + d, 0, W - 1)].astype(np.uint64)  # edge pixels are replicated
    far = img[:, np.clip(x - r - 1, 0, W - 1)].astype(np.uint64) + img[:, np.clip(x + r + 1, 0, W - 1)].astype(np.uint64)
    bulk = (acc * np.uint64(ww) + far * np.uint64(fw)) & np.uint64(0xFFFFFFFF)  # UINT32 arithmetic
    return (((bulk + np.uint64(1 << 23)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)).astype(np.uint8)


def gaussian_blur(img, sigma, passes=3):
    """``img.filter(ImageFilter.GaussianBlur(sigma))``: ``passes`` horizontal extended-box passes, then the same
    vertically, rounding to uint8 after every pass."""
    if sigma == 0:
        return img.copy()
    r, ww, fw = box_weights(gaussian_box_radius(sigma, passes))
    out = img
    for _ in range(passes):
        out = _box_pass_h(out, r, ww, fw)
    out = out.transpose(1, 0, 2)
    for _ in range(passes):
        out = _box_pass_h(out, r, ww, fw)
    return np.ascontiguousarray(out.transpose(1, 0, 2))


def fix16(v):
    """Pillow ``Geometry.c`` ``FIX``: 16.16 fixed point, round half up."""
    return int(math.floor(float(v) * 65536.0 + 0.5))


def affine_fixed_coeffs(coeffs):
    """Six integers of the fixed-point walk ``xin = (A2 + x*A0 + y*A1) >> 16``, ``yin = (A5 + x*A3 + y*A4) >> 16``
    (``Geometry.c`` ``affine_fixed``: the half-pixel centre offset is folded into A2 / A5)."""
    a = [float(c) for c in coeffs]
    return [fix16(a[0]), fix16(a[1]), fix16(a[2] + a[0] * 0.5 + a[1] * 0.5),
            fix16(a[3]), fix16(a[4]), fix16(a[5] + a[3] * 0.5 + a[4] * 0.5)]


def affine_fixed_ok(coeffs, out_w, out_h):
    """``Geometry.c`` ``check_fixed`` at the four output corners: Pillow only takes the fixed-point path then."""
    a = [float(c) for c in coeffs]
    for (x, y) in ((0, 0), (out_w, out_h), (0, out_h), (out_w, 0)):
        if not (abs(x * a[0] + y * a[1] + a[2]) < 32768.0 and abs(x * a[3] + y * a[4] + a[5]) < 32768.0):
            return False
    return True


def affine_nearest(img, coeffs, out_w, out_h):
    """``img.transform((out_w, out_h), Image.AFFINE, coeffs)`` with the default NEAREST filter; pixels that map outside
    the source stay 0."""
    assert affine_fixed_ok(coeffs, out_w, out_h), "outside Pillow's fixed-point affine path"
    A = affine_fixed_coeffs(coeffs)
    H, W = img.shape[:2]
    x = np.arange(out_w, dtype=np.int64)[None, :]
    y = np.arange(out_h, dtype=np.int64)[:, None]
    xin = (A[2] + x * A[0] + y * A[1]) >> 16
    yin = (A[5] + x * A[3] + y * A[4]) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.zeros((out_h, out_w) + img.shape[2:], img.dtype)
    out[ok] = img[yin[ok], xin[ok]]
    return out


# ------------------------------------------------------------------------------------- torchvision (PIL backend) ops
def contrast_mean(rgb):
    """``int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)`` (``ImageEnhance.Contrast.__init__``)."""
    L = luma(rgb)
    return int(float(int(L.astype(np.uint64).sum())) / float(L.size) + 0.5)


def hue_shift(hue_factor):
    """``np.uint8(hue_factor * 255)`` as used by ``adjust_hue``: truncation toward zero, then wrap modulo 256."""
    return int(hue_factor * 255) & 0xFF


def apply_color_op(rgb, op, factor):
    if op == OP_BRIGHTNESS:  # ImageEnhance.Brightness: blend(black, img, f)
        return blend(np.zeros_like(rgb), rgb, factor)
    if op == OP_SATURATION:  # ImageEnhance.Color: blend(gray, img, f)
        return blend(np.repeat(luma(rgb)[..., None], 3, -1), rgb, factor)
    if op == OP_CONTRAST:  # ImageEnhance.Contrast: blend(mean gray level of the WHOLE image, img, f)
        return blend(np.full_like(rgb, contrast_mean(rgb)), rgb, factor)
    if op == OP_HUE:  # adjust_hue
        if not (-0.5 <= factor <= 0.5):
            raise ValueError("hue_factor is not in [-0.5, 0.5].")
        hsv = rgb2hsv(rgb)
        hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift(factor)).astype(np.uint8)
        return hsv2rgb(hsv)
    raise ValueError(op)


def get_color_params(brightness=0, contrast=0, saturation=0, hue=0, rng=_random):
    """``imgtrans.py:5-28``: draws in the order brightness, contrast, saturation, hue."""
    b = rng.uniform(max(0, 1 - brightness), 1 + brightness) if brightness > 0 else None
    c = rng.uniform(max(0, 1 - contrast), 1 + contrast) if contrast > 0 else None
    s = rng.uniform(max(0, 1 - saturation), 1 + saturation) if saturation > 0 else None
    h = rng.uniform(-hue, hue) if hue > 0 else None
    return b, c, s, h


def color_jitter_plan(brightness=0, contrast=0, saturation=0, hue=0, rng=_random):
    """``imgtrans.py:30-53``: the op list is built as [brightness, saturation, hue, contrast] and shuffled."""
    b, c, s, h = get_color_params(brightness, contrast, saturation, hue, rng)
    ops = []
    if b is not None:
        ops.append((OP_BRIGHTNESS, b))
    if s is not None:
        ops.append((OP_SATURATION, s))
    if h is not None:
        ops.append((OP_HUE, h))
    if c is not None:
        ops.append((OP_CONTRAST, c))
    rng.shuffle(ops)
    return ops


def to_tensor_normalize(rgb, mean=(0.5, 0.5, 0.5), std=(1, 1, 1), black_padding=False, inp_res=None):
    """``to_tensor(img).float()``, optional black frame, ``normalize`` (``handataset.py:389-405``) -> [3,H,W] fp32."""
    t = (rgb.transpose(2, 0, 1).astype(F32) / F32(255)).astype(F32)
    if black_padding:
        pad = int(inp_res * 0.2)
        t[:, 0:pad, :] = 0
        t[:, -pad:-1, :] = 0
        t[:, :, 0:pad] = 0
        t[:, :, -pad:-1] = 0
    m = np.asarray(mean, F32)[:, None, None]
    s = np.asarray(std, F32)[:, None, None]
    return ((t - m).astype(F32) / s).astype(F32)


# ------------------------------------------------------------------------------------------------- handutils.py
def get_affine_trans_no_rot(center, scale, res):
    """``handutils.py:94-101``."""
    t = np.zeros((3, 3))
    t[0, 0] = float(res[1]) / scale
    t[1, 1] = float(res[0]) / scale
    t[0, 2] = res[1] * (-float(center[0]) / scale + 0.5)
    t[1, 2] = res[0] * (-float(center[1]) / scale + 0.5)
    t[2, 2] = 1
    return t


def get_affine_transform(center, scale, res, rot=0):
    """``handutils.py:63-91``: (crop o rotation about the origin) and the rotation-free crop about the centre rotated
    around the image middle, both as float32."""
    rot_mat = np.zeros((3, 3))
    sn, cs = np.sin(rot), np.cos(rot)
    rot_mat[0, :2] = [cs, -sn]
    rot_mat[1, :2] = [sn, cs]
    rot_mat[2, 2] = 1
    hom = np.asarray(center).tolist() + [1]
    origin_rot_center = rot_mat.dot(hom)[:2]
    t_mat = np.eye(3)
    t_mat[0, 2] = -res[1] / 2
    t_mat[1, 2] = -res[0] / 2
    t_inv = t_mat.copy()
    t_inv[:2, 2] *= -1
    transformed_center = t_inv.dot(rot_mat).dot(t_mat).dot(hom)
    total = get_affine_trans_no_rot(origin_rot_center, scale, res).dot(rot_mat)
    post = get_affine_trans_no_rot(transformed_center[:2], scale, res)
    return total.astype(F32), post.astype(F32)


def transform_coords(pts, affine_trans):
    """``handutils.py:36-45``: homogeneous transform, truncated to int."""
    hom = np.concatenate([pts, np.ones([np.array(pts).shape[0], 1])], 1)
    return affine_trans.dot(hom.transpose()).transpose()[:, :2].astype(int)


def transform_img(rgb, affine_trans, res):
    """``handutils.py:48-60`` + the crop of ``handataset.py:385-387`` (a no-op: the transform already has size res)."""
    trans = np.linalg.inv(affine_trans)
    coeffs = (trans[0, 0], trans[0, 1], trans[0, 2], trans[1, 0], trans[1, 1], trans[1, 2])
    return affine_nearest(rgb, coeffs, res[0], res[1])


def points_from_mesh(faces, vertices, vertex_nb=600):
    """``vertexsample.py:11-29``: area-weighted uniform surface samples (global ``np.random`` draws, same order)."""
    v = vertices[faces]
    areas = 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)
    proba = areas / areas.sum()
    rand_idxs = np.random.choice(range(areas.shape[0]), size=vertex_nb, p=proba)
    u = np.random.rand(vertex_nb, 1)
    w = np.random.rand(vertex_nb, 1)
    out = u + w > 1
    u[out] = 1 - u[out]
    w[out] = 1 - w[out]
    tris = vertices[faces[rand_idxs]]
    return tris[:, 0] + u * (tris[:, 1] - tris[:, 0]) + w * (tris[:, 2] - tris[:, 0])


# ---------------------------------------------------------------------------------------------- HandDataset.get_sample
def get_sample(pose, idx, queries, *, inp_res=256, center_idx=9, point_nb=600, max_rot=np.pi, train=True,
               scale_jittering=0.3, center_jittering=0.2, hue=0.15, saturation=0.5, contrast=0.5, brightness=0.5,
               blur_radius=0.5, sides="both", block_rot=False, black_padding=False, as_obj_only=False):
    """``HandDataset.get_sample`` (``handataset.py:103-411``) for the queries of the training path
    (``traineval.py`` requests images, joints2d/3d, verts3d, objpoints3d, sides, camintrs, affinetrans, center3d).
    ``pose`` offers the reference's pose-dataset accessors plus ``base_names`` / ``trans_names`` (the names of the
    Base / Trans queries it supports, i.e. ``all_queries`` without the Enum types); images are uint8 ``[H,W,3]`` arrays.
    RNG draws use the global ``np.random`` / ``random`` generators in the reference's order.  Keys are plain strings
    (the Enum ``.name`` of the reference's query)."""
    q = set(queries)
    sample = {}
    needs_cs = "images" in q
    if needs_cs:
        center, scale = pose.get_center_scale(idx)
        center = np.array(center)
    flip = False
    if "sides" in q:
        side = pose.get_sides(idx)
        if sides == "right" and side == "left":
            flip, side = True, "right"
        elif sides == "left" and side == "right":
            flip, side = True, "left"
        sample["sides"] = side
    if "images" in q:
        img = np.asarray(pose.get_image(idx))
        if flip:
            img = img[:, ::-1]
        src_w = img.shape[1]
    if flip:
        center[0] = src_w - center[0]
    if train and needs_cs:
        center_offsets = center_jittering * scale * np.random.uniform(low=-1, high=1, size=2)
        center = center + center_offsets.astype(int)
        sj = scale_jittering * np.random.randn() + 1
        sj = np.clip(sj, 1 - scale_jittering, 1 + scale_jittering)
        scale = scale * sj
        rot = np.random.uniform(low=-max_rot, high=max_rot)
    else:
        rot = 0
    if block_rot:
        rot = max_rot
    rot_mat = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(F32)
    if "joints2d" in q or "images" in q:
        affinetrans, post_rot_trans = get_affine_transform(center, scale, [inp_res, inp_res], rot=rot)
        if "affinetrans" in q:
            sample["affinetrans"] = affinetrans
    if "joints2d" in q:
        joints2d = pose.get_joints2d(idx)
        if flip:
            joints2d = joints2d.copy()
            joints2d[:, 0] = src_w - joints2d[:, 0]
        sample["joints2d"] = np.array(transform_coords(joints2d, affinetrans))
    if "camintrs" in q:
        sample["camintrs"] = post_rot_trans.dot(pose.get_camintr(idx))
    center3d = None
    if q & {"joints3d", "verts3d", "objpoints3d"}:
        obj_only = as_obj_only or (("objpoints3d" in q) and not ((set(pose.base_names) & {"joints3d"}) | (set(pose.trans_names) & {"joints3d", "verts3d"})))
        if not obj_only:
            joints3d = pose.get_joints3d(idx)
            if flip:
                joints3d[:, 0] = -joints3d[:, 0]
            if train:
                joints3d = rot_mat.dot(joints3d.transpose(1, 0)).transpose()
            if center_idx is not None:
                center3d = (joints3d[9] + joints3d[0]) / 2 if center_idx == -1 else joints3d[center_idx]
            if "joints3d" in q:
                sample["joints3d"] = joints3d - center3d if center_idx is not None else joints3d
    if "verts3d" in q:
        verts = pose.get_verts3d(idx)
        if flip:
            verts[:, 0] = -verts[:, 0]
        verts = rot_mat.dot(verts.transpose(1, 0)).transpose()
        if center_idx is not None:
            verts = verts - center3d
        sample["verts3d"] = verts
    if "objpoints3d" in q:
        if "objpoints3d" in pose.base_names:
            pts = pose.get_objpoints3d(idx, point_nb=point_nb)
            if flip:
                pts[:, 0] = -pts[:, 0]
            obj = rot_mat.dot(pts.transpose(1, 0)).transpose()
        elif "objverts3d" in pose.base_names:
            ov, of = pose.get_obj_verts_faces(idx)
            if flip:
                ov[:, 0] = -ov[:, 0]
            obj = points_from_mesh(of, ov, vertex_nb=point_nb).astype(F32)
            obj = rot_mat.dot(obj.transpose(1, 0)).transpose()
        else:
            raise ValueError("Requested TransQueries.objpoints3d for dataset without BaseQueries.objpoints3d and BaseQueries.objverts3d")
        if obj_only:
            center3d = (obj.max(0) + obj.min(0)) / 2
        if center_idx is not None or obj_only:
            obj = obj - center3d
        if obj_only:
            obj = obj / np.linalg.norm(obj, 2, 1).max()
        sample["objpoints3d"] = obj
    if "center3d" in q:
        sample["center3d"] = center3d
    if "images" in q:
        if train:
            sigma = _random.random() * blur_radius
            img = gaussian_blur(np.ascontiguousarray(img), sigma)
            for op, f in color_jitter_plan(brightness=brightness, saturation=saturation, hue=hue, contrast=contrast):
                img = apply_color_op(img, op, f)
        img = transform_img(np.ascontiguousarray(img), affinetrans, [inp_res, inp_res])
        sample["images"] = to_tensor_normalize(img, black_padding=black_padding, inp_res=inp_res)
    return sample


# ------------------------------------------------------------------------------- CPU model of the C-ABI entry point
def imgstream_fwd(images, records, out_res, black_pad=0, mean=(0.5, 0.5, 0.5), std=(1, 1, 1)):
    """What ``obman_imgstream_fwd`` (include/obman_hip.h, K10) must return for a batch: ``images`` uint8 [H,W,3] arrays,
    ``records`` dicts with the fields of ``obman_img_params`` (flip, A[6], blur=(r, ww, fw), ops=[(op, factor)]).
    Composition of the pinned primitives above in the reference's order: flip, blur, colour ops on the whole source
    image, fixed-point nearest warp, tensorise, frame, normalise.  -> float32 [B,3,out_res,out_res]."""
    out = []
    for img, rec in zip(images, records):
        img = np.ascontiguousarray(img[:, ::-1] if rec["flip"] else img)
        r, ww, fw = rec["blur"]
        if r >= 0:
            for _ in range(3):
                img = _box_pass_h(img, r, ww, fw)
            img = img.transpose(1, 0, 2)
            for _ in range(3):
                img = _box_pass_h(img, r, ww, fw)
            img = np.ascontiguousarray(img.transpose(1, 0, 2))
        for op, f in rec["ops"]:
            img = apply_color_op(img, op, f)
        A = rec["A"]
        H, W = img.shape[:2]
        x = np.arange(out_res, dtype=np.int64)[None, :]
        y = np.arange(out_res, dtype=np.int64)[:, None]
        xin = (A[2] + x * A[0] + y * A[1]) >> 16
        yin = (A[5] + x * A[3] + y * A[4]) >> 16
        ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
        crop = np.zeros((out_res, out_res, 3), np.uint8)
        crop[ok] = img[yin[ok], xin[ok]]
        t = (crop.transpose(2, 0, 1).astype(F32) / F32(255)).astype(F32)
        if black_pad:
            t[:, 0:black_pad, :] = 0
            t[:, -black_pad:-1, :] = 0
            t[:, :, 0:black_pad] = 0
            t[:, :, -black_pad:-1] = 0
        m = np.asarray(mean, F32)[:, None, None]
        s = np.asarray(std, F32)[:, None, None]
        out.append(((t - m).astype(F32) / s).astype(F32))
    return np.stack(out)
