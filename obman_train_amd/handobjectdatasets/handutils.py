"""Crop geometry of the input stream (host side, a few flops per sample).

Mirrors ``handobjectdatasets/handutils.py`` of the reference (function names, argument meaning, float32 results):
``get_annot_scale`` :8-22, ``get_annot_center`` :25-33, ``transform_coords`` :36-45, ``get_affine_transform`` :63-91,
``get_affine_trans_no_rot`` :94-101.  ``transform_img`` has no host implementation here:
it is the warp kernel (``fixed_point_affine`` below prepares its six integers).
"""
import math

import numpy as np


def get_annot_scale(annots, visibility=None, scale_factor=2.2):
    """Side of the square crop: the larger extent of the 2-D annotations times ``scale_factor``."""
    if visibility is not None:
        annots = annots[visibility]
    lo, hi = annots.min(0), annots.max(0)
    return max(hi[0] - lo[0], hi[1] - lo[1]) * scale_factor


def get_annot_center(annots, visibility=None):
    if visibility is not None:
        annots = annots[visibility]
    lo, hi = annots.min(0), annots.max(0)
    return np.asarray([int((hi[0] + lo[0]) / 2), int((hi[1] + lo[1]) / 2)])


def transform_coords(pts, affine_trans, invert=False):
    """pts [n,2] -> integer pixel coordinates under the 3x3 homogeneous transform."""
    if invert:
        affine_trans = np.linalg.inv(affine_trans)
    hom = np.concatenate([pts, np.ones([np.array(pts).shape[0], 1])], 1)
    return affine_trans.dot(hom.transpose()).transpose()[:, :2].astype(int)


def get_affine_trans_no_rot(center, scale, res):
    """Axis-aligned crop of the ``scale``-sided square around ``center`` to ``res`` = (rows, cols) pixels."""
    t = np.zeros((3, 3))
    t[0, 0] = float(res[1]) / scale
    t[1, 1] = float(res[0]) / scale
    t[0, 2] = res[1] * (-float(center[0]) / scale + 0.5)
    t[1, 2] = res[0] * (-float(center[1]) / scale + 0.5)
    t[2, 2] = 1
    return t


def _rotation(rot):
    m = np.zeros((3, 3))
    sn, cs = np.sin(rot), np.cos(rot)
    m[0, :2] = [cs, -sn]
    m[1, :2] = [sn, cs]
    m[2, 2] = 1
    return m


def get_affine_transform(center, scale, res, rot=0):
    """-> (crop o rotation about the pixel origin, rotation-free crop about the centre as rotated around the middle of
    the output image), both float32.  The first warps the image, the second left-multiplies the camera intrinsics."""
    rot_mat = _rotation(rot)
    hom = np.asarray(center).tolist() + [1]
    origin_rot_center = rot_mat.dot(hom)[:2]
    to_mid = np.eye(3)
    to_mid[0, 2] = -res[1] / 2
    to_mid[1, 2] = -res[0] / 2
    from_mid = to_mid.copy()
    from_mid[:2, 2] *= -1
    mid_rot_center = from_mid.dot(rot_mat).dot(to_mid).dot(hom)
    total = get_affine_trans_no_rot(origin_rot_center, scale, res).dot(rot_mat)
    post_rot = get_affine_trans_no_rot(mid_rot_center[:2], scale, res)
    return total.astype(np.float32), post_rot.astype(np.float32)


def fixed_point_affine(affine_trans, res):
    """Six 16.16 integers of the warp kernel for ``transform_img(img, affine_trans, res)`` (reference
    ``handutils.py:48-60``: PIL ``img.transform(res, AFFINE, inverse coefficients)``, NEAREST): PIL walks
    ``xin = (A2 + x*A0 + y*A1) >> 16`` with the half-pixel offset folded into A2/A5 and only while the four output corners
    stay inside +-32768 source pixels - beyond that it switches to a float path, which this stream does not implement."""
    inv = np.linalg.inv(affine_trans)
    a = [float(inv[0, 0]), float(inv[0, 1]), float(inv[0, 2]), float(inv[1, 0]), float(inv[1, 1]), float(inv[1, 2])]
    for (x, y) in ((0, 0), (res[0], res[1]), (0, res[1]), (res[0], 0)):
        if not (abs(x * a[0] + y * a[1] + a[2]) < 32768.0 and abs(x * a[3] + y * a[4] + a[5]) < 32768.0):
            raise ValueError("affine crop reaches beyond +-32768 source pixels: outside the fixed-point warp")

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))

    return [fix(a[0]), fix(a[1]), fix(a[2] + a[0] * 0.5 + a[1] * 0.5), fix(a[3]), fix(a[4]), fix(a[5] + a[3] * 0.5 + a[4] * 0.5)]
