"""Generate tests/golden/inputstream.npz by running the REFERENCE's own ``HandDataset.get_sample``
(``/root/reference/handobjectdatasets/handataset.py:103-411``) on seeded synthetic pose datasets, and pin the
oracle's Pillow restatements against the installed Pillow.  Dev-container only (needs /root/reference and Pillow);
the fixture it writes is data (inputs are regenerated from seeds by tests/golden/poses.py; expected outputs stored).

Import shims (SURVEY App. A recipe), none of which changes the code under test:
* ``torchvision.transforms.functional`` is absent from the image.  The six functions the reference calls are supplied
  as stand-ins that follow torchvision's published PIL backend and do all pixel work through the real Pillow:
  ``adjust_brightness/contrast/saturation`` = ``ImageEnhance.{Brightness,Contrast,Color}(img).enhance(f)``;
  ``adjust_hue`` = uint8-wrapping add on the H channel of ``img.convert("HSV")``; ``to_tensor`` = uint8 HWC -> CHW / 255;
  ``normalize`` = ``(t - mean) / std``.  (torchvision parity unpinned, see oracle/inputstream.py.)
* ``cv2`` (imported by ``viz2d.py:1`` for drawing only) is an empty stub module.

    python tests/golden/make_golden_inputstream.py
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import matplotlib  # noqa: E402

matplotlib.use("Agg")
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402

import poses  # noqa: E402
from oracle import inputstream as ois  # noqa: E402


def install_shims():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")

    def adjust_brightness(img, f):
        return ImageEnhance.Brightness(img).enhance(f)

    def adjust_contrast(img, f):
        return ImageEnhance.Contrast(img).enhance(f)

    def adjust_saturation(img, f):
        return ImageEnhance.Color(img).enhance(f)

    def adjust_hue(img, hue_factor):
        if not (-0.5 <= hue_factor <= 0.5):
            raise ValueError("hue_factor ({}) is not in [-0.5, 0.5].".format(hue_factor))
        mode = img.mode
        if mode in {"L", "1", "I", "F"}:
            return img
        h, s, v = img.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            np_h += np.array(int(hue_factor * 255)).astype(np.uint8)  # wraps modulo 256
        h = Image.fromarray(np_h, "L")
        return Image.merge("HSV", (h, s, v)).convert(mode)

    def to_tensor(pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t

    def normalize(tensor, mean, std):
        mean = torch.as_tensor(mean, dtype=tensor.dtype)[:, None, None]
        std = torch.as_tensor(std, dtype=tensor.dtype)[:, None, None]
        return tensor.clone().sub_(mean).div_(std)

    for f in (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue, to_tensor, normalize):
        setattr(fn, f.__name__, f)
    tvt.functional = fn
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": fn})
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)


def pin_pillow_primitives():
    """Bit-exact checks of every Pillow restatement in oracle/inputstream.py against the installed Pillow."""
    rng = np.random.RandomState(0)
    A, B = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    ia, ib = Image.fromarray(A, "L"), Image.fromarray(B, "L")
    for alpha in list(rng.uniform(0, 2.5, 60)) + [0.0, 1.0, 0.5, 1.5, 1e-3, 1.999]:
        assert (np.asarray(Image.blend(ia, ib, alpha)) == ois.blend(A, B, alpha)).all(), ("blend", alpha)
    allv = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(allv >> 16) & 255, (allv >> 8) & 255, allv & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert (np.asarray(Image.fromarray(cube, "RGB").convert("L")) == ois.luma(cube)).all(), "luma"
    assert (np.asarray(Image.fromarray(cube, "RGB").convert("HSV")) == ois.rgb2hsv(cube)).all(), "rgb2hsv"
    assert (np.asarray(Image.fromarray(cube, "HSV").convert("RGB")) == ois.hsv2rgb(cube)).all(), "hsv2rgb"
    for (H, W) in [(37, 53), (270, 480), (5, 3), (1, 9), (64, 64)]:
        img = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        for sigma in [0.0, 1e-3, 0.05, 0.2, 0.4999, 0.5, 0.8, 1.3, 2.0, 3.7] + list(rng.uniform(0, 4, 12)):
            want = np.asarray(Image.fromarray(img, "RGB").filter(ImageFilter.GaussianBlur(sigma)))
            assert (want == ois.gaussian_blur(img, sigma)).all(), ("blur", H, W, sigma)
    for trial in range(300):
        H, W = rng.randint(8, 300), rng.randint(8, 500)
        img = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        rot, sc = rng.uniform(-np.pi, np.pi), rng.uniform(0.3, 3.0)
        c, s = np.cos(rot) * sc, np.sin(rot) * sc
        coeffs = (c, -s, rng.uniform(-50, W), s, c, rng.uniform(-50, H))
        if trial % 3 == 0:
            coeffs = tuple(float(np.float32(v)) for v in coeffs)
        want = np.asarray(Image.fromarray(img, "RGB").transform((64, 48), Image.AFFINE, coeffs))
        assert (want == ois.affine_nearest(img, coeffs, 64, 48)).all(), ("affine", trial)
    img = rng.randint(0, 256, size=(50, 70, 3)).astype(np.uint8)
    pil = Image.fromarray(img, "RGB")
    for f in [0.0, 0.3, 1.0, 1.7, 2.2]:
        assert (np.asarray(ImageEnhance.Brightness(pil).enhance(f)) == ois.apply_color_op(img, ois.OP_BRIGHTNESS, f)).all()
        assert (np.asarray(ImageEnhance.Color(pil).enhance(f)) == ois.apply_color_op(img, ois.OP_SATURATION, f)).all()
        assert (np.asarray(ImageEnhance.Contrast(pil).enhance(f)) == ois.apply_color_op(img, ois.OP_CONTRAST, f)).all()
    print("Pillow %s: blend / L / HSV (exhaustive), GaussianBlur, AFFINE-NEAREST, ImageEnhance restatements bit-exact"
          % Image.__version__)


def run_reference():
    install_shims()
    from handobjectdatasets import handataset  # noqa: E402  (the reference)
    from handobjectdatasets.queries import BaseQueries, TransQueries

    def bkey(name):
        return BaseQueries[name]

    def tkey(name):
        return TransQueries[name]

    out = {}
    for case, (pose_kw, ds_kw, idxs, seed) in poses.CASES.items():
        pose = poses.SeededPoses(as_pil=True, base_key=bkey, trans_key=tkey, point_nb=ds_kw.get("point_nb", 600), **pose_kw)
        queries = [BaseQueries.sides if n == "sides" else TransQueries[n] for n in poses.QUERIES]
        ds = handataset.HandDataset(pose, queries=queries, **ds_kw)
        for idx in idxs:
            np.random.seed(seed * 100 + idx)
            random.seed(seed * 100 + idx)
            s = ds.get_sample(idx)
            tag = "%s/%d/" % (case, idx)
            img = s[TransQueries.images].numpy()
            u8 = np.rint((img.astype(np.float64) + 0.5) * 255).astype(np.uint8)
            # the stored bytes lose nothing: the fp32 image is exactly uint8/255 - 0.5
            assert ((u8.astype(np.float32) / np.float32(255) - np.float32(0.5)).astype(np.float32) == img).all()
            out[tag + "images_u8"] = u8
            out[tag + "affinetrans"] = s[TransQueries.affinetrans].numpy()
            out[tag + "joints2d"] = s[TransQueries.joints2d].numpy()
            out[tag + "joints3d"] = s[TransQueries.joints3d].numpy()
            out[tag + "verts3d"] = np.asarray(s[TransQueries.verts3d])
            out[tag + "objpoints3d"] = s[TransQueries.objpoints3d].numpy()
            out[tag + "center3d"] = np.asarray(s[TransQueries.center3d])
            out[tag + "camintrs"] = np.asarray(s[TransQueries.camintrs])
            out[tag + "side"] = np.array(s[BaseQueries.sides])
    path = os.path.join(HERE, "inputstream.npz")
    np.savez_compressed(path, pillow_version=np.array(Image.__version__), **out)
    print("inputstream.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024, len(out)))


if __name__ == "__main__":
    pin_pillow_primitives()
    run_reference()
