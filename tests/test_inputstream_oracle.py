"""oracle/inputstream.py against the reference's own HandDataset.get_sample outputs (tests/golden/inputstream.npz,
written by tests/golden/make_golden_inputstream.py) and, where Pillow is importable, against Pillow itself."""
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import poses  # noqa: E402
from oracle import inputstream as ois  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "inputstream.npz"))


def u8_to_image(u8):
    return (u8.astype(np.float32) / np.float32(255) - np.float32(0.5)).astype(np.float32)


@pytest.mark.parametrize("case", sorted(poses.CASES))
def test_oracle_get_sample_matches_reference_golden(case):
    pose_kw, ds_kw, idxs, seed = poses.CASES[case]
    pose = poses.SeededPoses(point_nb=ds_kw.get("point_nb", 600), **pose_kw)
    for idx in idxs:
        np.random.seed(seed * 100 + idx)
        random.seed(seed * 100 + idx)
        s = ois.get_sample(pose, idx, poses.QUERIES, **ds_kw)
        tag = "%s/%d/" % (case, idx)
        want_img = u8_to_image(GOLD[tag + "images_u8"])
        assert s["images"].dtype == np.float32 and s["images"].shape == want_img.shape
        assert (s["images"] == want_img).all(), "image bytes differ: %d px" % int((s["images"] != want_img).sum())
        assert (s["affinetrans"] == GOLD[tag + "affinetrans"]).all()
        assert (s["joints2d"] == GOLD[tag + "joints2d"]).all()
        for k in ("joints3d", "verts3d", "objpoints3d", "center3d", "camintrs"):
            np.testing.assert_array_equal(np.asarray(s[k]), GOLD[tag + k], err_msg=k)
        assert str(GOLD[tag + "side"]) == s["sides"]


def test_golden_covers_flip_padding_and_fill():
    """Sanity of the fixture itself: flipped samples exist, black padding and out-of-source fill produce exact 0 bytes."""
    assert {str(GOLD["fhb_like_train/%d/side" % i]) for i in range(4)} == {"left"}
    pad = GOLD["obman_like_mesh_pad/0/images_u8"]
    assert (pad[:, :19, :] == 0).all() and (pad[:, :, :19] == 0).all()
    assert any((GOLD["strong_jitter/%d/images_u8" % i] == 0).all(0).any() for i in range(3))


def test_pillow_restatements_bit_exact_when_pillow_present():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFilter

    rng = np.random.RandomState(5)
    allv = np.arange(0, 1 << 24, 7, dtype=np.uint32)  # every 7th colour here; the generator checks all 2^24
    cols = np.stack([(allv >> 16) & 255, (allv >> 8) & 255, allv & 255], -1).astype(np.uint8).reshape(-1, 1, 3)
    assert (np.asarray(Image.fromarray(cols, "RGB").convert("HSV")) == ois.rgb2hsv(cols)).all()
    assert (np.asarray(Image.fromarray(cols, "HSV").convert("RGB")) == ois.hsv2rgb(cols)).all()
    assert (np.asarray(Image.fromarray(cols, "RGB").convert("L")) == ois.luma(cols)).all()
    img = rng.randint(0, 256, size=(61, 83, 3)).astype(np.uint8)
    for sigma in (0.11, 0.49, 1.3, 2.6):
        assert (np.asarray(Image.fromarray(img, "RGB").filter(ImageFilter.GaussianBlur(sigma))) == ois.gaussian_blur(img, sigma)).all()
    for _ in range(20):
        rot, sc = rng.uniform(-np.pi, np.pi), rng.uniform(0.4, 2.5)
        c, s = np.cos(rot) * sc, np.sin(rot) * sc
        coeffs = (c, -s, rng.uniform(-20, 83), s, c, rng.uniform(-20, 61))
        assert (np.asarray(Image.fromarray(img, "RGB").transform((40, 30), Image.AFFINE, coeffs)) == ois.affine_nearest(img, coeffs, 40, 30)).all()
    assert PIL is not None
