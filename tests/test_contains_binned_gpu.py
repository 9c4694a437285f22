"""K4, round 5: the grid-culled inside test (csrc/contains.hip, contains_binned_kernel - what obman_mesh_contains_fwd /
obman_mesh_contains_groups_fwd run) must return hit words BIT-IDENTICAL to the all-pairs kernel
(obman_mesh_contains_bruteforce_fwd, the formulation of contactutils.py:62-159 and the round 1-4 product kernel): both
evaluate a pair with the same fp32 operations, and the binned kernel may only skip pairs whose evaluation cannot pass.

* every scene family of tests/test_contact_gpu.py (blobs, multi-patch, grouped mode, the bench shapes);
* > 1e5 random scenes in one launch per family: slivers, triangles that contain the ray direction, tiny triangles around the
  parallel threshold, query points placed ON projected triangle borders within a few ulp, far outliers, huge offsets from the
  origin, duplicated points, NaN / inf in points and vertices, degenerate (repeated-vertex) faces;
* the tile / chunk geometry: P across the 1024-point tile boundary, F across the chunk boundaries, triangle splits.
"""
import numpy as np
import pytest
import torch

from obman_train_amd.icosphere import icosphere, multi_patch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _both(points, verts, faces, patches=1):
    from obman_train_amd import ops

    a = ops.mesh_contains_hits(points, verts, faces, patches=patches, raw_bits=True)
    b = ops.mesh_contains_hits(points, verts, faces, patches=patches, raw_bits=True, all_pairs=True)
    return a, b


def _assert_same(points, verts, faces, patches=1):
    a, b = _both(points, verts, faces, patches)
    if not torch.equal(a, b):
        bad = (a != b).nonzero()
        raise AssertionError("binned != all-pairs at %d of %d points, first %s: %s vs %s" % (
            bad.shape[0], a.numel(), bad[0].tolist(), a[tuple(bad[0])].item(), b[tuple(bad[0])].item()))
    return a


def _blob(subdiv, B, seed, radius=40.0, patches=1):
    rng = np.random.RandomState(seed)
    v, f = multi_patch(subdiv, patches)
    scale = radius * (1.0 + 0.3 * np.sin(4.0 * v[:, :1]) * np.cos(3.0 * v[:, 1:2]))
    pts = v[None] * scale[None] * rng.uniform(0.6, 1.4, size=(B, 1, 3))
    if patches > 1:
        n = v.shape[0] // patches
        for p in range(patches):
            pts[:, p * n:(p + 1) * n] += rng.normal(0, 25.0, size=(B, 1, 3))
    return T(pts.astype(np.float32)), f.astype(np.int32)


@pytest.mark.parametrize("B,P,subdiv,patches", [(1, 1, 0, 1), (2, 100, 1, 1), (3, 778, 2, 1), (2, 1500, 2, 3), (2, 70, 3, 1),
                                                (64, 778, 3, 1), (4, 778, 3, 25), (3, 1024, 2, 1), (3, 1025, 2, 1),
                                                (2, 5000, 2, 2)])
def test_binned_equals_all_pairs_on_blobs(B, P, subdiv, patches):
    verts, faces = _blob(subdiv, B, 3, patches=patches)
    rng = np.random.RandomState(4)
    origins = T(rng.normal(0, 35, size=(B, P, 3)).astype(np.float32))
    if P > 2:
        origins[:, 0] = 1e4  # a far outlier stretches the grid
    fc = T(faces).cuda()
    hits = _assert_same(origins.cuda(), verts.cuda(), fc)
    assert P < 50 or 0.02 < ((hits & 1) == 1).float().mean().item() < 0.98  # both classes present
    if patches > 1:
        bits = _assert_same(origins.cuda(), verts.cuda(), fc, patches=patches)
        assert int(bits.max().item()) < (1 << patches)


def test_binned_equals_all_pairs_at_bench_shapes():
    """configs[2] (64 x 778 x 32 000 faces, 25 patches) and configs[4] (128 000 faces) shapes: a hand-sized point cloud next
    to a multi-patch object, and the random-initialisation case the bench runs (the whole object inside a few mm)."""
    rng = np.random.RandomState(11)
    for subdiv, B in ((3, 64), (4, 8)):
        v, f = multi_patch(subdiv, 25)
        for obj_scale in (60.0, 0.5):
            verts = (v[None] * obj_scale * rng.uniform(0.7, 1.3, size=(B, 1, 3))).astype(np.float32)
            n = v.shape[0] // 25
            for p in range(25):
                verts[:, p * n:(p + 1) * n] += rng.normal(0, obj_scale * 0.4, size=(B, 1, 3)).astype(np.float32)
            hand = rng.normal(0, 45, size=(B, 778, 3)).astype(np.float32) * np.array([1.0, 0.5, 0.25], np.float32)
            hand += rng.normal(0, 20, size=(B, 1, 3)).astype(np.float32)
            pts, vv, fc = T(hand).cuda(), T(verts).cuda(), T(f.astype(np.int32)).cuda()
            _assert_same(pts, vv, fc)
            _assert_same(pts, vv, fc, patches=25)


def _random_scenes(rng, B, P, F, Nv):
    """B small scenes with adversarial content (see the module docstring)."""
    scale = 10.0 ** rng.uniform(-3, 2.5, size=(B, 1, 1))
    off = rng.normal(0, 1, size=(B, 1, 3)) * scale * 10.0 ** rng.uniform(-1, 2, size=(B, 1, 1))
    verts = rng.normal(0, 1, size=(B, Nv, 3)) * scale + off
    faces = rng.randint(0, Nv, size=(F, 3))
    faces[: F // 16, 1] = faces[: F // 16, 0]  # degenerate faces (repeated vertex)
    fam = rng.randint(0, 6, size=(B, F))
    # per-scene per-face modifications need per-scene vertices: give every face its own three vertices
    Nv2 = 3 * F
    a = verts[:, faces[:, 0]]
    d1 = verts[:, faces[:, 1]] - a
    d2 = verts[:, faces[:, 2]] - a
    ray = np.array([0.4395064455, 0.617598629942, 0.652231566745])
    sl = 10.0 ** rng.uniform(-6, 0, size=(B, F, 1))
    d2 = np.where((fam == 1)[..., None], d1 * rng.uniform(0.2, 2, size=(B, F, 1)) + d2 * sl, d2)           # slivers
    d2 = np.where((fam == 2)[..., None], ray * np.linalg.norm(d1, axis=2, keepdims=True) + d2 * sl, d2)     # edge-on
    tiny = np.where(fam == 3, 10.0 ** rng.uniform(-4.5, -2.0, size=(B, F)), 1.0)[..., None]                  # near tol
    d1, d2 = d1 * tiny, d2 * tiny
    v2 = np.stack([a, a + d1, a + d2], 2).reshape(B, Nv2, 3).astype(np.float32)
    f2 = np.arange(Nv2, dtype=np.int32).reshape(F, 3)
    # points: on the borders of random faces (exact barycentrics of the fp32 vertices, fp64), nudged by a few ulp
    v64 = v2.astype(np.float64).reshape(B, F, 3, 3)
    pick = rng.randint(0, F, size=(B, P))
    bi = np.arange(B)[:, None]
    A, e1, e2 = v64[bi, pick, 0], v64[bi, pick, 1] - v64[bi, pick, 0], v64[bi, pick, 2] - v64[bi, pick, 0]
    kind = rng.randint(0, 6, size=(B, P))
    ca = rng.uniform(-0.3, 1.3, size=(B, P))
    cb = rng.uniform(-0.3, 1.3, size=(B, P))
    ca = np.where(kind == 0, 0.0, ca)
    cb = np.where(kind == 1, 0.0, cb)
    cb = np.where(kind == 2, 1.0 - ca, cb)
    ca = np.where(kind == 3, rng.choice([0.0, 1.0], size=(B, P)), ca)
    cb = np.where(kind == 3, np.where(ca == 0, rng.choice([0.0, 1.0], size=(B, P)), 0.0), cb)
    nud = rng.choice([0, 0, 1, -1, 4, -4, 64, -64], size=(B, P, 2)) * 2.0 ** -24
    ca = ca + nud[..., 0] * np.maximum(np.abs(ca), 1e-3)
    cb = cb + nud[..., 1] * np.maximum(np.abs(cb), 1e-3)
    s = -np.abs(rng.normal(0, 1, size=(B, P, 1))) * scale * 10.0 ** rng.uniform(-1, 1.5, size=(B, P, 1))
    pts = A + ca[..., None] * e1 + cb[..., None] * e2 + s * ray
    pts = np.where((kind == 5)[..., None], rng.normal(0, 1, size=(B, P, 3)) * scale * 3 + off, pts)  # anywhere
    pts = pts.astype(np.float32)
    pts[:, 1] = pts[:, 0]                                  # duplicated points
    pts[::7, 2] *= 1e3                                     # far outliers
    return pts, v2, f2


@pytest.mark.parametrize("seed,B,P,F", [(0, 40000, 24, 32), (1, 40000, 48, 20), (2, 30000, 8, 64)])
def test_binned_equals_all_pairs_on_random_scenes(seed, B, P, F):
    """110 000 scenes over the three parametrisations, each launched as one batch."""
    rng = np.random.RandomState(seed)
    pts, verts, faces = _random_scenes(rng, B, P, F, Nv=max(8, F // 2))
    # non-finite input in a few scenes
    pts[5, 3, 1] = np.nan
    pts[6, 2, 0] = np.inf
    verts[7, 4, 2] = np.nan
    verts[8, 1, 0] = -np.inf
    p, v, f = T(pts).cuda(), T(verts).cuda(), T(faces).cuda()
    hits = _assert_same(p, v, f)
    frac = ((hits & 1) == 1).float().mean().item()
    assert 0.01 < frac < 0.9, frac   # the border-hugging points do produce hits
    if F % 4 == 0:
        _assert_same(p, v, f, patches=4)


def test_binned_tile_and_chunk_boundaries():
    """P around the 1024-point tile, F around the 256-triangle lane stride and the chunk split, B = 1 (maximum split)."""
    rng = np.random.RandomState(5)
    v, f = icosphere(3)
    for P in (1, 2, 1023, 1024, 1025, 2049):
        for F in (1, 2, 255, 256, 257, 511, 513, 1280):
            for B in (1, 3):
                verts = (v[None] * 30.0 * rng.uniform(0.8, 1.2, size=(B, 1, 3))).astype(np.float32)
                pts = rng.normal(0, 25, size=(B, P, 3)).astype(np.float32)
                _assert_same(T(pts).cuda(), T(verts).cuda(), T(f[:F].astype(np.int32)).cuda())
    # empty inputs
    from obman_train_amd import ops

    z = ops.mesh_contains_hits(torch.zeros(2, 5, 3).cuda(), T(verts).cuda()[:2].expand(2, -1, -1).contiguous(),
                               torch.zeros(0, 3, dtype=torch.int32).cuda())
    assert z.shape == (2, 5) and int(z.abs().sum().item()) == 0
