"""Unit icosphere template (vertices + faces).

Replaces ``trimesh.creation.icosphere(subdivisions)`` used at
``mano_train/networks/branches/atlasbranch.py:63-76`` (reference); trimesh is
an un-pinned external dependency, only the geometry matters.  Vertex counts
are 12/42/162/642/2562 for 0..4 subdivisions, faces 20*4**s.  The vertex
*order* is this generator's own: anything that compares ``objpoints3d``
element-wise must feed the same template to both sides.

``multi_patch`` builds the P-patch template of BASELINE.json configs 3/5 (an
extension; the reference has one sphere): P unit spheres at DISTINCT centres of
the decoder's input domain (patch 0 at the origin = the reference's template,
the others on a Fibonacci lattice of radius 3), faces offset per patch.  The
shared PointGenCon therefore maps every patch to its own surface (distinct
inputs), and each patch stays a closed surface, so the inside test is the OR of
the per-patch ray parities (``ops.mesh_contains_hits(..., patches=P)``).
"""
from functools import lru_cache

import numpy as np


def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array(
        [
            [-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
            [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
            [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1],
        ],
        dtype=np.float64,
    )
    f = np.array(
        [
            [0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
            [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
            [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
            [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1],
        ],
        dtype=np.int64,
    )
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v, f


@lru_cache(maxsize=8)
def _icosphere_cached(subdivisions):
    verts, faces = _icosahedron()
    verts = [tuple(p) for p in verts]
    for _ in range(subdivisions):
        midpoint = {}
        new_faces = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            idx = midpoint.get(key)
            if idx is None:
                m = (np.asarray(verts[a]) + np.asarray(verts[b])) * 0.5
                m /= np.linalg.norm(m)
                idx = len(verts)
                verts.append(tuple(m))
                midpoint[key] = idx
            return idx

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new_faces += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        faces = np.asarray(new_faces, dtype=np.int64)
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def icosphere(subdivisions=3):
    """-> (verts [n,3] float64 on the unit sphere, faces [f,3] int64, outward CCW)."""
    v, f = _icosphere_cached(int(subdivisions))
    return v.copy(), f.copy()


PATCH_RING_RADIUS = 3.0  # unit spheres centred >= ~2.1 apart for P <= 25: disjoint in the decoder's input domain


def patch_centres(patches):
    """[P,3] float64: patch 0 at the origin, patches 1..P-1 on a Fibonacci lattice of radius PATCH_RING_RADIUS."""
    c = np.zeros((patches, 3), dtype=np.float64)
    m = patches - 1
    if m > 0:
        k = np.arange(m, dtype=np.float64) + 0.5
        z = 1.0 - 2.0 * k / m
        r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        phi = k * (np.pi * (3.0 - 5.0 ** 0.5))
        c[1:] = PATCH_RING_RADIUS * np.stack([r * np.cos(phi), r * np.sin(phi), z], 1)
    return c


def multi_patch(subdivisions=3, patches=1):
    """P-patch template: verts [P*n,3] (unit sphere + patch centre), faces [P*f,3] (vertex ids offset per patch)."""
    v, f = icosphere(subdivisions)
    if patches == 1:
        return v, f
    n = v.shape[0]
    centres = patch_centres(patches)
    vs = np.concatenate([v + centres[p] for p in range(patches)], 0)
    fs = np.concatenate([f + p * n for p in range(patches)], 0)
    return vs, fs
