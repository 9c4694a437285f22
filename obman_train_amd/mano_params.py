"""MANO model parameter pack.

The reference gets its hand model from the *external* ``manopth.ManoLayer``
(``mano_train/networks/branches/manobranch.py:6,92-105``) which unpickles the
licence-gated ``MANO_{LEFT,RIGHT}.pkl``.  Neither is available, so this module
provides

* ``synthetic_mano(side)`` - a seeded, self-consistent parameter pack with the
  real MANO *topology* (778 verts / 1538 faces from the contact-zone asset),
  16 joints on the MANO kinematic tree, row-stochastic skinning weights,
  convex-combination joint regressor and small random blend shapes.  It is what
  tests, smoke and bench use.  **MANO arithmetic parity with manopth is
  unpinned** (SURVEY §8c); the LBS algorithm follows the published MANO/SMPL
  formulation (Romero et al. 2017, Loper et al. 2015).
* ``load_mano_pickle(path)`` - best-effort loader for a user's real MANO pickle
  (chumpy objects are unpickled through a stand-in class, no chumpy needed).

Pack layout (all float32 numpy, metres):
  v_template [778,3] · shapedirs [778,3,10] · posedirs [778,3,135] ·
  J_regressor [16,778] · weights [778,16] · hands_components [45,45] ·
  hands_mean [45] · faces [1538,3] int32 · parents [16] int32 · tips [5] int32 ·
  palm_ids [2] int32
"""
import os
import pickle

import numpy as np

from .contactzones import TIP_IDXS, hand_template

# MANO kinematic tree: index(1-3) middle(4-6) pinky(7-9) ring(10-12) thumb(13-15)
PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], dtype=np.int32)
# 16 LBS joints + 5 tips -> 21-joint convention (thumb,index,middle,ring,pinky chains)
JOINT_REORDER = np.array([0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20], dtype=np.int32)
TIPS_RIGHT = np.array(TIP_IDXS, dtype=np.int32)          # thumb,index,middle,ring,pinky
TIPS_LEFT = np.array([745, 317, 445, 556, 673], dtype=np.int32)  # as recalled from manopth (one id differs)
PALM_IDS = np.array([95, 22], dtype=np.int32)            # as recalled from manopth; unpinned


def synthetic_mano(side="right", seed=0):
    rng = np.random.RandomState(1234 + seed + (0 if side == "right" else 7))
    verts, faces = hand_template()
    verts = verts.astype(np.float64)
    if side == "left":  # mirror the template so left/right differ as the real model pair does
        verts = verts * np.array([-1.0, 1.0, 1.0])
        faces = faces[:, ::-1].copy()
    tips = TIPS_RIGHT if side == "right" else TIPS_LEFT
    # finger order of the kinematic tree: index, middle, pinky, ring, thumb
    tip_of_chain = [tips[1], tips[2], tips[4], tips[3], tips[0]]
    # wrist = centroid of the open boundary loop (vertices on edges used once)
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    wrist = verts[np.unique(ue[cnt == 1])].mean(0)
    centre = verts.mean(0)
    joints = np.zeros((16, 3))
    joints[0] = wrist
    for c, tip in enumerate(tip_of_chain):
        base = wrist + 0.55 * (centre - wrist) + 0.25 * (verts[tip] - centre)
        for k, frac in enumerate((0.0, 0.4, 0.72)):
            joints[1 + 3 * c + k] = base + frac * (verts[tip] - base)
    # joint regressor: convex combination of the 12 nearest vertices (sparse, rows sum to 1)
    d2 = ((verts[None] - joints[:, None]) ** 2).sum(-1)  # [16,778]
    J_reg = np.zeros((16, 778))
    for j in range(16):
        nn = np.argsort(d2[j])[:12]
        w = np.exp(-d2[j, nn] / (d2[j, nn].mean() + 1e-12))
        J_reg[j, nn] = w / w.sum()
    Jt = J_reg @ verts
    # skinning weights: soft assignment to nearest bones, at most 4 non-zeros per vertex
    dj = ((verts[:, None] - Jt[None]) ** 2).sum(-1)  # [778,16]
    sig = np.median(np.sort(dj, 1)[:, 0]) + 1e-9
    W = np.exp(-dj / (2.0 * sig))
    thresh = np.sort(W, 1)[:, -4][:, None]
    W = np.where(W >= thresh, W, 0.0)
    W /= W.sum(1, keepdims=True)
    shapedirs = rng.normal(0.0, 1.5e-3, size=(778, 3, 10))
    posedirs = rng.normal(0.0, 4e-4, size=(778, 3, 135))
    q, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    comps = q * np.linspace(1.2, 0.15, 45)[:, None]  # decaying PCA basis rows
    pack = dict(
        v_template=verts, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_reg,
        weights=W, hands_components=comps, hands_mean=np.zeros(45),
    )
    pack = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in pack.items()}
    pack.update(
        faces=np.ascontiguousarray(faces, dtype=np.int32), parents=PARENTS.copy(),
        tips=tips.copy(), palm_ids=PALM_IDS.copy(), side=side,
    )
    return pack


class _ChStandIn:
    """Unpickling stand-in for ``chumpy.ch.Ch`` objects found in MANO pickles."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    def __array__(self, dtype=None):
        x = self.__dict__.get("x", self.__dict__.get("_x"))
        return np.asarray(x, dtype=dtype)


class _MANOUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChStandIn
        return super().find_class(module, name)


def load_mano_pickle(path, side="right", flat_hand_mean=True):
    """Real ``MANO_{LEFT,RIGHT}.pkl`` -> parameter pack (same keys as ``synthetic_mano``)."""
    with open(path, "rb") as fh:
        raw = _MANOUnpickler(fh, encoding="latin1").load()
    jreg = raw["J_regressor"]
    jreg = np.asarray(jreg.todense()) if hasattr(jreg, "todense") else np.asarray(jreg)
    pack = dict(
        v_template=np.asarray(raw["v_template"]), shapedirs=np.asarray(raw["shapedirs"]),
        posedirs=np.asarray(raw["posedirs"]), J_regressor=jreg, weights=np.asarray(raw["weights"]),
        hands_components=np.asarray(raw["hands_components"]),
        hands_mean=np.zeros(45) if flat_hand_mean else np.asarray(raw["hands_mean"]),
    )
    pack = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in pack.items()}
    pack.update(
        faces=np.ascontiguousarray(np.asarray(raw["f"]), dtype=np.int32), parents=PARENTS.copy(),
        tips=(TIPS_RIGHT if side == "right" else TIPS_LEFT).copy(), palm_ids=PALM_IDS.copy(), side=side,
    )
    return pack


SYNTHETIC_ROOT = "synthetic"


def get_mano_pack(mano_root="misc/mano", side="right"):
    """The hand model the reference's ``ManoLayer(mano_root=...)`` would load (``manobranch.py:92-105``):
    ``<mano_root>/MANO_{RIGHT,LEFT}.pkl``.  A missing file raises ``FileNotFoundError`` exactly as manopth's ``open`` does -
    a wrong ``mano_root`` (or the default resolved from another working directory) must not silently train on a fake hand.
    The seeded synthetic stand-in is used only when asked for: ``mano_root="synthetic"`` (bench, smoke, CONFIGS), or the
    environment variable ``OBMAN_MANO_SYNTHETIC=1`` as the fallback for a missing file (set by the test suite)."""
    if mano_root == SYNTHETIC_ROOT:
        return synthetic_mano(side)
    fname = os.path.join(mano_root or "", "MANO_%s.pkl" % side.upper())
    if os.path.exists(fname):
        return load_mano_pickle(fname, side=side)
    if os.environ.get("OBMAN_MANO_SYNTHETIC") == "1":
        return synthetic_mano(side)
    raise FileNotFoundError(
        "%s not found: download the MANO models (README of hassony2/manopth) into mano_root, or pass "
        "mano_root=\"synthetic\" for the seeded stand-in model (NOT a real hand)" % fname)
