#!/usr/bin/env python
"""Write a MANO-layout pickle (``MANO_RIGHT.pkl`` / ``MANO_LEFT.pkl``) from a parameter pack.

    python tools/make_mano_pickle.py OUT_DIR            # both sides, from the seeded synthetic pack

The licence-gated MANO files cannot ship with this repository; this generator produces files with the SAME container
layout manopth's loader expects (python-2 era pickle read with ``encoding="latin1"``: a dict with ``hands_components``,
``hands_mean``, ``hands_coeffs``, ``f``, ``J_regressor`` as a SciPy sparse matrix, ``kintree_table``, ``weights``,
``posedirs``, ``v_template``, ``bs_style``, ``bs_type`` and ``shapedirs`` as a ``chumpy.ch.Ch`` object) so that
``obman_train_amd.mano_params.load_mano_pickle`` - the only way a user gets real-MANO numbers - is exercised end to end
without chumpy: the ``Ch`` object is emitted through a stand-in class registered under the module name ``chumpy.ch``.
The numbers are the synthetic pack's (real MANO topology, seeded blend shapes), optionally with a non-zero ``hands_mean``.
"""
import os
import pickle
import sys
import types

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _ch_class():
    """A class that pickles as ``chumpy.ch.Ch`` with state {'x': ndarray} (what chumpy stores for a leaf array)."""
    mod_pkg = sys.modules.get("chumpy") or types.ModuleType("chumpy")
    mod = sys.modules.get("chumpy.ch") or types.ModuleType("chumpy.ch")

    class Ch(object):
        def __init__(self, x):
            self.x = np.asarray(x)

        def __getstate__(self):
            return {"x": self.x}

        def __setstate__(self, state):
            self.__dict__.update(state)

    Ch.__module__ = "chumpy.ch"
    Ch.__qualname__ = "Ch"
    mod.Ch = Ch
    mod_pkg.ch = mod
    return Ch, mod_pkg, mod


def write_mano_pickle(path, pack, hands_mean=None, protocol=2):
    """``pack``: dict as returned by ``mano_params.synthetic_mano`` / ``load_mano_pickle``."""
    import scipy.sparse as sp

    Ch, mod_pkg, mod = _ch_class()
    parents = np.asarray(pack["parents"]).astype(np.int64)
    kintree = np.stack([np.where(parents < 0, 4294967295, parents), np.arange(16)]).astype(np.int64)
    mean = np.asarray(pack["hands_mean"] if hands_mean is None else hands_mean, dtype=np.float64)
    raw = {
        "hands_components": np.asarray(pack["hands_components"], dtype=np.float64),
        "hands_mean": mean,
        "hands_coeffs": np.zeros((1, 45)),
        "f": np.asarray(pack["faces"]).astype(np.uint32),
        "J_regressor": sp.csc_matrix(np.asarray(pack["J_regressor"], dtype=np.float64)),
        "kintree_table": kintree,
        "weights": np.asarray(pack["weights"], dtype=np.float64),
        "posedirs": np.asarray(pack["posedirs"], dtype=np.float64),
        "v_template": np.asarray(pack["v_template"], dtype=np.float64),
        "shapedirs": Ch(np.asarray(pack["shapedirs"], dtype=np.float64)),
        "bs_style": "lbs",
        "bs_type": "lrotmin",
    }
    saved = {k: sys.modules.get(k) for k in ("chumpy", "chumpy.ch")}
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod_pkg, mod
    try:
        with open(path, "wb") as fh:
            pickle.dump(raw, fh, protocol=protocol)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return path


def main():
    from obman_train_amd.mano_params import synthetic_mano

    out = sys.argv[1] if len(sys.argv) > 1 else "misc/mano"
    os.makedirs(out, exist_ok=True)
    for side in ("right", "left"):
        rng = np.random.RandomState(5 if side == "right" else 6)
        print(write_mano_pickle(os.path.join(out, "MANO_%s.pkl" % side.upper()), synthetic_mano(side),
                                hands_mean=rng.normal(0, 0.15, size=45)))


if __name__ == "__main__":
    main()
