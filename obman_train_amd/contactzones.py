"""Contact-zone table used by ``contact_zones="zones"``.

Mirrors ``handobjectdatasets/contactutils.py:8-14`` (reference), which
``compute_contact_loss`` calls with the *relative* path
``assets/contact_zones.pkl`` (``contactloss.py:262-265``).  A caller running
from a reference checkout has that pickle in its cwd and it is honoured; else
the packaged pickle-free conversion (``tests/golden/make_assets.py``) is used.
"""
import os
import pickle
from functools import lru_cache

import numpy as np

_PACKAGED = os.path.join(os.path.dirname(__file__), "assets", "contact_zones.npz")


@lru_cache(maxsize=16)
def load_contacts(save_contact_paths="assets/contact_zones.pkl"):
    """-> (hand_verts [778,3] metres, {zone_idx: [vertex ids]})."""
    if save_contact_paths and save_contact_paths.endswith(".pkl") and os.path.exists(save_contact_paths):
        with open(save_contact_paths, "rb") as fh:
            data = pickle.load(fh)
        zones = {int(k): [int(i) for i in v] for k, v in data["contact_zones"].items()}
        return np.asarray(data["verts"]), zones
    data = np.load(_PACKAGED)
    sizes = data["zone_sizes"]
    ids = data["zone_ids"]
    zones, start = {}, 0
    for z, n in enumerate(sizes):
        zones[z] = [int(i) for i in ids[start:start + int(n)]]
        start += int(n)
    return data["verts"].astype(np.float64), zones


@lru_cache(maxsize=1)
def hand_template():
    """-> (verts [778,3] float32 metres, faces [1538,3] int32) MANO-topology hand mesh."""
    data = np.load(_PACKAGED)
    return data["verts"].astype(np.float32), data["faces"].astype(np.int32)


TIP_IDXS = (745, 317, 444, 556, 673)  # contactloss.py:258
