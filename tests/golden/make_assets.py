"""Dev-container-only: convert the reference's data asset ``assets/contact_zones.pkl``
(hand template verts [778,3] float64 metres, faces [1538,3] uint32, six
contact-zone vertex-id lists; read by the hot path at
``mano_train/networks/branches/contactloss.py:262-265`` via
``handobjectdatasets/contactutils.py:8-14``) into a pickle-free ``.npz`` that
ships with the package, so tests/bench on the GPU box (no /root/reference) have
the zone ids and a MANO-topology hand mesh.

    python tests/golden/make_assets.py            # needs /root/reference
"""
import os
import pickle
import sys

import numpy as np

REF = os.environ.get("OBMAN_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(__file__), "..", "..", "obman_train_amd", "assets", "contact_zones.npz")


def main():
    with open(os.path.join(REF, "assets", "contact_zones.pkl"), "rb") as fh:
        data = pickle.load(fh)
    zones = data["contact_zones"]
    arrs = {
        "verts": np.asarray(data["verts"], dtype=np.float32),
        "faces": np.asarray(data["faces"], dtype=np.int32),
        "zone_sizes": np.asarray([len(zones[k]) for k in sorted(zones)], dtype=np.int32),
        "zone_ids": np.concatenate([np.asarray(zones[k], dtype=np.int32) for k in sorted(zones)]),
    }
    np.savez_compressed(OUT, **arrs)
    print("wrote", os.path.abspath(OUT), {k: v.shape for k, v in arrs.items()})


if __name__ == "__main__":
    sys.exit(main())
