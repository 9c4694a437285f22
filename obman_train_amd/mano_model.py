"""Pack a MANO parameter pack (``mano_params``) into the device blob ``csrc/mano_lbs.hip`` reads.

Layout (float32 offsets, must match the ``OFF_*`` constants of the kernel): PCA basis [45,45] |
hands_mean [45] | template [2334] | shape basis k-major [10,2334] | pose basis k-major [135,2334] |
J_regressor.template [48] | J_regressor.shapedirs k-major [10,48] | skinning weights joint-major
[16,778] | fingertip ids [5] | palm ids [2] (ids stored as exact floats).  The joint regressor is
folded into the template / shape basis on the host in float64 (it is linear), so the kernel never
touches the 16x778 regressor.
"""
import numpy as np
import torch

from . import _lib


def pack_blob(pack):
    vt = pack["v_template"].astype(np.float64).reshape(2334)
    sd = pack["shapedirs"].astype(np.float64).reshape(2334, 10)
    pd = pack["posedirs"].astype(np.float64).reshape(2334, 135)
    jreg = pack["J_regressor"].astype(np.float64)
    jt = (jreg @ vt.reshape(778, 3)).reshape(48)
    js = np.einsum("jv,vck->kjc", jreg, sd.reshape(778, 3, 10)).reshape(10, 48)
    comps = np.zeros((45, 45))
    hc = pack["hands_components"].astype(np.float64)
    comps[: hc.shape[0]] = hc
    parts = [
        comps.reshape(-1), pack["hands_mean"].astype(np.float64), vt, sd.T.reshape(-1), pd.T.reshape(-1), jt,
        js.reshape(-1), pack["weights"].astype(np.float64).T.reshape(-1), pack["tips"].astype(np.float64),
        pack["palm_ids"].astype(np.float64),
    ]
    blob = np.concatenate(parts).astype(np.float32)
    return blob


class ManoModelBlob:
    """Device-resident model blob (one per hand side)."""

    def __init__(self, pack):
        self.pack = pack
        self.host = torch.from_numpy(pack_blob(pack))
        self.faces = torch.from_numpy(np.asarray(pack["faces"]).astype(np.int64))
        self._dev = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            n = _lib.lib().obman_mano_model_floats()
            if n != self.host.numel():
                raise _lib.ObmanHipError("MANO blob has %d floats, kernel expects %d" % (self.host.numel(), n))
            self._dev[key] = self.host.to(device)
        return self._dev[key]
