"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol ``include/obman_hip.h`` declares.  No compute call is made (no GPU here)."""
import os

import pytest

from tests.conftest import REPO


def test_library_builds_and_exports_every_declared_symbol():
    from obman_train_amd import _lib
    from obman_train_amd.build import build_library

    path = build_library()
    assert os.path.exists(path)
    handle = _lib.lib()
    declared = _lib.declared_symbols(os.path.join(REPO, "include", "obman_hip.h"))
    assert len(declared) >= 6
    for name in declared:
        assert hasattr(handle, name), "symbol %s declared in obman_hip.h but not exported" % name
        assert name in _lib._SIGNATURES, "symbol %s has no ctypes signature in _lib.py" % name
    assert handle.obman_abi_version() == _lib.ABI_VERSION


def test_ops_refuse_cpu_tensors():
    import torch

    from obman_train_amd import _lib, ops

    x = torch.zeros(1, 4, 3)
    with pytest.raises(_lib.ObmanHipError):
        ops.chamfer(x, x)
    with pytest.raises(_lib.ObmanHipError):
        ops.pairmin(x, x)
