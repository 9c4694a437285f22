import os
import sys

import pytest

# no MANO files in any test environment: HandNet(mano_root="misc/mano") falls back to the seeded synthetic model here only
os.environ.setdefault("OBMAN_MANO_SYNTHETIC", "1")
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must FAIL (not skip) on a GPU box whose HIP library is missing; on a
    CPU-only box they are deselected by ``-m "not gpu"`` and skipped if selected anyway."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    from tests.golden.common import GOLDEN_DIR

    def _load(name):
        return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)

    return _load


def record_measurement(name, values):
    """Append measured parity errors to gpurun_out/parity_measured.jsonl (scratch; copied to profiles/ by hand) so that a
    tolerance can be frozen at a small multiple of what the hardware actually produced.  Never fails a test."""
    import json

    try:
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.jsonl"), "a") as fh:
            fh.write(json.dumps({"test": name, **values}) + "\n")
    except OSError:
        pass
