"""Every A/B knob that switches a kernel generation is pinned by the parity tests it must still pass (VERDICT r04 task 8: "six
env knobs with one of them tested is not a product").

The knobs are read once per process, so each setting runs the c1 = 515 decoder parity cases (fp32 and bf16 flavours, train and
eval BatchNorm, 1 and 25 patches: tests/test_decoder_gpu.py) - or, for the inside test, the contact parity cases - in a child
pytest process with the variable set.  What each value selects:
  OBMAN_DEC_ROWS2F=0   first-generation fp32 rows GEMMs (gemm_rows_kernel) instead of decoder_rows2f.h
  OBMAN_DEC_TN3=0      first-generation fp32 weight-gradient GEMMs instead of decoder_tn3.h
  OBMAN_DEC_F2PQ=0     second-generation layer-2 data gradient with a materialised gy1
  OBMAN_DEC_ROWS2=0    first-generation bf16 rows GEMMs (decoder_bf16.h) instead of decoder_rows2.h / rows3.h
  OBMAN_DEC_ROWS3=0    h3 on the rows2 kernel (register operand queue) instead of the LDS-DMA ring of decoder_rows3.h
  OBMAN_DEC_L4W=0      first-generation layer-4 kernels (32 / 64 lanes per row)
  OBMAN_DEC_TN2W=0     the bf16 layer-2 weight gradient on five 128 x 320 tiles (tn2_bf16_kernel) instead of the round-6 wide tile
  OBMAN_MC_BINNED=0    the all-pairs inside-test kernel behind the product entry points
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_DECODER = ["tests/test_decoder_gpu.py", "-k", "515"]
_CONTACT = ["tests/test_contact_gpu.py", "-k", "contains or golden"]


@pytest.mark.parametrize("knob,target", [
    ("OBMAN_DEC_ROWS2F", _DECODER), ("OBMAN_DEC_TN3", _DECODER), ("OBMAN_DEC_F2PQ", _DECODER), ("OBMAN_DEC_ROWS2", _DECODER),
    ("OBMAN_DEC_ROWS3", _DECODER), ("OBMAN_DEC_L4W", _DECODER), ("OBMAN_DEC_TN2W", _DECODER),
    ("OBMAN_MC_BINNED", _CONTACT)])
def test_fallback_generation_passes_the_parity_cases(knob, target):
    env = dict(os.environ, **{knob: "0"})
    proc = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "--timeout", "600", "-p", "no:cacheprovider"] + target,
                          cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    tail = (proc.stdout + proc.stderr)[-1500:]
    assert proc.returncode == 0, "%s=0:\n%s" % (knob, tail)
    assert " passed" in proc.stdout and "no tests ran" not in proc.stdout, tail
