"""Build ``csrc/*.hip`` into the in-tree C-ABI library ``csrc/libobman_hip.so`` for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
snapshot.  ``python -m obman_train_amd.build [--force]``.  Every translation unit is compiled to its own
object (in parallel, re-compiled only when it or a header changed) and the objects are linked into one
shared library.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libobman_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "..", "..", "include", "*.h"))


def _newer(deps, target):
    return (not os.path.exists(target)) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def _stale():
    return _newer(sources() + _headers(), LIB)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)


def build_library(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = _headers()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as pool:
        list(pool.map(lambda c: _run(c, verbose), jobs))
    _run([hipcc, "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", LIB + ".tmp"] + objs, verbose)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
