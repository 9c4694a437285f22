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
# measurement builds only (e.g. OBMAN_EXTRA_HIPCC_FLAGS=-DOBMAN_F2_TIMING with --force): never set for the product library
FLAGS += os.environ.get("OBMAN_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "..", "..", "include", "*.h"))


def _newer(deps, target):
    return (not os.path.exists(target)) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


STAMP = os.path.join(OBJ, ".flags")  # the flag set the objects / library were built with: a measurement build (extra -D flags) must
#                                       not survive as "fresh" once the flags are back to the product's (mtimes alone cannot tell)


def _flags_changed():
    try:
        with open(STAMP) as fh:
            return fh.read() != " ".join(FLAGS)
    except OSError:
        return True


def _stale():
    return _flags_changed() or _newer(sources() + _headers(), LIB)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)


def build_library(force=False, verbose=False):
    """Compile + link under an exclusive file lock: the ranks of one ``torch.distributed.run`` launch that all find the
    library stale take turns (the first one builds, the others find it fresh), and every output is written to a
    process-private temporary and renamed into place, so nobody ever links or loads a half-written file."""
    if not force and not _stale():
        return LIB
    import fcntl

    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():  # another process built it while this one waited
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = _headers()
    tag = ".tmp.%d" % os.getpid()
    force = force or _flags_changed()  # every object was compiled with another flag set
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append((obj, [hipcc] + FLAGS + ["-c", src, "-o", obj + tag]))

    def compile_one(job):
        obj, cmd = job
        _run(cmd, verbose)
        os.replace(obj + tag, obj)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as pool:
        list(pool.map(compile_one, jobs))
    _run([hipcc, "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", LIB + tag] + objs, verbose)
    os.replace(LIB + tag, LIB)
    with open(STAMP + tag, "w") as fh:
        fh.write(" ".join(FLAGS))
    os.replace(STAMP + tag, STAMP)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
