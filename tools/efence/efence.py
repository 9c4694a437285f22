#!/usr/bin/env python
"""Memory-safety runs of the HIP path under the guard-page allocator (tools/efence/efence_alloc.cpp).  TEST INFRASTRUCTURE.

    python tools/efence/efence.py --selftest                       # the tool itself: in-range accesses pass, 1 byte past faults
    python tools/efence/efence.py --config c3 --batch 64 --encoder-dtype bf16 --decoder-dtype bf16 --steps 2
    python tools/efence/efence.py --left ...                       # guard BEFORE every tensor instead of after it

Every tensor of the process (inputs, outputs, workspaces, autograd-saved state, MIOpen workspaces) sits right-aligned against an
unmapped page, and every launcher of the C-ABI is traced + synchronised (OBMAN_TRACE_LAUNCH=1), so an out-of-bounds access
kills the process with "Memory access fault by GPU" right after the stderr line naming the launch that did it.  Exit code 0 and
the final "[efence] clean" line = no access outside any buffer, no canary touched.
"""
import argparse
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
LIB = os.path.join(HERE, "libefence.so")
SRC = os.path.join(HERE, "efence_alloc.cpp")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", tmp],
                       check=True)
        os.replace(tmp, LIB)
    return LIB


def install(left=False, align=None):
    """Route every device allocation of this process through the guard-page allocator.  Must run before the first device
    allocation."""
    import torch

    if left:
        os.environ["OBMAN_EFENCE_LEFT"] = "1"
    if align:
        os.environ["OBMAN_EFENCE_ALIGN"] = str(int(align))
    os.environ.setdefault("OBMAN_TRACE_LAUNCH", "1")
    alloc = torch.cuda.memory.CUDAPluggableAllocator(build(), "efence_malloc", "efence_free")
    torch.cuda.memory.change_current_allocator(alloc)
    return ctypes.CDLL(LIB)


def stats(handle):
    out = (ctypes.c_size_t * 6)()
    handle.efence_stats(out)
    return {"allocations": out[0], "frees": out[1], "live": out[2], "peak_mapped_bytes": out[3], "canary_hits": out[4],
            "released_from_quarantine": out[5]}


def _probe(handle, t, beyond, write, sink):
    import torch

    handle.efence_probe.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = handle.efence_probe(t.data_ptr(), t.numel() * t.element_size(), beyond, write, sink.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rc


def selftest_child(mode):
    import torch

    handle = install(left=mode.startswith("left"))
    t = torch.zeros(4096 + 48, dtype=torch.uint8, device="cuda")  # deliberately not a page multiple
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    if mode == "inside":
        assert _probe(handle, t, 0, 0, sink) == 0 and _probe(handle, t, 0, 1, sink) == 0
        assert int(t[-1]) == 0x5a
        print("[efence] selftest inside: ok", file=sys.stderr)
    elif mode == "read_past":
        _probe(handle, t, 1, 0, sink)      # first byte past the end: must fault
        print("[efence] selftest read_past: NOT DETECTED", file=sys.stderr)
    elif mode == "write_past":
        _probe(handle, t, 1, 1, sink)
        print("[efence] selftest write_past: NOT DETECTED", file=sys.stderr)
    elif mode == "left_read_before":
        _probe(handle, t, -1, 0, sink)
        print("[efence] selftest left_read_before: NOT DETECTED", file=sys.stderr)
    elif mode == "write_before":          # right-aligned mode: lands in mapped slack -> canary, detected at free
        _probe(handle, t, -1, 1, sink)
        del t
        print("[efence] selftest write_before: NOT DETECTED", file=sys.stderr)
    sys.stderr.flush()
    os._exit(0)


def selftest():
    """Each case in its own process (a detected access kills it)."""
    ok = True
    for mode, must_die in (("inside", False), ("read_past", True), ("write_past", True), ("left_read_before", True),
                           ("write_before", True)):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--selftest-child", mode], capture_output=True, text=True,
                           timeout=600)
        died = p.returncode != 0
        tail = (p.stderr or "").strip().splitlines()[-3:]
        print("[efence] selftest %-18s exit %4d  %s   | %s" % (mode, p.returncode, "ok" if died == must_die else "WRONG", " / ".join(tail)))
        ok &= died == must_die
    return ok


def run_steps(args):
    import warnings

    import torch

    handle = install(left=args.left, align=args.align)
    warnings.simplefilter("ignore")
    os.environ.setdefault("OBMAN_MANO_SYNTHETIC", "1")
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.queries import TransQueries
    from obman_train_amd.synthetic import CONFIGS, make_batch
    from obman_train_amd.trainer import make_optimizer, train_step

    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False  # the find phase allocates and frees hundreds of workspaces: slow under this allocator
    torch.manual_seed(0)
    model = HandNet(**CONFIGS[args.config]).to(dev).train()
    if args.encoder_dtype == "bf16":
        model.base_net.autocast_dtype = torch.bfloat16
    model.atlas_branch.decoder.mfma_dtype = args.decoder_dtype
    opt = make_optimizer(model, "adam", lr=1e-4, capturable=True)
    sample = make_batch(args.batch, dev, seed=0, image_size=args.image_size)
    sample[TransQueries.images] = sample[TransQueries.images].contiguous(memory_format=torch.channels_last)
    total = None
    for i in range(args.steps):
        print("[efence] step %d" % i, file=sys.stderr, flush=True)
        total, _, _ = train_step(model, opt, sample)
        torch.cuda.synchronize()
        print("[efence] step %d total loss %.6f" % (i, float(total)), file=sys.stderr, flush=True)
    if args.eval:
        model.eval()
        with torch.no_grad():
            model.forward(sample)
        torch.cuda.synchronize()
    del total, model, opt, sample
    import gc

    gc.collect()
    torch.cuda.synchronize()
    st = stats(handle)
    print("[efence] clean: %s config=%s batch=%d enc=%s dec=%s %s-aligned" % (st, args.config, args.batch, args.encoder_dtype,
                                                                                args.decoder_dtype, "left" if args.left else "right"),
          file=sys.stderr, flush=True)
    os._exit(0 if st["canary_hits"] == 0 else 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--selftest-child", default=None)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--encoder-dtype", default="f32")
    ap.add_argument("--decoder-dtype", default="f32")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--eval", action="store_true", help="also one eval-mode forward")
    ap.add_argument("--left", action="store_true")
    ap.add_argument("--align", type=int, default=None)
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE set before the HIP library loads (A/B knobs)")
    args = ap.parse_args()
    for kv in args.env:
        k, _, v = kv.partition("=")
        os.environ[k] = v
    if args.selftest_child:
        selftest_child(args.selftest_child)
    if args.selftest:
        sys.exit(0 if selftest() else 1)
    run_steps(args)


if __name__ == "__main__":
    main()
