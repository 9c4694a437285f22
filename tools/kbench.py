"""Per-kernel micro-benchmark on the GPU box (HIP events on the launch stream).

    python tools/kbench.py [chamfer|contains|mano|decoder|all]
Prints one JSON line per measurement: algorithmic bytes/flops / average time."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch


def timeit(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e-3 / iters


def bench_chamfer():
    from obman_train_amd import ops

    for B, n_p, n_g in ((64, 642, 600), (64, 16050, 600), (64, 64050, 600), (64, 2562, 600)):
        p = (torch.randn(B, n_p, 3, device="cuda") * 40).requires_grad_()
        g = torch.randn(B, n_g, 3, device="cuda") * 40
        t_f = timeit(lambda: ops.chamfer(p, g))
        l1, l2 = ops.chamfer(p, g)
        loss = (l1 + l2).mean()
        t_b = timeit(lambda: torch.autograd.grad(loss, p, retain_graph=True))
        pairs = 2.0 * B * n_p * n_g  # both directions evaluate every pair once
        print(json.dumps(dict(
            kernel="chamfer", B=B, n_pred=n_p, n_gt=n_g, fwd_us=t_f * 1e6, bwd_us=t_b * 1e6,
            fwd_alg_GBps=20.0 * (n_p + n_g) * B / t_f / 1e9, fwd_Gpairs_per_s=pairs / t_f / 1e9,
            fwd_valu_TFLOPs=pairs * 8 / t_f / 1e12)))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.zeros(1, device="cuda")
    if which in ("chamfer", "all"):
        bench_chamfer()
