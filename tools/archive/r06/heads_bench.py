"""The small Linear layers of the MANO / Atlas heads at bs 64: F.linear (addmm with a bias epilogue: hipBLASLt) against mm + add, per shape.
   gpurun -- 'python tools/archive/r06/heads_bench.py'"""
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
shapes = [(512, 1024), (1024, 256), (256, 33), (256, 10), (512, 256), (256, 3), (256, 1)]


def t(fn, n=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:  # noqa: BLE001
            print(lib, "unavailable", e)
            continue
    for K, N in shapes:
        x = torch.randn(64, K, device=dev)
        W = torch.randn(N, K, device=dev, requires_grad=True)
        b = torch.randn(N, device=dev, requires_grad=True)
        gy = torch.randn(64, N, device=dev)
        lin = t(lambda: F.linear(x, W, b))
        mm = t(lambda: torch.mm(x, W.t()) + b)
        bwd_x = t(lambda: torch.mm(gy, W))
        bwd_w = t(lambda: torch.mm(gy.t(), x))
        print("%-9s K %4d N %4d | linear %6.1f us | mm + add %6.1f us | dX %6.1f us | dW %6.1f us" % (lib, K, N, lin, mm, bwd_x, bwd_w), flush=True)
