"""fp32 decoder forward + backward twice in one process on the same inputs, with the caching allocator's free blocks filled with NaN
in between: any element that depends on memory the kernels did not write themselves shows up as a NaN or a changed bit.
   gpurun -- 'python tools/r04/det_check.py; OBMAN_DEC_TN3=0 python tools/r04/det_check.py'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from obman_train_amd import ops
from obman_train_amd.icosphere import multi_patch
from obman_train_amd.networks.branches.atlasutils import PointGenCon

torch.manual_seed(0)
B = int(os.environ.get("B", 4))
dec = PointGenCon(bottleneck_size=515, out_factor=200).cuda().train()
dec.mfma_dtype = os.environ.get("DT", "f32")
grid = torch.from_numpy(multi_patch(3, 1)[0].astype(np.float32)).cuda()
feats = torch.randn(B, 512, device="cuda")
cot = torch.randn(B, grid.shape[0], 3, device="cuda")


def poison():
    junk = []
    for k in range(9, 31):
        for _ in range(3 if k < 28 else 1):
            junk.append(torch.full((2 ** k // 4,), float("nan"), device="cuda"))
    torch.cuda.synchronize()
    del junk


def run():
    for p in dec.parameters():
        p.grad = None
    f = feats.clone().requires_grad_()
    out = ops.pointgen_decode(dec, f, grid)
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    return {"out": out.detach().clone(), "features": f.grad.clone(), **{n: p.grad.clone() for n, p in dec.named_parameters()}}


poison()
a = run()
for rep in range(3):
    poison()
    b = run()
    bad = [(n, int((a[n] != b[n]).sum()), int(torch.isnan(b[n]).sum())) for n in a if not torch.equal(a[n], b[n])]
    print("TN3=%s rep %d:" % (os.environ.get("OBMAN_DEC_TN3", "1"), rep), "bit-identical" if not bad else bad)
print("nan in first run:", {n: int(torch.isnan(v).sum()) for n, v in a.items() if torch.isnan(v).any()})
