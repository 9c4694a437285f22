"""Debug 2: are leaf gradients of a captured forward + backward (no optimizer in the graph) valid after a replay?  No buckets at all."""
import os, sys, warnings, gc
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ.update(OBMAN_MANO_SYNTHETIC="1")
import torch
from obman_train_amd.networks.handnet import HandNet
from obman_train_amd.synthetic import CONFIGS, make_batch

warnings.simplefilter("ignore")
torch.backends.cudnn.benchmark = False
torch.backends.cudnn.deterministic = True
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = HandNet(**CONFIGS["c3p1"]).to(dev).train()
sample = make_batch(4, dev, seed=20, image_size=64)
for _ in range(2):
    total, results, losses = model.forward(sample)
    model.zero_grad(set_to_none=True)
    total.backward()
eager = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
del total, results, losses
gc.collect()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
keep = os.environ.get("KEEP_OUT", "1") == "1"
with torch.cuda.graph(g):
    total, results, losses = model.forward(sample)
    model.zero_grad(set_to_none=True)
    total.backward()
grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
if not keep:
    del results, losses
for it in range(2):
    g.replay()
    torch.cuda.synchronize()
    bad = []
    for k, w in eager.items():
        err = float((grads[k] - w).abs().max() / w.abs().max().clamp_min(1e-30))
        if not err < 1e-3:
            bad.append((k, err))
    print("replay %d: %d of %d wrong; loss %.4f" % (it, len(bad), len(eager), float(total)), bad[:4], flush=True)
# where do the wrong ones live relative to each other?
ptrs = sorted((v.data_ptr(), v.numel() * 4, k) for k, v in grads.items())
over = [(a[2], b[2]) for a, b in zip(ptrs, ptrs[1:]) if a[0] + a[1] > b[0]]
print("overlapping gradient storages:", over[:6])
names = [k for k, _ in bad]
order = [k for k, _ in model.named_parameters()]
print("wrong indices in parameter order:", [order.index(k) for k in names][:60])
