"""Run-to-run determinism of one HandNet forward + backward (same weights, same batch, one process): losses must be bit-identical
(no atomics in any forward kernel); gradients may differ only where MIOpen's weight / data gradient kernels accumulate with atomics.
   gpurun -- 'python tools/r04/step_det.py'"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

warnings.simplefilter("ignore")
from obman_train_amd.networks.handnet import HandNet
from obman_train_amd.synthetic import CONFIGS, make_batch

dev = torch.device("cuda:0")
cfg = os.environ.get("CFG", "c3p1")
torch.manual_seed(0)
model = HandNet(**CONFIGS[cfg]).to(dev).train()
sample = make_batch(int(os.environ.get("B", 4)), dev, seed=30, image_size=int(os.environ.get("RES", 64)))


def run():
    model.zero_grad(set_to_none=True)
    total, results, losses = model(sample)
    total.backward()
    torch.cuda.synchronize()
    return ({k: v.detach().clone() for k, v in losses.items() if torch.is_tensor(v)},
            {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})


run()
la, ga = run()
for rep in range(3):
    lb, gb = run()
    dl = {k: float((la[k].float() - lb[k].float()).abs().max() / (la[k].float().abs().max() + 1e-30)) for k in la if not torch.equal(la[k], lb[k])}
    groups = {}
    for n in ga:
        if not torch.equal(ga[n], gb[n]):
            key = n.split(".")[0] + ("." + n.split(".")[1] if n.startswith("base_net") else "")
            rel = float((ga[n] - gb[n]).abs().max() / (ga[n].abs().max() + 1e-30))
            groups[key] = max(groups.get(key, 0.0), rel)
    print("rep %d: losses %s; gradient groups that differ (max |diff| / max |g|): %s" % (rep, dl or "bit-identical", groups or "none"))
print("total", float(la["total_loss"]))
