"""Debug: split-graph data-parallel step with a 1-rank gloo group (host-staged), against the eager step."""
import os, sys, warnings
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OBMAN_MANO_SYNTHETIC="1")
import torch
import torch.distributed as dist

dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
from obman_train_amd.dp import GradientBuckets
from obman_train_amd.dp_selftest import stage_collectives_through_host_if_needed
from obman_train_amd.networks.handnet import HandNet
from obman_train_amd.synthetic import CONFIGS, make_batch
from obman_train_amd.trainer import GraphedTrainStep, make_optimizer

warnings.simplefilter("ignore")
print(stage_collectives_through_host_if_needed(torch.device("cuda", 0)))
torch.backends.cudnn.benchmark = False
torch.backends.cudnn.deterministic = True
torch.manual_seed(0)
model = HandNet(**CONFIGS["c3p1"]).to("cuda:0").train()
opt = make_optimizer(model, "adam", lr=1e-4, capturable=True)
buckets = GradientBuckets(model.parameters(), bucket_bytes=4 * 1024 * 1024, exclude=model.unused_parameters(), force=True)
sample = make_batch(4, torch.device("cuda", 0), seed=20, image_size=64)
total, results, losses = model.forward(sample)
buckets.zero_grad()
total.backward()
buckets.finish()
eager = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
print("eager loss", float(total))
del total, results, losses
step = GraphedTrainStep(model, opt, sample, warmup=1, restore_state=True, buckets=buckets)
print("mode", step.mode)
names = {p: k for k, p in model.named_parameters()}
step.graph.replay()
torch.cuda.synchronize()
bad = 0
for p, g in step._grads.items():
    w = eager[names[p]]
    err = float((g - w).abs().max() / w.abs().max().clamp_min(1e-30))
    if not err < 1e-3:
        bad += 1
        if bad < 8:
            print("static grad after graph A replay differs:", names[p], tuple(p.shape), err, "packed" if buckets.buckets[buckets._where[p]][0] is not None else "direct",
                  "finite" if bool(torch.isfinite(g).all()) else "nonfinite")
print("graph A: %d of %d static gradients wrong; loss %s" % (bad, len(step._grads), float(step.total)))
buckets.exchange(step._grads)
bad = 0
for k, p in model.named_parameters():
    if p.grad is None:
        continue
    err = float((p.grad - eager[k]).abs().max() / eager[k].abs().max().clamp_min(1e-30))
    if not err < 1e-3:
        bad += 1
        if bad < 8:
            print("after exchange differs:", k, err)
print("after exchange: %d wrong" % bad)
