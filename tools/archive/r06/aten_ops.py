"""Which aten operators of one train step launch device kernels OUTSIDE the convolutions and this package's HIP kernels, from where?
   gpurun -- 'CFG=c3 ENC=bf16 DEC=bf16 python tools/archive/r06/aten_ops.py'      (the "other PyTorch kernels" of configs[2])"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

from obman_train_amd.networks.handnet import HandNet
from obman_train_amd.queries import TransQueries
from obman_train_amd.synthetic import CONFIGS, make_batch
from obman_train_amd.trainer import make_optimizer, train_step

dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
model = HandNet(**CONFIGS[os.environ.get("CFG", "c3")]).to(dev).train()
if os.environ.get("ENC", "bf16") == "bf16":
    model.base_net.autocast_dtype = torch.bfloat16
model.atlas_branch.decoder.mfma_dtype = os.environ.get("DEC", "bf16")
opt = make_optimizer(model, os.environ.get("OPT", "adam"), lr=1e-4)
sample = make_batch(64, dev, seed=0, image_size=256)
sample[TransQueries.images] = sample[TransQueries.images].contiguous(memory_format=torch.channels_last)
for _ in range(12):
    train_step(model, opt, sample)
torch.cuda.synchronize()
steps = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(steps):
        train_step(model, opt, sample)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.device_time_total > 0 for c in ev.cpu_children):
        continue  # keep leaves
    stack = [s for s in (ev.stack or []) if "site-packages" not in s and "dist-packages" not in s][:3]
    key = (ev.name, str(ev.input_shapes)[:70], " <- ".join(s.split("/")[-1][:60] for s in stack))
    agg[key][0] += 1
    agg[key][1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
skip = ("convolution", "miopen", "cudnn")
tot = 0.0
byname = collections.defaultdict(lambda: [0.0, 0.0])
for (name, shapes, stack), (n, t) in rows:
    if any(s in name for s in skip):
        continue
    tot += t / steps
    byname[name][0] += n / steps
    byname[name][1] += t / steps
    if t / steps >= 2.0:
        print("%7.1f us/step %5.1f calls/step  %-28s %s\n        %s" % (t / steps, n / steps, name, shapes, stack))
print("non-convolution aten device time per step: %.1f us" % tot)
print("by operator:")
for name, (n, t) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print("%7.1f us/step %6.1f calls/step  %s" % (t, n, name))
