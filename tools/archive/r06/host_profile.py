"""Where does the HOST spend its time in one eager train step?  cProfile over 30 steps (GPU work is asynchronous: this is enqueue cost).
   gpurun -- 'CFG=c3 ENC=bf16 DEC=bf16 python tools/archive/r06/host_profile.py'"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

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
opt = make_optimizer(model, "adam", lr=1e-4)
sample = make_batch(64, dev, seed=0, image_size=256)
sample[TransQueries.images] = sample[TransQueries.images].contiguous(memory_format=torch.channels_last)
for _ in range(12):
    train_step(model, opt, sample)
torch.cuda.synchronize()
import time

n = 30
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    train_step(model, opt, sample)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host enqueue per step (under cProfile): %.2f ms" % ((t1 - t0) / n * 1e3))
st = pstats.Stats(pr)
st.sort_stats("cumulative")
st.print_stats(45)
