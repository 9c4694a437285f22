"""Host enqueue time vs device time of one train step (is the launch path close to being the bottleneck?)."""
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from obman_train_amd.networks.handnet import HandNet  # noqa: E402
from obman_train_amd.synthetic import CONFIGS, make_batch  # noqa: E402
from obman_train_amd.trainer import make_optimizer, train_step  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
model = HandNet(**CONFIGS[name]).to(dev).train()
opt = make_optimizer(model)
sample = make_batch(64, dev)
for _ in range(10):
    train_step(model, opt, sample)
torch.cuda.synchronize()
host, total = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step(model, opt, sample)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
print(json.dumps({"config": name, "host_enqueue_ms": 1e3 * sorted(host)[10], "step_ms": 1e3 * sorted(total)[10]}))
