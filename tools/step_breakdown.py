"""Steady-state breakdown of one C2/C3 train step on the GPU box (CUDA-event timed, 20 iters each)."""
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")

from obman_train_amd.networks.handnet import HandNet  # noqa: E402
from obman_train_amd.queries import TransQueries  # noqa: E402
from obman_train_amd.synthetic import CONFIGS, make_batch  # noqa: E402
from obman_train_amd.trainer import make_optimizer  # noqa: E402


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = HandNet(**CONFIGS[name]).to(dev).train()
    opt = make_optimizer(model)
    sample = make_batch(B, dev)
    img = sample[TransQueries.images]
    out = {"config": name, "batch": B}

    def resnet_fwd():
        with torch.no_grad():
            model.base_net(img)

    def resnet_fwdbwd():
        f, _ = model.base_net(img)
        f.sum().backward()

    def full_fwd():
        with torch.no_grad():
            model.forward(sample)

    def full_fwdbwd():
        total, _, _ = model.forward(sample)
        opt.zero_grad(set_to_none=True)
        total.backward()

    def full_step():
        total, _, _ = model.forward(sample)
        opt.zero_grad(set_to_none=True)
        total.backward()
        opt.step()

    full_step()
    out["resnet_fwd_ms"] = timeit(resnet_fwd)
    print(json.dumps(out), flush=True)
    out["resnet_fwdbwd_ms"] = timeit(resnet_fwdbwd)
    out["full_fwd_ms"] = timeit(full_fwd)
    out["full_fwdbwd_ms"] = timeit(full_fwdbwd)
    out["full_step_ms"] = timeit(full_step)
    out["adam_ms"] = timeit(opt.step)
    # NCHW variant of the encoder for comparison
    x_nchw = img.contiguous()
    net = model.base_net

    def resnet_nchw():
        y = net.maxpool(net.relu(net.bn1(net.conv1(x_nchw))))
        y = net.layer4(net.layer3(net.layer2(net.layer1(y))))
        y.mean(3).mean(2).sum().backward()

    out["resnet_fwdbwd_nchw_ms"] = timeit(resnet_nchw)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
