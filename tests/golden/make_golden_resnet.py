"""DEV-CONTAINER ONLY - golden vectors of the reference's own image encoder.

    python tests/golden/make_golden_resnet.py

``/root/reference/mano_train/networks/bases/resnet.py`` imports as-is (no shims needed).  The reference's
``resnet18`` / ``resnet50`` are built with ``pretrained=False``, their weights replaced by ``seeded_state`` (regenerated
from a numpy seed on the test side, so the fixture holds no weights), and run forward + backward at 64x64 in train and eval
mode.  Stored: the input, the features, the input gradient, a few parameter gradients (large ones sub-sampled, see ``subsample``) and updated BatchNorm running
statistics, and the state-dict layout (names + shapes).  Data only; nothing of the reference's text is kept.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("OBMAN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from tests.golden.common import load_seeded, subsample  # noqa: E402

SEED = {"resnet18": 1801, "resnet50": 5001}
PROBES = {  # parameter gradients / running statistics worth pinning: first, a strided stage entry, last
    "resnet18": ["conv1.weight", "bn1.weight", "layer2.0.downsample.0.weight", "layer2.0.downsample.1.bias",
                 "layer3.1.conv2.weight", "layer4.1.bn2.weight", "layer4.1.bn2.bias"],
    "resnet50": ["conv1.weight", "layer1.0.downsample.0.weight", "layer2.0.conv2.weight", "layer3.5.bn3.weight",
                 "layer4.2.conv3.weight", "layer4.2.bn3.bias"],
}
STATS = {"resnet18": ["bn1", "layer2.0.downsample.1", "layer4.1.bn2"], "resnet50": ["bn1", "layer4.2.bn3"]}


def main():
    from mano_train.networks.bases import resnet as ref_resnet

    torch.manual_seed(0)
    torch.set_num_threads(4)
    out = {}
    for name in ("resnet18", "resnet50"):
        rng = np.random.RandomState(77)
        x = torch.from_numpy(rng.uniform(-0.5, 0.5, size=(3, 3, 64, 64)).astype(np.float32))
        cot = torch.from_numpy(rng.normal(size=(3, 512 if name == "resnet18" else 2048)).astype(np.float32))
        out[name + "_x"] = x.numpy()
        out[name + "_cot"] = cot.numpy()
        for mode in ("train", "eval"):
            net = load_seeded(getattr(ref_resnet, name)(pretrained=False), SEED[name])
            net.train(mode == "train")
            xin = x.clone().requires_grad_()
            feats, extra = net(xin)
            assert extra == {}
            (feats * cot).sum().backward()
            tag = "%s_%s_" % (name, mode)
            out[tag + "features"] = feats.detach().numpy()
            out[tag + "gx"] = xin.grad.numpy()
            params = dict(net.named_parameters())
            for p in PROBES[name]:
                out[tag + "g:" + p] = subsample(params[p].grad.numpy())
            assert params["fc.weight"].grad is None  # the classifier head is never reached
            sd = net.state_dict()
            for s in STATS[name]:
                out[tag + "rm:" + s] = sd[s + ".running_mean"].numpy()
                out[tag + "rv:" + s] = sd[s + ".running_var"].numpy()
                out[tag + "nbt:" + s] = sd[s + ".num_batches_tracked"].numpy()
        sd = getattr(ref_resnet, name)(pretrained=False).state_dict()
        out[name + "_layout"] = np.array(["%s %s" % (k, "x".join(str(d) for d in v.shape)) for k, v in sd.items()])
    path = os.path.join(HERE, "resnet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
