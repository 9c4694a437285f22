"""Shared by the golden generator (dev container) and the tests (anywhere).

Nothing here touches /root/reference.  ``seeded_state`` regenerates model weights from a
numpy seed so fixtures only store inputs and expected outputs, not megabytes of weights.
"""
import os

import numpy as np
import torch
from torch import nn

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


def seeded_state(shapes, seed):
    """{name: shape} -> {name: float32/int64 tensor}; order = sorted names; numpy RandomState is portable."""
    rng = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        if leaf == "running_var":
            arr = rng.uniform(0.5, 1.5, size=shape)
        elif leaf == "running_mean":
            arr = rng.normal(0, 0.1, size=shape)
        elif leaf == "bias":
            arr = rng.normal(0, 0.05, size=shape)
        elif leaf == "weight" and len(shape) == 1:  # BatchNorm scale
            arr = 1.0 + rng.normal(0, 0.1, size=shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
            arr = rng.normal(0, 1.0 / np.sqrt(max(fan_in, 1)), size=shape)
        out[name] = torch.from_numpy(np.asarray(arr, dtype=np.float32))
    return out


def load_seeded(module, seed):
    sd = module.state_dict()
    new = seeded_state({k: v.shape for k, v in sd.items()}, seed)
    module.load_state_dict(new)
    return module


class TinyEncoder(nn.Module):
    """Stand-in for ResNet18 in the HandNet fixture: 4x4 average pool -> Linear(48, 512)."""

    def __init__(self, out_features=512):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(4)
        self.proj = nn.Linear(48, out_features)

    def forward(self, x):
        return torch.tanh(self.proj(self.pool(x).flatten(1)) * 4.0), {}


def synth_hand_object(batch, n_obj_gt, seed, hand_template_m):
    """Seeded GT hand verts [B,778,3] mm, joints [B,21,3] mm, object cloud [B,n,3] mm (ellipsoid near the palm)."""
    rng = np.random.RandomState(seed)
    tmpl = hand_template_m.astype(np.float64) * 1000.0
    tmpl = tmpl - tmpl.mean(0)
    verts = tmpl[None] + rng.normal(0, 5.0, size=(batch, 778, 3))
    joints = rng.normal(0, 40.0, size=(batch, 21, 3))
    u = rng.normal(size=(batch, n_obj_gt, 3))
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    axes = rng.uniform(20, 80, size=(batch, 1, 3))
    centre = rng.normal(0, 30.0, size=(batch, 1, 3)) + np.array([0.0, -60.0, 0.0])
    obj = u * axes + centre
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    return f32(verts), f32(joints), f32(obj)


def pack_bits(mask):
    return np.packbits(np.asarray(mask, dtype=np.uint8).reshape(-1))


def unpack_bits(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(bool)


def subsample(a, limit=8192):
    """Every k-th element of the flattened array, k = the smallest odd stride that fits ``limit`` (how large gradients
    are stored in ``resnet.npz``; the generator and the tests share this rule)."""
    flat = np.asarray(a).reshape(-1)
    k = 1
    while flat.size // k > limit:
        k += 2
    return flat[::k].copy()
