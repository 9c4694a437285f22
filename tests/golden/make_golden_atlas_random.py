"""DEV-CONTAINER ONLY - golden vectors of the reference's ``AtlasBranch.forward`` (random sphere samples).

    python tests/golden/make_golden_atlas_random.py

``atlasbranch.py:78-108`` draws its point sets with ``rand_grid.data.normal_(0, 1)``: to pin it, ``torch.Tensor.normal_`` is
replaced for the duration of the call by a copy from a seeded array (stored in the fixture as ``rand_grid``), so the
reference runs unmodified on known draws.  Train and eval BatchNorm, with and without the translation head; outputs, the
feature gradient, a few parameter gradients and the updated running statistics are stored.  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from tests.golden import make_golden as mg  # noqa: E402  (sets sys.path for the reference, provides the shims)
from tests.golden.common import load_seeded  # noqa: E402

SEED, POINTS, FEAT = 71, 50, 32


def main():
    mg.install_shims()
    from mano_train.networks.branches.atlasbranch import AtlasBranch

    rng = np.random.RandomState(72)
    B = 4
    feats0 = rng.normal(0, 1, size=(B, FEAT)).astype(np.float32)
    draws = rng.normal(0, 1, size=(B, 3, POINTS)).astype(np.float32)
    cot = rng.normal(0, 1, size=(B, POINTS, 3)).astype(np.float32)
    out = dict(feats=feats0, rand_grid=draws, cot=cot, seed=SEED, points_nb=POINTS)
    real_normal = torch.Tensor.normal_
    for trans in (False, True):
        for mode in ("train", "eval"):
            br = load_seeded(AtlasBranch(use_residual=False, points_nb=POINTS, bottleneck_size=FEAT, predict_trans=trans,
                                         inference_ico_divisions=1, out_factor=200), SEED)
            br.train(mode == "train")
            feats = torch.from_numpy(feats0).requires_grad_()

            def fixed_normal(self, mean=0, std=1, **kw):
                assert tuple(self.shape) == draws.shape, self.shape
                return self.copy_(torch.from_numpy(draws))

            torch.Tensor.normal_ = fixed_normal
            try:
                res = br(feats)
            finally:
                torch.Tensor.normal_ = real_normal
            (res["objpoints3d"] * torch.from_numpy(cot)).sum().backward()
            tag = "t%d_%s_" % (int(trans), mode)
            for k, v in res.items():
                out[tag + k] = v.detach().numpy()
            out[tag + "grad_feats"] = feats.grad.numpy()
            params = dict(br.named_parameters())
            for name in ("decoder.conv1.weight", "decoder.bn1.weight", "decoder.bn1.bias", "decoder.conv2.weight", "decoder.conv4.bias"):
                out[tag + "g:" + name] = params[name].grad.numpy()
            sd = br.state_dict()
            for name in ("decoder.bn1.running_mean", "decoder.bn1.running_var", "decoder.bn3.running_var"):
                out[tag + "s:" + name] = sd[name].numpy()
    path = os.path.join(HERE, "atlas_random.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
