#!/usr/bin/env python
"""DEV-CONTAINER ONLY (needs /root/reference): SURVEY §8d's cross-check of the CPU baseline.

    python tools/cpu_crosscheck.py [--batch 4] [--steps 6] [--contact]

`bench.py`'s `cpu_baseline` times `oracle/` - this repository's restatement of the reference's formulation - because the
reference itself cannot travel to the GPU box.  This script times, on the SAME machine and threads, (a) the reference's own
`HandNet` (imported from /root/reference under the import shims of `tests/golden/make_golden.py`: identity `.cuda()`, the MANO
layer backed by the oracle restatement, everything else the reference's code, real ResNet-18) and (b) the oracle step, for a
full train step (forward + backward + Adam) of the configs[0] model, and prints the ratio.  SURVEY asks for <= ~10 %.
The result is recorded in DESIGN.md §2."""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--contact", action="store_true", help="configs[2] losses on the single sphere (c3p1) instead of configs[0]")
    args = ap.parse_args()
    from tests.golden import make_golden as mg

    mg.install_shims()
    os.environ.setdefault("OBMAN_MANO_SYNTHETIC", "1")
    from handobjectdatasets.queries import BaseQueries as RB, TransQueries as RT
    from mano_train.networks.bases import resnet as ref_resnet

    _r18 = ref_resnet.resnet18  # no network here: random init instead of the ImageNet download (same arithmetic)
    ref_resnet.resnet18 = lambda pretrained=False, **kw: _r18(pretrained=False, **kw)
    from mano_train.networks.handnet import HandNet as RefHandNet

    import bench
    from obman_train_amd.synthetic import CONFIGS, make_batch
    from obman_train_amd.queries import BaseQueries, TransQueries

    import warnings
    warnings.simplefilter("ignore")
    name = "c3p1" if args.contact else "c2"
    cfg = {k: v for k, v in CONFIGS[name].items() if k not in ("atlas_patches", "mano_root")}
    sample = make_batch(args.batch, "cpu", seed=0, image_size=args.image_size)
    ref_sample = {RT.images: sample[TransQueries.images], RT.verts3d: sample[TransQueries.verts3d],
                  RT.joints3d: sample[TransQueries.joints3d], RT.objpoints3d: sample[TransQueries.objpoints3d],
                  RB.sides: sample[BaseQueries.sides], "root": "wrist"}
    cwd = os.getcwd()
    os.chdir(mg.REF)  # assets/contact_zones.pkl is read relative to the cwd (contactloss.py:262-265)
    try:
        torch.manual_seed(0)
        model = RefHandNet(**cfg)
        model.train()
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)

        def ref_step():
            total, _, _ = model.forward(dict(ref_sample))
            opt.zero_grad()
            total.backward()
            opt.step()

        for _ in range(2):
            ref_step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ref_step()
        t_ref = (time.perf_counter() - t0) / args.steps
    finally:
        os.chdir(cwd)
    threads = torch.get_num_threads()
    # the oracle step exactly as bench.py's cpu_baseline runs it (bs-4 leg), same thread count
    port = bench.cpu_baseline(CONFIGS[name], seconds=max(4.0, 2.5 * t_ref * args.steps), image_size=args.image_size, cfg_name=name)
    leg = [l for l in port["legs"] if l["batch"] == 4][0] if args.batch == 4 else port["legs"][-1]
    t_port = leg["s_per_step"]
    print("config %s, batch %d, %dx%d, %d threads" % (name, args.batch, args.image_size, args.image_size, threads))
    print("reference HandNet (shimmed): %.3f s/step = %.2f img/s" % (t_ref, args.batch / t_ref))
    print("oracle restatement        : %.3f s/step = %.2f img/s  (batch %d leg)" % (t_port, leg["batch"] / t_port, leg["batch"]))
    print("oracle / reference step time: %.3f" % (t_port / t_ref))


if __name__ == "__main__":
    main()
