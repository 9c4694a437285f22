#!/usr/bin/env python
"""Close the "MANO parity unpinned" gap on a machine that HAS manopth and the licence-gated MANO files.

    python tools/compare_manopth.py --mano-root misc/mano [--device cuda] [--ncomps 30]

Runs ``manopth.manolayer.ManoLayer`` (the layer the reference calls at ``mano_train/networks/branches/manobranch.py:92-105,
170-182``) and this package on the same seeded poses / shapes for both hand sides, with and without PCA, and prints the
largest vertex / joint difference in millimetres plus the gradient difference with respect to pose and shape:

* ``oracle.mano.mano_lbs`` (the CPU restatement the HIP kernel is tested against) on the pack ``load_mano_pickle`` reads
  from the same files - always;
* ``obman_train_amd.ops.mano_lbs`` (csrc/mano_lbs.hip through the C-ABI) - with ``--device cuda`` on a ROCm box.

Pass criterion (north_star): <= 1e-4 relative, i.e. <= ~0.02 mm on a hand of ~200 mm extent.  Neither manopth nor the MANO
files exist in the development container, so this script has never been run by the authors: parity stays labelled
"unpinned" in DESIGN.md until somebody runs it and records the output.
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mano-root", default="misc/mano")
    ap.add_argument("--device", default="cpu", choices=["cpu", "cuda"])
    ap.add_argument("--ncomps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--center-idx", type=int, default=0)
    args = ap.parse_args()
    try:
        from manopth.manolayer import ManoLayer
    except ImportError as exc:
        raise SystemExit("manopth is not importable here (%s): install github.com/hassony2/manopth and chumpy" % exc)
    from oracle import mano as omano
    from obman_train_amd.mano_params import load_mano_pickle

    rng = np.random.RandomState(0)
    worst = 0.0
    for side in ("right", "left"):
        fname = os.path.join(args.mano_root, "MANO_%s.pkl" % side.upper())
        for use_pca in (True, False):
            for flat in (True, False):
                ncomps = args.ncomps if use_pca else 45
                layer = ManoLayer(ncomps=ncomps, center_idx=args.center_idx, side=side, mano_root=args.mano_root,
                                  use_pca=use_pca, flat_hand_mean=flat)
                pose = torch.from_numpy(rng.normal(0, 0.4, size=(args.batch, 3 + ncomps)).astype(np.float32))
                betas = torch.from_numpy(rng.normal(0, 1.0, size=(args.batch, 10)).astype(np.float32))
                cot_v = torch.from_numpy(rng.normal(size=(args.batch, 778, 3)).astype(np.float32))
                cot_j = torch.from_numpy(rng.normal(size=(args.batch, 21, 3)).astype(np.float32))

                def run(fn, dev="cpu"):
                    p, b = pose.to(dev).requires_grad_(), betas.to(dev).requires_grad_()
                    v, j = fn(p, b)
                    ((v * cot_v.to(dev)).sum() + (j * cot_j.to(dev)).sum()).backward()
                    return v.detach().cpu(), j.detach().cpu(), p.grad.cpu(), b.grad.cpu()

                ref = run(lambda p, b: layer(p, th_betas=b))
                pack = load_mano_pickle(fname, side=side, flat_hand_mean=flat)
                pk = omano.pack_to_torch(pack)
                cands = {"oracle": run(lambda p, b: omano.mano_lbs(pk, p, b, ncomps=ncomps, center_idx=args.center_idx,
                                                                   use_pca=use_pca))}
                if args.device == "cuda":
                    from obman_train_amd import ops
                    from obman_train_amd.mano_model import ManoModelBlob

                    blob = ManoModelBlob(pack).on(torch.device("cuda", 0))
                    cands["hip"] = run(lambda p, b: ops.mano_lbs(p, b, blob, ncomps=ncomps, use_pca=use_pca,
                                                                 center_idx=args.center_idx), "cuda")
                for name, got in cands.items():
                    dv, dj = (got[0] - ref[0]).abs().max().item(), (got[1] - ref[1]).abs().max().item()
                    gp = ((got[2] - ref[2]).abs().max() / ref[2].abs().max()).item()
                    gb = ((got[3] - ref[3]).abs().max() / ref[3].abs().max()).item()
                    scale = ref[0].abs().max().item()
                    worst = max(worst, dv / scale, dj / scale)
                    print("%-5s pca=%-5s flat_mean=%-5s %-6s verts %.3e mm  joints %.3e mm  (extent %.0f mm)  dpose %.1e  dbetas %.1e"
                          % (side, use_pca, flat, name, dv, dj, scale, gp, gb))
    print("worst relative output difference: %.2e  (%s north_star's 1e-4)" % (worst, "within" if worst <= 1e-4 else "ABOVE"))


if __name__ == "__main__":
    main()
