"""Per-kernel micro-benchmark on the GPU box (HIP events on the launch stream).

    python tools/kbench.py [chamfer|contains|mano|decoder|imgstream|all]
Prints one JSON line per measurement: algorithmic bytes/flops / average time."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch


def timeit(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e-3 / iters


def kernel_us(fn, kid, iters=30, warmup=5):
    """Average device duration (us) of kernel id ``kid`` (csrc/prof.h) over ``iters`` calls of ``fn``."""
    from obman_train_amd import _lib

    for _ in range(warmup):
        fn()
    _lib.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms, n = _lib.prof_summary(kid)
    if kid in (1, 10):  # the second direction's own launch is booked under its own id (csrc/prof.h): a CALL is both
        ms += _lib.prof_summary(13 if kid == 1 else 12)[0]
    _lib.prof_enable(False)
    return ms * 1e3 / iters  # per CALL (an op may launch the kernel more than once, e.g. one launch per direction)


def bench_chamfer():
    from obman_train_amd import ops

    sizes = ((64, 642, 600), (64, 2562, 600), (64, 16050, 600), (64, 64050, 600))
    if os.environ.get("OBMAN_KBENCH_ONE"):
        sizes = ((64, 64050, 600),)
    if os.environ.get("OBMAN_KBENCH_NPRED"):  # e.g. 642: one size (PMC passes)
        sizes = ((64, int(os.environ["OBMAN_KBENCH_NPRED"]), 600),)
    rotate = int(os.environ.get("OBMAN_KBENCH_ROTATE", "0"))  # >0: cycle through that many input sets (PMC passes: keeps the MALL from serving re-used inputs)
    for B, n_p, n_g in sizes:
        p = (torch.randn(B, n_p, 3, device="cuda") * 40).requires_grad_()
        g = torch.randn(B, n_g, 3, device="cuda") * 40
        if rotate:
            sets = [(torch.randn(B, n_p, 3, device="cuda") * 40, torch.randn(B, n_g, 3, device="cuda") * 40) for _ in range(rotate)]
            state = {"i": 0}

            def fwd():
                a, b = sets[state["i"] % rotate]
                state["i"] += 1
                return ops.chamfer(a, b)
        else:
            fwd = lambda: ops.chamfer(p, g)  # noqa: E731
        t_f = kernel_us(fwd, 10, iters=max(30, 2 * rotate)) * 1e-6
        l1, l2 = ops.chamfer(p, g)
        loss = (l1 + l2).mean()
        t_b = kernel_us(lambda: torch.autograd.grad(loss, p, retain_graph=True), 11) * 1e-6
        pairs = 2.0 * B * n_p * n_g  # both directions evaluate every pair once
        print(json.dumps(dict(
            kernel="chamfer", qpt=os.environ.get("OBMAN_PM_QPT", "auto"), s5=os.environ.get("OBMAN_PM_S5", "1"), rotate=rotate, B=B, n_pred=n_p, n_gt=n_g,
            fwd_us=round(t_f * 1e6, 2), bwd_us=round(t_b * 1e6, 2),
            fwd_alg_GBps=round(20.0 * (n_p + n_g) * B / t_f / 1e9, 1), fwd_Tpairs_per_s=round(pairs / t_f / 1e12, 3),
            fwd_valu_TFLOPs=round(pairs * 5 / t_f / 1e12, 1))), flush=True)


def bench_contactmin():
    """The contact term's pair-min: 778 hand vertices against the object's points, ONLY the hand side's minima wanted
    (networks/branches/contactloss.py).  Whole call by stream events (fill + sweep + resolve / unpack)."""
    from obman_train_amd import ops

    for B, n_obj in ((64, 16050), (64, 64050)):
        hand = torch.randn(B, 778, 3, device="cuda") * 40
        obj = torch.randn(B, n_obj, 3, device="cuda") * 40
        t = timeit(lambda: ops.pairmin(hand, obj, want_y=False), iters=40, warmup=5)
        print(json.dumps(dict(kernel="contact_pairmin", fused=os.environ.get("OBMAN_PM_FUSED", "1"), B=B, n_hand=778, n_obj=n_obj,
                              call_us=round(t * 1e6, 1), Tpairs_per_s=round(B * 778.0 * n_obj / t / 1e12, 3))), flush=True)


def bench_tile_sweep():
    """BASELINE.json configs[4]: LDS reference-tile sweep of the pair-min kernel at 25 x 2562 predicted vertices.
    The tile cap is read once per process (OBMAN_PM_TILE), so this re-executes itself per value."""
    import subprocess

    # round 6: the product path at this size is the fused sweep, whose reference side (600 ground-truth points = 9.6 KB) is ONE LDS tile by
    # construction - the tile parameter only exists for the two independent sweeps (the checker path, OBMAN_PM_FUSED=0), where the
    # ground truth -> predicted direction stages 64 050 references through LDS tile by tile
    for tile in (256, 512, 1024, 2048, 3072):
        env = dict(os.environ, OBMAN_PM_TILE=str(tile), OBMAN_KBENCH_ONE="1", OBMAN_PM_FUSED="0")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "chamfer"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                print(json.dumps(dict(kernel="chamfer_tile_sweep", path="two independent sweeps (OBMAN_PM_FUSED=0)", tile_points=tile, lds_bytes=tile * 16, **{k: d[k] for k in ("n_pred", "fwd_us", "fwd_Tpairs_per_s")})), flush=True)


def bench_mano():
    from obman_train_amd import ops
    from obman_train_amd.mano_model import ManoModelBlob
    from obman_train_amd.mano_params import synthetic_mano

    blob = ManoModelBlob(synthetic_mano("right")).on("cuda")
    for B in (64, 512):
        pose = (torch.randn(B, 33, device="cuda") * 0.5).requires_grad_()
        betas = torch.randn(B, 10, device="cuda").requires_grad_()
        t_f = kernel_us(lambda: ops.mano_lbs(pose, betas, blob), 6)
        v, j = ops.mano_lbs(pose, betas, blob)
        loss = v.sum() + j.sum()
        t_b = kernel_us(lambda: torch.autograd.grad(loss, (pose, betas), retain_graph=True), 7)
        print(json.dumps(dict(kernel="mano_lbs", B=B, fwd_us=round(t_f, 2), bwd_us=round(t_b, 2))), flush=True)


def bench_contains():
    """Inside test at the configs[2] / configs[4] shapes, grid-culled product kernel vs the all-pairs checker kernel, on three
    geometries: `blob` = 25 patches of a 40 mm blob around a 35 mm point cloud (rounds 1-4 bench), `hand` = an anisotropic
    hand-sized cloud next to a 60 mm multi-patch object (what a trained model looks like), `init` = the whole object inside
    a few mm (what the random-initialised decoder of bench.py produces)."""
    import numpy as np

    from obman_train_amd import ops
    from obman_train_amd.icosphere import multi_patch

    rng = np.random.RandomState(0)
    for B, subdiv, patches in ((64, 3, 1), (64, 3, 25), (64, 4, 25)):
        v, f = multi_patch(subdiv, patches)
        faces = torch.from_numpy(f.astype(np.int32)).cuda()
        n = v.shape[0] // patches
        for geom in ("blob", "hand", "init"):
            if geom == "blob":
                verts = torch.from_numpy(v.astype(np.float32)).cuda().unsqueeze(0).repeat(B, 1, 1) * 40
                verts = verts + torch.randn(B, 1, 3, device="cuda") * 5
                pts = torch.randn(B, 778, 3, device="cuda") * 35
            else:
                scale = 60.0 if geom == "hand" else 0.5
                vv = (v[None] * scale * rng.uniform(0.7, 1.3, size=(B, 1, 3))).astype(np.float32)
                for p in range(patches):
                    vv[:, p * n:(p + 1) * n] += rng.normal(0, scale * 0.4, size=(B, 1, 3)).astype(np.float32)
                hand = rng.normal(0, 45, size=(B, 778, 3)).astype(np.float32) * np.array([1.0, 0.5, 0.25], np.float32)
                hand += rng.normal(0, 20, size=(B, 1, 3)).astype(np.float32)
                verts, pts = torch.from_numpy(vv).cuda(), torch.from_numpy(hand).cuda()
            for grouped in ((1,) if patches == 1 else (1, patches)):
                t = kernel_us(lambda: ops.mesh_contains_hits(pts, verts, faces, patches=grouped, raw_bits=True), 3)
                t_all = kernel_us(lambda: ops.mesh_contains_hits(pts, verts, faces, patches=grouped, raw_bits=True, all_pairs=True), 3)
                same = torch.equal(ops.mesh_contains_hits(pts, verts, faces, patches=grouped, raw_bits=True),
                                   ops.mesh_contains_hits(pts, verts, faces, patches=grouped, raw_bits=True, all_pairs=True))
                pairs = B * 778.0 * f.shape[0]
                print(json.dumps(dict(kernel="contains", geometry=geom, B=B, F=int(f.shape[0]), patches=grouped, us=round(t, 2),
                                      all_pairs_us=round(t_all, 2), speedup=round(t_all / t, 1), identical=bool(same),
                                      all_pairs_Gpairs_per_s=round(pairs / t_all / 1e3, 1))), flush=True)


def bench_imgstream():
    """K10: one batch of 64 source images (FHB 480x270 and ObMan 256x256) -> 64 x 3 x 256 x 256 fp32, default jitter."""
    import random
    import time

    import numpy as np

    from obman_train_amd.handobjectdatasets import handutils, imgtrans
    from obman_train_amd.handobjectdatasets.imagestage import DeviceImageStage, ImagePlan

    for (H, W) in ((270, 480), (256, 256)):
        rng = np.random.RandomState(0)
        random.seed(0)
        plans = []
        for b in range(64):
            img = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
            aff, _ = handutils.get_affine_transform(np.array([W // 2 + rng.randint(-20, 20), H // 2 + rng.randint(-20, 20)]),
                                                    rng.uniform(150, 260), [256, 256], rot=rng.uniform(-np.pi, np.pi))
            plans.append(ImagePlan(img, b % 2, handutils.fixed_point_affine(aff, [256, 256]),
                                   blur=imgtrans.box_blur_weights(random.random() * 0.5),
                                   ops=imgtrans.color_jitter_plan(brightness=0.5, saturation=0.5, hue=0.15, contrast=0.5)))
        for cl in (False, True):
            stage = DeviceImageStage(inp_res=256, channels_last=cl)
            host, words, max_blur, any_contrast = stage.pack(plans)
            src, par = host.cuda(), words.cuda()
            from obman_train_amd import ops

            t = timeit(lambda: ops.image_stream(src, par, max_blur, any_contrast, 256, channels_last=cl), iters=50, warmup=5)
            t0 = time.perf_counter()
            for _ in range(5):
                stage.pack(plans)
            t_pack = (time.perf_counter() - t0) / 5
            # algorithmic bytes: source read (4 B/px, twice when blurred or contrast-reduced: blur pass + warp gather of the
            # crop footprint is <= the image) + blurred copy write + fp32 output write
            alg = 64 * (H * W * 4 * 3 + 256 * 256 * 12)
            print(json.dumps(dict(kernel="imgstream", src_hw=[H, W], B=64, channels_last=cl, device_us=round(t * 1e6, 1),
                                  alg_GBps=round(alg / t / 1e9, 1), images_per_s_device=round(64 / t),
                                  host_pack_ms=round(t_pack * 1e3, 2))), flush=True)


def bench_bnpool():
    """Stem BatchNorm + ReLU + MaxPool(3,2,1) at the headline shape: 64 x 64 ch x 128 x 128 (268 MB activation)."""
    from obman_train_amd import ops

    bn = torch.nn.BatchNorm2d(64).cuda().train()
    pool = torch.nn.MaxPool2d(3, stride=2, padding=1)
    x = torch.randn(64, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    t_f = timeit(lambda: ops.bn_relu_maxpool(bn, x, pool, count=False), iters=20, warmup=5)
    y = ops.bn_relu_maxpool(bn, x, pool, count=False)
    g = torch.randn_like(y)
    t_b = timeit(lambda: torch.autograd.grad(y, (x, bn.weight, bn.bias), g, retain_graph=True), iters=20, warmup=5)
    xb, yb = x.numel() * 4, y.numel() * 4
    # fwd: x read twice (stats, pool) + pooled write; bwd: x read twice, pooled y + dy read twice, dx written
    print(json.dumps(dict(kernel="bn_relu_maxpool", fwd_us=round(t_f * 1e6, 1), bwd_us=round(t_b * 1e6, 1),
                          fwd_TBps=round((2 * xb + yb) / t_f / 1e12, 2), bwd_TBps=round((3 * xb + 4 * yb) / t_b / 1e12, 2))), flush=True)


def bench_decoder():
    import numpy as np

    from obman_train_amd import ops
    from obman_train_amd.icosphere import multi_patch
    from obman_train_amd.networks.branches.atlasutils import PointGenCon

    cases = [(64, 1, "f32"), (64, 1, "bf16")]
    if os.environ.get("OBMAN_KBENCH_C3"):  # configs[2]: 25 patches
        cases += [(64, 25, "f32"), (64, 25, "bf16")]
    level = 3
    if os.environ.get("OBMAN_KBENCH_DEC"):  # e.g. "bf16:25" = one case (for profiling); "bf16:25:4" = the 2562-point template of configs[4]
        mode, patches, *lv = os.environ["OBMAN_KBENCH_DEC"].split(":")
        cases = [(64, int(patches), mode)]
        level = int(lv[0]) if lv else 3
    for B, patches, mode in cases:
        dec = PointGenCon(bottleneck_size=515).cuda().train()
        dec.mfma_dtype = mode
        grid = torch.from_numpy(multi_patch(level, patches)[0].astype(np.float32)).cuda()
        feats = torch.randn(B, 512, device="cuda").requires_grad_()
        t_f = kernel_us(lambda: ops.pointgen_decode(dec, feats, grid), 8, iters=10, warmup=3)
        if os.environ.get("OBMAN_GEMM_VARIANT"):
            print(json.dumps(dict(kernel="decoder", variant=os.environ["OBMAN_GEMM_VARIANT"], fwd_us=round(t_f, 1))), flush=True)
            continue
        out = ops.pointgen_decode(dec, feats, grid)
        loss = out.square().mean()
        params = [feats] + list(dec.parameters())
        t_b = kernel_us(lambda: torch.autograd.grad(loss, params, retain_graph=True), 9, iters=10, warmup=3)
        R = B * grid.shape[0]
        flop_f = 2.0 * R * (515 * 257 + 257 * 128 + 128 * 3)
        print(json.dumps(dict(kernel="decoder", mfma=mode, B=B, N=int(grid.shape[0]), fwd_us=round(t_f, 1), bwd_us=round(t_b, 1),
                              fwd_TFLOPs=round(flop_f / t_f / 1e6, 1), bwd_TFLOPs=round(2 * flop_f / t_b / 1e6, 1))), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.zeros(1, device="cuda")
    if which in ("chamfer", "all"):
        bench_chamfer()
    if which == "tiles":
        bench_tile_sweep()
    if which == "contactmin":
        bench_contactmin()
    if which in ("mano", "all"):
        bench_mano()
    if which in ("contains", "all"):
        bench_contains()
    if which in ("bnpool", "all"):
        bench_bnpool()
    if which in ("imgstream", "all"):
        bench_imgstream()
    if which in ("decoder", "all"):
        bench_decoder()
