"""K10 (csrc/imgstream.hip) through the C-ABI against the oracle and the reference's golden outputs: bit-exact (byte work)."""
import itertools
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import poses  # noqa: E402
from oracle import inputstream as ois  # noqa: E402

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "inputstream.npz"))


def _product(case):
    from obman_train_amd.handobjectdatasets.handataset import HandDataset
    from obman_train_amd.queries import BaseQueries, TransQueries

    pose_kw, ds_kw, idxs, seed = poses.CASES[case]
    pose = poses.SeededPoses(base_key=lambda n: BaseQueries[n], trans_key=lambda n: TransQueries[n],
                             point_nb=ds_kw.get("point_nb", 600), **pose_kw)
    queries = [BaseQueries.sides if n == "sides" else TransQueries[n] for n in poses.QUERIES]
    return HandDataset(pose, queries=queries, **ds_kw), idxs, seed


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("case", sorted(poses.CASES))
def test_device_batch_matches_reference_golden(case, channels_last):
    from obman_train_amd.handobjectdatasets.handataset import HandDataset
    from obman_train_amd.queries import TransQueries

    ds, idxs, seed = _product(case)
    samples = []
    for idx in idxs:
        np.random.seed(seed * 100 + idx)
        random.seed(seed * 100 + idx)
        samples.append(ds.get_sample(idx))
    batch = HandDataset.collate(samples, ds.image_stage(channels_last=channels_last))
    got = batch[TransQueries.images]
    assert got.is_cuda and got.shape == (len(idxs), 3, ds.inp_res, ds.inp_res)
    assert got.is_contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    for k, idx in enumerate(idxs):
        want = (GOLD["%s/%d/images_u8" % (case, idx)].astype(np.float32) / np.float32(255) - np.float32(0.5)).astype(np.float32)
        assert (got[k].cpu().numpy() == want).all(), (case, idx)
    assert batch[TransQueries.joints3d].shape == (len(idxs), 21, 3)


def _random_plans(rng, n, sizes, res, max_sigma):
    from obman_train_amd.handobjectdatasets import handutils, imgtrans
    from obman_train_amd.handobjectdatasets.imagestage import ImagePlan

    perms = list(itertools.permutations([1, 2, 3, 4]))
    plans = []
    for b in range(n):
        H, W = sizes[b % len(sizes)]
        img = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        if b % 7 == 3:
            img[:] = rng.randint(0, 256, size=3)  # flat image: grey / saturated corner cases of the HSV round trip
        centre = np.array([rng.randint(-W // 4, W + W // 4 + 1), rng.randint(-H // 4, H + H // 4 + 1)])
        aff, _ = handutils.get_affine_transform(centre, rng.uniform(0.3, 2.5) * max(H, W, 4), [res, res], rot=rng.uniform(-np.pi, np.pi))
        order = perms[rng.randint(len(perms))][: rng.randint(0, 5)]
        ops = []
        for op in order:
            f = [0.0, 1.0, rng.uniform(0, 1), rng.uniform(1, 2.5)][rng.randint(4)]
            if op == imgtrans.OP_HUE:
                f = [-0.5, 0.5, 0.0, rng.uniform(-0.5, 0.5)][rng.randint(4)]
            ops.append((op, float(f)))
        sigma = [0.0, rng.uniform(0, 0.5), rng.uniform(0.5, max_sigma)][rng.randint(3)]
        plans.append(ImagePlan(img, bool(rng.randint(2)), handutils.fixed_point_affine(aff, [res, res]),
                               blur=imgtrans.box_blur_weights(sigma), ops=ops))
    return plans


def _oracle(plans, res, pad=0, mean=(0.5, 0.5, 0.5), std=(1, 1, 1)):
    recs = [{"flip": p.flip, "A": p.affine_fixed, "blur": p.blur, "ops": p.ops} for p in plans]
    return ois.imgstream_fwd([p.image for p in plans], recs, res, black_pad=pad, mean=mean, std=std)


@pytest.mark.parametrize("res,pad", [(48, False), (65, True)])
def test_ragged_batch_random_params_match_oracle(res, pad):
    from obman_train_amd.handobjectdatasets import DeviceImageStage

    rng = np.random.RandomState(21 + res)
    sizes = [(37, 53), (1, 1), (64, 64), (5, 90), (100, 33), (2, 3), (70, 129)]
    plans = _random_plans(rng, 28, sizes, res, max_sigma=4.0)
    assert max(p.blur[0] for p in plans) >= 1 and any(p.blur[0] < 0 for p in plans)
    stage = DeviceImageStage(inp_res=res, black_padding=pad, mean=(0.4, 0.5, 0.6), std=(0.5, 1.0, 2.0))
    got = stage(plans).cpu().numpy()
    want = _oracle(plans, res, pad=int(res * 0.2) if pad else 0, mean=(0.4, 0.5, 0.6), std=(0.5, 1.0, 2.0))
    bad = [(b, int((got[b] != want[b]).sum())) for b in range(len(plans)) if (got[b] != want[b]).any()]
    assert not bad, bad


def test_full_size_batch_matches_oracle_and_identity_crop_returns_source_bytes():
    """BASELINE.json configs[4] input shape: 64 FHB-sized frames (480x270) -> 64 x 3 x 256 x 256, default jitter ranges."""
    from obman_train_amd.handobjectdatasets import DeviceImageStage, ImagePlan

    rng = np.random.RandomState(4)
    plans = _random_plans(rng, 64, [(270, 480)], 256, max_sigma=0.5)
    stage = DeviceImageStage(inp_res=256)
    got = stage(plans)
    again = stage(plans)
    assert torch.equal(got, again)  # integer atomics only: deterministic
    want = _oracle(plans, 256)
    assert (got.cpu().numpy() == want).all()
    # size-independent property: the identity warp without ops is the source image itself, /255 - 0.5
    src = rng.randint(0, 256, size=(256, 256, 3)).astype(np.uint8)
    ident = ImagePlan(src, False, [65536, 0, 32768, 0, 65536, 32768])
    out = stage([ident])[0].cpu().numpy()
    assert (out == (src.transpose(2, 0, 1).astype(np.float32) / np.float32(255) - np.float32(0.5))).all()
    mirrored = stage([ImagePlan(src, True, [65536, 0, 32768, 0, 65536, 32768])])[0].cpu().numpy()
    assert (mirrored == out[:, :, ::-1]).all()


def test_error_behaviour():
    from obman_train_amd import ops
    from obman_train_amd._lib import ObmanHipError
    from obman_train_amd.handobjectdatasets import DeviceImageStage, ImagePlan

    img = np.zeros((8, 8, 3), np.uint8)
    stage = DeviceImageStage(inp_res=8)
    with pytest.raises(ObmanHipError):  # LDS tile of the blur kernel is sized for r <= 8
        stage([ImagePlan(img, False, [65536, 0, 32768, 0, 65536, 32768], blur=(9, 1000, 10))])
    with pytest.raises(ObmanHipError):
        ops.image_stream(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), torch.zeros(1, 24, dtype=torch.int32).cuda(), -1, False, 8)
    with pytest.raises(ValueError):
        ops.image_stream(torch.zeros(1, 8, 8, 3, dtype=torch.uint8).cuda(), torch.zeros(2, 24, dtype=torch.int32).cuda(), -1, False, 8)


def test_train_step_consumes_the_device_stream():
    """HandDataset -> DeviceBatchLoader -> HandNet.forward/backward/Adam: the batch dict is what the model's forward
    expects (handnet.py:198-392), images arrive as a device tensor in channels_last layout."""
    from obman_train_amd.handobjectdatasets import DeviceBatchLoader, HandDataset, SyntheticPoses
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.queries import BaseQueries, TransQueries
    from obman_train_amd.synthetic import CONFIGS
    from obman_train_amd.trainer import make_optimizer, train_step

    np.random.seed(0)
    random.seed(0)
    ds = HandDataset(SyntheticPoses(n=4, src_hw=(135, 240)), inp_res=128, sides="left",
                     queries=[TransQueries.images, TransQueries.joints3d, TransQueries.verts3d, TransQueries.objpoints3d,
                              TransQueries.center3d, TransQueries.affinetrans, TransQueries.camintrs, BaseQueries.sides])
    # the prefetching loader (side stream + event hand-over) must deliver the same bytes as the synchronous one
    def images_of(prefetch):
        np.random.seed(0)
        random.seed(0)
        return [b[TransQueries.images].clone() for b in DeviceBatchLoader(ds, batch_size=2, drop_last=True, channels_last=True, prefetch=prefetch)]

    sync_imgs, pre_imgs = images_of(False), images_of(True)
    assert len(sync_imgs) == len(pre_imgs) == 2 and all(torch.equal(a, b) for a, b in zip(sync_imgs, pre_imgs))
    np.random.seed(0)
    random.seed(0)
    loader = DeviceBatchLoader(ds, batch_size=2, num_workers=0, drop_last=True, channels_last=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = HandNet(**CONFIGS["c2"]).to(dev).train()
    opt = make_optimizer(model)
    seen = 0
    for batch in loader:
        assert batch[TransQueries.images].is_cuda and batch[TransQueries.images].shape == (2, 3, 128, 128)
        batch["root"] = "wrist"
        total, results, losses = train_step(model, opt, batch)
        assert torch.isfinite(total).all()
        assert results["verts"].shape == (2, 778, 3)
        seen += 1
    assert seen == 2


def test_epoch_pass_over_the_whole_input_pipeline():
    """get_dataset -> DeviceBatchLoader (prefetch) -> ConcatDataloader -> epoch_pass: the wiring of traineval.py:200-330 on this
    package end to end, one training epoch and one validation epoch over two synthetic 'datasets'."""
    import warnings

    from obman_train_amd.datautils import ConcatDataloader
    from obman_train_amd.handobjectdatasets import DeviceBatchLoader
    from obman_train_amd.netscripts.epochpass3d import epoch_pass
    from obman_train_amd.netscripts.get_datasets import get_dataset
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.synthetic import CONFIGS
    from obman_train_amd.trainer import make_optimizer

    np.random.seed(0)
    random.seed(0)
    torch.manual_seed(0)
    loaders = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in (0, 1):
            ds = get_dataset("synthetic", "train", sides="left", meta={"size": 4, "src_hw": (135, 240), "seed": seed})
            ds.inp_res = 128
            loaders.append(DeviceBatchLoader(ds, batch_size=2, drop_last=True, channels_last=True))
    model = HandNet(**CONFIGS["c2"]).cuda()
    opt = make_optimizer(model)
    meters, _ = epoch_pass(ConcatDataloader(loaders), model, epoch=0, optimizer=opt, train=True, debug=False)
    vals = {k: m.avg for k, m in meters.average_meters.items()}
    assert vals and all(np.isfinite(v) for v in vals.values()), vals
    meters, _ = epoch_pass(ConcatDataloader(loaders), model, epoch=0, train=False, debug=False)
    assert all(np.isfinite(m.avg) for m in meters.average_meters.values())
