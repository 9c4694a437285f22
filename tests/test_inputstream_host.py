"""Host side of the GPU input stream (obman_train_amd/handobjectdatasets): the annotation arithmetic and the drawn image
parameters of the product ``HandDataset`` against the reference's own ``get_sample`` outputs (tests/golden/inputstream.npz).
The pixels are rendered here by the ORACLE's CPU model of the kernel contract from the product's ``ImagePlan`` - that pins
the parameters without a GPU; tests/test_inputstream_gpu.py renders the same plans with the HIP kernels."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import poses  # noqa: E402
from obman_train_amd.handobjectdatasets import handutils, imgtrans  # noqa: E402
from obman_train_amd.handobjectdatasets.handataset import HandDataset  # noqa: E402
from obman_train_amd.handobjectdatasets.imagestage import ImagePlan  # noqa: E402
from obman_train_amd.queries import BaseQueries, TransQueries  # noqa: E402
from oracle import inputstream as ois  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "inputstream.npz"))


def product_dataset(case):
    pose_kw, ds_kw, idxs, seed = poses.CASES[case]
    pose = poses.SeededPoses(base_key=lambda n: BaseQueries[n], trans_key=lambda n: TransQueries[n],
                             point_nb=ds_kw.get("point_nb", 600), **pose_kw)
    queries = [BaseQueries.sides if n == "sides" else TransQueries[n] for n in poses.QUERIES]
    return HandDataset(pose, queries=queries, **ds_kw), idxs, seed


def plan_record(plan):
    return {"flip": plan.flip, "A": plan.affine_fixed, "blur": plan.blur, "ops": plan.ops}


def draw(ds, idx, seed):
    np.random.seed(seed * 100 + idx)
    random.seed(seed * 100 + idx)
    return ds.get_sample(idx)


@pytest.mark.parametrize("case", sorted(poses.CASES))
def test_product_get_sample_matches_reference_golden(case):
    ds, idxs, seed = product_dataset(case)
    for idx in idxs:
        s = draw(ds, idx, seed)
        tag = "%s/%d/" % (case, idx)
        assert (s[TransQueries.affinetrans].numpy() == GOLD[tag + "affinetrans"]).all()
        assert (s[TransQueries.joints2d].numpy() == GOLD[tag + "joints2d"]).all()
        for q, k in ((TransQueries.joints3d, "joints3d"), (TransQueries.verts3d, "verts3d"), (TransQueries.objpoints3d, "objpoints3d"),
                     (TransQueries.center3d, "center3d"), (TransQueries.camintrs, "camintrs")):
            np.testing.assert_array_equal(np.asarray(s[q]), GOLD[tag + k], err_msg=k)
        assert s[BaseQueries.sides] == str(GOLD[tag + "side"])
        plan = s[TransQueries.images]
        assert isinstance(plan, ImagePlan)
        pad = int(ds.inp_res * 0.2) if ds.black_padding else 0
        img = ois.imgstream_fwd([plan.image], [plan_record(plan)], ds.inp_res, black_pad=pad)[0]
        want = (GOLD[tag + "images_u8"].astype(np.float32) / np.float32(255) - np.float32(0.5)).astype(np.float32)
        assert (img == want).all(), "%d pixels differ" % int((img != want).sum())


def test_params_record_layout_matches_header():
    plan = ImagePlan(np.zeros((7, 9, 3), np.uint8), True, [1, -2, 3, 4, 5, -6], blur=(2, 111, 222),
                     ops=[(imgtrans.OP_HUE, -0.1), (imgtrans.OP_CONTRAST, 1.25)])
    rec = plan.params()
    words = np.frombuffer(bytes(rec), dtype=np.int32)
    assert words.size == 24
    assert list(words[:3]) == [7, 9, 1] and list(words[3:9]) == [1, -2, 3, 4, 5, -6]
    assert list(words[9:12]) == [2, 111, 222] and words[12] == 2 and list(words[13:15]) == [3, 4]
    assert np.frombuffer(bytes(rec), dtype=np.float32)[18] == np.float32(1.25)
    assert words[21] == (int(-0.1 * 255) & 0xFF) == 231


def test_blur_weights_and_fixed_affine_agree_with_oracle():
    rng = np.random.RandomState(3)
    for sigma in [0.0, 1e-4, 0.3, 0.49, 0.5, 1.3, 2.9] + list(rng.uniform(0, 4, 20)):
        want = (-1, 0, 0) if sigma == 0 else ois.box_weights(ois.gaussian_box_radius(sigma))
        assert imgtrans.box_blur_weights(sigma) == tuple(int(v) for v in want)
    for _ in range(20):
        aff, _post = handutils.get_affine_transform(np.array([rng.randint(50, 400), rng.randint(50, 200)]), rng.uniform(60, 300),
                                                    [256, 256], rot=rng.uniform(-3, 3))
        inv = np.linalg.inv(aff)
        coeffs = (inv[0, 0], inv[0, 1], inv[0, 2], inv[1, 0], inv[1, 1], inv[1, 2])
        assert handutils.fixed_point_affine(aff, [256, 256]) == ois.affine_fixed_coeffs(coeffs)
    huge = np.array([[1e-9, 0, 0], [0, 1e-9, 0], [0, 0, 1]], np.float32)
    with pytest.raises(ValueError):
        handutils.fixed_point_affine(huge, [256, 256])


def test_error_behaviour_follows_the_reference():
    ds, idxs, seed = product_dataset("fhb_like_train")
    with pytest.raises(NotImplementedError):
        HandDataset(ds.pose_dataset, normalize_img=True)
    with pytest.raises(NotImplementedError):
        ds.get_sample(0, query=[TransQueries.segms])
    ds.hue = 0.7  # torchvision's adjust_hue rejects |hue| > 0.5: with hue=0.7 some draw must raise
    with pytest.raises(ValueError):
        for k in range(64):
            random.seed(k)
            ds.get_sample(0, query=[TransQueries.images])
    with pytest.raises(ValueError):
        ImagePlan(np.zeros((4, 4), np.uint8), False, [0] * 6)


def test_collate_without_images_is_default_collate():
    ds, idxs, seed = product_dataset("eval_no_aug")
    q = [TransQueries.joints3d, TransQueries.verts3d, BaseQueries.meta]  # (a side flip needs the image width, as in the reference)
    batch = HandDataset.collate([ds.get_sample(i, query=q) for i in idxs], stage=None)
    assert batch[TransQueries.joints3d].shape == (2, 21, 3) and isinstance(batch[TransQueries.verts3d], torch.Tensor)
    assert batch[BaseQueries.meta] == {"objname": ["octahedron", "octahedron"]}


def test_device_stage_refuses_cpu():
    from obman_train_amd._lib import ObmanHipError
    from obman_train_amd.handobjectdatasets import DeviceImageStage

    with pytest.raises(ObmanHipError):
        DeviceImageStage(device="cpu")


class _CountingStage:
    """Stand-in for DeviceImageStage on a CPU-only box: records what the loader hands over."""

    def __init__(self, res):
        self.res, self.batches = res, []

    def __call__(self, plans):
        assert all(isinstance(p, ImagePlan) for p in plans)
        self.batches.append(len(plans))
        return torch.zeros(len(plans), 3, self.res, self.res)


@pytest.mark.parametrize("workers", [0, 2])
def test_device_batch_loader_batches_like_the_reference_dataloader(workers):
    from obman_train_amd.handobjectdatasets import DeviceBatchLoader, SyntheticPoses

    ds = HandDataset(SyntheticPoses(n=10, src_hw=(40, 60)), inp_res=32, sides="left",
                     queries=[TransQueries.images, TransQueries.joints3d, TransQueries.verts3d, TransQueries.objpoints3d,
                              TransQueries.center3d, BaseQueries.sides])
    stage = _CountingStage(32)
    loader = DeviceBatchLoader(ds, batch_size=4, shuffle=False, num_workers=workers, drop_last=True, stage=stage)
    assert len(loader) == 2
    batches = list(loader)
    assert stage.batches == [4, 4]
    b = batches[0]
    assert b[TransQueries.images].shape == (4, 3, 32, 32) and b[TransQueries.verts3d].shape == (4, 778, 3)
    assert b[TransQueries.objpoints3d].shape == (4, 600, 3) and b[TransQueries.objpoints3d].dtype == torch.float32
    assert b[BaseQueries.sides] == ["left"] * 4


def test_get_dataset_factory_mirrors_the_reference_arguments():
    """netscripts/get_datasets.get_dataset (reference get_datasets.py:11-139): query intersection, jitter ranges, limit_size
    subsetting; readers come from the reference package (absent here -> a clear ImportError), 'synthetic' is the stand-in."""
    import warnings

    from torch.utils.data import Subset

    from obman_train_amd.handobjectdatasets import DeviceBatchLoader
    from obman_train_amd.netscripts.get_datasets import get_dataset

    ds = get_dataset("synthetic", "train", sides="left", meta={"size": 12, "src_hw": (40, 60)},
                     max_queries=[TransQueries.images, TransQueries.joints3d, TransQueries.objverts3d, BaseQueries.sides])
    assert isinstance(ds, HandDataset) and len(ds) == 12
    assert ds.queries == [TransQueries.images, TransQueries.joints3d, BaseQueries.sides]  # objverts3d: not offered by the reader
    assert (ds.scale_jittering, ds.center_jittering, ds.max_rot, ds.as_obj_only) == (0.3, 0.2, np.pi, False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sub = get_dataset("synthetic", "train", meta={"size": 12, "src_hw": (40, 60)}, limit_size=5)
    assert isinstance(sub, Subset) and len(sub) == 5
    stage = _CountingStage(256)
    assert len(list(DeviceBatchLoader(sub, batch_size=2, stage=stage))) == 3 and stage.batches == [2, 2, 1]
    with pytest.raises(ValueError):
        get_dataset("imagenet", "train")
    with pytest.raises(ValueError):
        get_dataset("fhbhands_feet", "train", meta={"fhbhands_split_type": "", "fhbhands_split_choice": ""})
    with pytest.raises(ImportError):
        get_dataset("obman", "train", meta={"mode": "all", "override_scale": False})


def test_concat_dataloader_round_robin_and_tags():
    """datautils.ConcatDataloader (reference datautils.py:5-39): alternates loaders, stops with the shortest, tags batches."""
    import warnings

    from obman_train_amd.datautils import ConcatDataloader
    from obman_train_amd.handobjectdatasets import DeviceBatchLoader
    from obman_train_amd.netscripts.get_datasets import get_dataset

    q = [TransQueries.images, TransQueries.joints3d, BaseQueries.sides]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = get_dataset("synthetic", "train", meta={"size": 6, "src_hw": (40, 60)}, max_queries=q, sides="left")
        b = get_dataset("synthetic", "val", meta={"size": 9, "src_hw": (40, 60), "seed": 1}, max_queries=q, sides="left", limit_size=8)
    loaders = [DeviceBatchLoader(a, batch_size=2, stage=_CountingStage(256)), DeviceBatchLoader(b, batch_size=2, stage=_CountingStage(256))]
    cat = ConcatDataloader(loaders)
    assert len(cat) == 2 * 3
    batches = list(cat)
    assert [x["split"] for x in batches] == ["train", "val"] * 3          # a has 3 batches, b has 4: stops with the shorter
    assert all(x["dataset"] == "synthetic" and x["root"] == "wrist" and x["use_stereohands"] is False for x in batches)
    assert batches[0][TransQueries.joints3d].shape == (2, 21, 3)
