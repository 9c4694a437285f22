"""Dataset factory of the training script with the GPU input stream behind it.

Mirror of ``get_dataset`` (reference ``mano_train/netscripts/get_datasets.py:11-139``): same arguments, same query
intersection, same jitter ranges, same ``limit_size`` subsetting - but the returned ``HandDataset`` is this package's
(``obman_train_amd.handobjectdatasets``), whose image work runs on the GPU.  The on-disk readers (``ObMan``, ``FHBHands``,
``Core50``, ``YanaDemo``, ``StereoHands``: directory layouts, annotation caches) are not re-implemented: they are taken from the
reference's own ``handobjectdatasets`` package when it is importable, exactly as ``traineval.py`` uses them, and plugged into
the new ``HandDataset`` through their accessor protocol.  ``dat_name="synthetic"`` (not in the reference) selects the seeded
stand-in used by the benchmarks.
"""
import warnings

import numpy as np
from torch.utils.data import Subset

from ..handobjectdatasets import HandDataset, SyntheticPoses
from ..queries import BaseQueries, TransQueries

DEFAULT_QUERIES = (TransQueries.affinetrans, TransQueries.images, TransQueries.verts3d, TransQueries.center3d, TransQueries.joints3d,
                   TransQueries.objpoints3d, TransQueries.camintrs, BaseQueries.sides)


def _reference_readers():
    try:
        from handobjectdatasets import core50, fhbhands, obman, stereohands, yanademo  # the reference's readers
    except Exception as exc:  # not on PYTHONPATH, or one of their optional dependencies is missing
        raise ImportError("the dataset readers live in the reference's `handobjectdatasets` package; put the obman_train "
                          "checkout on PYTHONPATH (or use dat_name='synthetic')") from exc
    return core50, fhbhands, obman, stereohands, yanademo


def _pose_dataset(dat_name, split, mini_factor, meta, use_cache):
    if dat_name == "synthetic":
        return SyntheticPoses(n=int(meta.get("size", 256)), src_hw=tuple(meta.get("src_hw", (270, 480))), seed=int(meta.get("seed", 0)),
                              split=split)
    known = dat_name in ("obman", "core50", "yanademo", "stereohands") or "fhbhands" in dat_name
    if not known:
        raise ValueError("Unrecognized dataset name {}".format(dat_name))
    suffix = dat_name.split("_")[-1]
    if "fhbhands" in dat_name and suffix not in ("obj", "hand"):
        raise ValueError("suffix in {} after _ should be in [obj|hand], got {}".format(dat_name, suffix))
    core50, fhbhands, obman, stereohands, yanademo = _reference_readers()
    if dat_name == "obman":
        return obman.ObMan(mini_factor=mini_factor, mode=meta["mode"], override_scale=meta["override_scale"], segment=False, split=split,
                           use_cache=use_cache, use_external_points=True)
    if dat_name == "core50":
        meta.setdefault("class_name", "can")
        return core50.Core50(use_cache=False, mini_factor=mini_factor, class_name=meta["class_name"])
    if dat_name == "yanademo":
        return yanademo.YanaDemo(version=meta["version"], side=meta["side"])
    if dat_name == "stereohands":
        return stereohands.StereoHands(split=split, use_cache=use_cache, gt_detections=True)
    kwargs = dict(mini_factor=mini_factor, split=split, use_cache=use_cache, use_objects=suffix == "obj",
                  split_type=meta["fhbhands_split_type"], test_object=meta["fhbhands_split_choice"])
    if suffix == "obj":
        kwargs["topology"] = meta["fhbhands_topology"]
    return fhbhands.FHBHands(**kwargs)


def get_dataset(dat_name, split, train_it=True, mini_factor=None, black_padding=False, center_idx=9, point_nb=600, sides="both",
                meta=None, max_queries=DEFAULT_QUERIES, use_cache=True, limit_size=None):
    meta = {} if meta is None else meta
    pose_dataset = _pose_dataset(dat_name, split, mini_factor, meta, use_cache)
    # maximal set of queries the reader supports (reference: set intersection; kept in the caller's order here)
    supported = set(pose_dataset.all_queries)
    queries = [q for q in max_queries if q in supported]
    scale_jittering = 0.2 if dat_name == "stereohands" else 0.3
    meta.setdefault("override_scale", False)
    dataset = HandDataset(pose_dataset, black_padding=black_padding, block_rot=False, sides=sides, train=train_it, max_rot=np.pi,
                          normalize_img=False, center_idx=center_idx, point_nb=point_nb, scale_jittering=scale_jittering,
                          center_jittering=0.2, queries=queries, as_obj_only=meta["override_scale"])
    if limit_size is not None:
        if len(dataset) < limit_size:
            warnings.warn("limit size {} < dataset size {}, working with full dataset".format(limit_size, len(dataset)))
        else:
            warnings.warn("Working wth subset of {} of size {}".format(dat_name, limit_size))
            dataset = Subset(dataset, list(range(limit_size)))
    return dataset
