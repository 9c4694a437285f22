"""Sample-dict keys of the hot path.

The reference keys its ``sample`` dict with ``Enum`` members defined in
``handobjectdatasets/queries.py:4-47`` (reference).  Enum members of two
different classes never compare equal, so for a true drop-in the reference's
own classes are re-used whenever the caller has them importable (i.e. when
``traineval.py`` of the reference drives this package).  Otherwise the same
names/values are defined here, including the trailing space in
``TransQueries.joints2d`` (``queries.py:33``).
"""
from enum import Enum

try:  # running inside the reference's environment -> share its key objects
    from handobjectdatasets.queries import BaseQueries, TransQueries  # type: ignore
except Exception:  # stand-alone

    class BaseQueries(Enum):
        camintrs = "camintrs"
        depth = "depth"
        hand_poses = "hand_poses"
        hand_pcas = "hand_pcas"
        images = "images"
        joints2d = "joints2d"
        joints3d = "joints3d"
        meta = "meta"
        objpoints2d = "objpoints2d"
        objpoints3d = "objpoints3d"
        objverts3d = "objverts3d"
        objfaces = "objfaces"
        verts3d = "verts3d"
        sides = "sides"
        segms = "segms"
        manoidxs = "manoidxs"

    class TransQueries(Enum):
        camintrs = "camintrs"
        depth = "depth"
        images = "images"
        joints2d = "joints2d "  # sic: trailing space, as in the reference
        joints3d = "joints3d"
        objfaces = "objfaces"
        objpoints2d = "objpoints2d"
        objpoints3d = "objpoints3d"
        objverts3d = "objverts3d"
        segms = "segms"
        verts3d = "verts3d"
        center3d = "center3d"
        affinetrans = "affinetrans"
        rotmat = "rotmat"
        sdf = "sdf"
        sdf_points = "sdf_points"
        mapvals = "mapvals"
        mapidxs = "mapidxs"


__all__ = ["BaseQueries", "TransQueries"]
