"""Input stream of the hot path: the reference's ``handobjectdatasets`` sample contract with the pixel pipeline on the GPU.

``HandDataset`` (``handataset.py``) keeps the reference's constructor, queries and RNG draw order; the CPU image work
(blur, colour jitter, affine crop, tensorise, normalise) becomes an ``ImagePlan`` that ``DeviceImageStage``
(``imagestage.py``) executes for a whole batch through ``obman_imgstream_fwd`` (``csrc/imgstream.hip``).
"""
from .handataset import HandDataset  # noqa: F401
from .imagestage import DeviceImageStage, ImagePlan  # noqa: F401
from .loader import DeviceBatchLoader  # noqa: F401
from .syntheticposes import SyntheticPoses  # noqa: F401
