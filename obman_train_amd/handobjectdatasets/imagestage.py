"""Deferred image work of one sample (``ImagePlan``) and its batched execution on the GPU (``DeviceImageStage``).

The reference renders each training image on a DataLoader worker with PIL (``handataset.py:373-405``: blur + colour
jitter of the whole source image, affine crop, tensorise, normalise - milliseconds of CPU per sample).  Here the worker
only draws the augmentation parameters; the pixels of the whole batch are produced by three HIP kernels
(``csrc/imgstream.hip``) directly into the tensor the encoder reads, bit-identical to the PIL path.
"""
import ctypes

import numpy as np
import torch

from .. import _lib, ops
from . import imgtrans


class ImagePlan:
    """What ``get_sample`` would have done to one image: source pixels + flip + crop + blur + colour ops."""

    __slots__ = ("image", "flip", "affine_fixed", "blur", "ops")

    def __init__(self, image, flip, affine_fixed, blur=(-1, 0, 0), ops=()):
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("source image must be uint8 [H,W,3] RGB, got %s %s" % (image.dtype, image.shape))
        self.image = image
        self.flip = bool(flip)
        self.affine_fixed = [int(a) for a in affine_fixed]
        if any(not (-(1 << 31) <= a < (1 << 31)) for a in self.affine_fixed):
            raise ValueError("fixed-point affine coefficient overflows int32")
        self.blur = tuple(int(b) for b in blur)
        self.ops = [(int(op), float(f)) for op, f in ops]
        if len(self.ops) > 4:
            raise ValueError("at most four colour ops per sample")

    def params(self):
        """-> one ``obman_img_params`` record (include/obman_hip.h)."""
        p = _lib.ImgParams()
        p.src_h, p.src_w, p.flip = self.image.shape[0], self.image.shape[1], int(self.flip)
        for k in range(6):
            p.A[k] = self.affine_fixed[k]
        p.blur_r, p.blur_ww, p.blur_fw = self.blur
        p.n_ops = len(self.ops)
        for k, (op, f) in enumerate(self.ops):
            p.op[k] = op
            p.factor[k] = f  # Python float -> C float, as PIL's blend parses its alpha
            if op == imgtrans.OP_HUE:
                p.hue_shift = imgtrans.hue_shift(f)
        return p


class DeviceImageStage:
    """Batch of ``ImagePlan`` -> ``[B,3,inp_res,inp_res]`` fp32 on the ROCm device.

    ``black_padding`` / ``mean`` / ``std`` as in ``HandDataset`` (``handataset.py:389-404``); ``channels_last`` writes the
    NHWC memory format the encoder's convolutions run in (same logical NCHW shape, saves the layout-conversion pass)."""

    def __init__(self, inp_res=256, black_padding=False, mean=(0.5, 0.5, 0.5), std=(1.0, 1.0, 1.0), channels_last=False,
                 device="cuda"):
        self.inp_res, self.channels_last = int(inp_res), bool(channels_last)
        self.black_pad = int(self.inp_res * 0.2) if black_padding else 0
        self.mean, self.std = tuple(mean), tuple(std)
        self.device = torch.device(device)
        ops.require_rocm(self.device)
        self._staging = None  # pinned host buffer, grown on demand
        self._uploaded = None  # event after the last upload: the buffer is not rewritten before it has been read

    def pack(self, plans):
        """Host side: one pinned uint8 [B,Hp,Wp,3] RGB buffer (a memcpy per image) + the [B,24] int32 parameter block."""
        B = len(plans)
        Hp = max(p.image.shape[0] for p in plans)
        Wp = max(p.image.shape[1] for p in plans)
        need = B * Hp * Wp * 3
        if self._uploaded is not None:
            self._uploaded.synchronize()
        if self._staging is None or self._staging.numel() < need:
            self._staging = torch.empty(need, dtype=torch.uint8).pin_memory()
        host = self._staging[:need].view(B, Hp, Wp, 3)
        view = host.numpy()
        words = np.zeros((B, 24), np.int32)
        for b, plan in enumerate(plans):
            h, w = plan.image.shape[:2]
            view[b, :h, :w] = plan.image
            rec = plan.params()
            words[b] = np.frombuffer(ctypes.string_at(ctypes.addressof(rec), ctypes.sizeof(rec)), dtype=np.int32)
        max_blur = max(p.blur[0] for p in plans)
        any_contrast = any(op == imgtrans.OP_CONTRAST for p in plans for op, _ in p.ops)
        return host, torch.from_numpy(words), max_blur, any_contrast

    def __call__(self, plans):
        host, words, max_blur, any_contrast = self.pack(plans)
        src = host.to(self.device, non_blocking=True)
        par = words.to(self.device, non_blocking=True)
        self._uploaded = torch.cuda.Event()
        self._uploaded.record()
        return ops.image_stream(src, par, max_blur, any_contrast, self.inp_res, channels_last=self.channels_last,
                                black_pad=self.black_pad, mean=self.mean, std=self.std)
