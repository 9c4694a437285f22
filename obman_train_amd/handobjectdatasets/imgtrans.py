"""Colour-jitter and blur *parameters* (the pixels are touched on the GPU only).

Mirrors ``handobjectdatasets/imgtrans.py`` of the reference: ``get_color_params`` (:5-28) draws the four factors with
the global ``random`` module in the order brightness, contrast, saturation, hue; ``color_jitter`` (:30-53) builds the op
list [brightness, saturation, hue, contrast] and ``random.shuffle``s it - here the shuffled list of (op code, factor)
is returned instead of applied.  Same draws, same order, so a seeded run augments exactly like the reference.
"""
import math
import random

import numpy as np

OP_BRIGHTNESS, OP_SATURATION, OP_HUE, OP_CONTRAST = 1, 2, 3, 4


def get_color_params(brightness=0, contrast=0, saturation=0, hue=0, rng=random):
    b = rng.uniform(max(0, 1 - brightness), 1 + brightness) if brightness > 0 else None
    c = rng.uniform(max(0, 1 - contrast), 1 + contrast) if contrast > 0 else None
    s = rng.uniform(max(0, 1 - saturation), 1 + saturation) if saturation > 0 else None
    h = rng.uniform(-hue, hue) if hue > 0 else None
    return b, c, s, h


def color_jitter_plan(brightness=0, contrast=0, saturation=0, hue=0, rng=random):
    """-> shuffled [(op, factor)], op in {1 brightness, 2 saturation, 3 hue, 4 contrast}."""
    b, c, s, h = get_color_params(brightness=brightness, contrast=contrast, saturation=saturation, hue=hue, rng=rng)
    ops = []
    if b is not None:
        ops.append((OP_BRIGHTNESS, b))
    if s is not None:
        ops.append((OP_SATURATION, s))
    if h is not None:
        if not (-0.5 <= h <= 0.5):  # torchvision adjust_hue's contract
            raise ValueError("hue_factor ({}) is not in [-0.5, 0.5].".format(h))
        ops.append((OP_HUE, h))
    if c is not None:
        ops.append((OP_CONTRAST, c))
    rng.shuffle(ops)
    return ops


def hue_shift(hue_factor):
    """uint8 increment of the H channel: ``np.uint8(hue_factor * 255)`` = truncation toward zero, wrap modulo 256."""
    return int(hue_factor * 255) & 0xFF


def box_blur_weights(sigma, passes=3):
    """Gaussian blur of standard deviation ``sigma`` as PIL runs it (``ImageFilter.GaussianBlur`` = ``passes`` extended
    box filters per axis): -> (integer radius r, ww, fw), the 8.24 fixed-point weights of the 2r+1 inner taps and of the
    two fractional outer taps; (-1, 0, 0) when sigma is 0 (PIL returns a copy).  PIL computes these in C ``float``:
    reproduced with float32 scalars."""
    if sigma == 0:
        return -1, 0, 0
    f = np.float32
    sigma2 = f(f(f(sigma) * f(sigma)) / f(passes))
    big_l = f(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(math.floor((float(big_l) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    radius = f(l + a)
    r = int(radius)
    ww = int(np.uint32(f(1 << 24) / f(f(radius * f(2)) + f(1))))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw
