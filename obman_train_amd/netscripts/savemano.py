"""Per-batch result dumps in the reference's pickle layout (``mano_train/netscripts/savemano.py:57-82``):
``{"sample": {...}, "results": {...}}`` with Enum keys replaced by their ``.value``, tensors by numpy arrays and nested
dicts converted recursively, so the reference's offline tools (``load_batch_info`` -> simulation / intersection
scripts) read files written by this implementation.  ``epoch_pass(save_results=True)`` calls ``save_batch_info``."""
import os
import pickle
from enum import Enum

import torch


def untensor(tree):
    out = {}
    for key, value in tree.items():
        name = key.value if isinstance(key, Enum) else key
        if isinstance(value, torch.Tensor):
            out[name] = value.detach().cpu().numpy()
        elif isinstance(value, dict):
            out[name] = untensor(value)
        else:
            out[name] = value
    return out


def save_batch_info(save_path, results, sample):
    folder = os.path.dirname(save_path)
    if folder:
        os.makedirs(folder, exist_ok=True)
    with open(save_path, "wb") as fh:
        pickle.dump({"sample": untensor(sample), "results": untensor(results)}, fh)


def load_batch(save_path):
    with open(save_path, "rb") as fh:
        return pickle.load(fh)
