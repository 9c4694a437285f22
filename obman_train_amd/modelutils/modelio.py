"""Checkpoint I/O compatible with the reference's files.  Mirror of ``mano_train/modelutils/modelio.py:10-104``:
``checkpoint.pth.tar`` = ``{"epoch", "network", "state_dict", "best_score", "optimizer"}``, snapshots every
``snapshot`` epochs, ``model_best.pth.tar``; ``load_checkpoint`` accepts state dicts with or without the
``module.`` prefix of ``nn.DataParallel`` (adapting to what the model in hand expects), the ``load_atlas`` remap
(``base_net`` -> ``atlas_base_net``), and ``load_checkpoints`` averages several checkpoints.

``strict`` is honoured exactly as ``nn.Module.load_state_dict`` defines it (missing or unexpected keys raise when it is
True, the reference's behaviour), with one exception: the ``mano_branch.mano_layer_{right,left}.*`` buffers of manopth
that reference checkpoints carry are dropped before loading - the MANO model lives in a device blob here, not in module
buffers.  Compatibility is therefore one-directional under ``strict=True``: reference checkpoints load here; checkpoints
written here lack those buffers, so the reference loads them only with ``strict=False`` (what ``traineval.py:139-141``
passes anyway)."""
import os
import shutil
import traceback
import warnings

import torch


def _adapt_prefix(state_dict, model):
    wants = next(iter(model.state_dict().keys()), "").startswith("module.")
    has = next(iter(state_dict.keys()), "").startswith("module.")
    if wants == has:
        return dict(state_dict)
    if wants:
        return {"module." + k: v for k, v in state_dict.items()}
    return {k[len("module."):]: v for k, v in state_dict.items()}


def _load_filtered(model, state_dict, strict):
    own = model.state_dict()
    missing = set(own) - set(state_dict)
    extra = set(state_dict) - set(own)
    if missing:
        warnings.warn("Missing keys ! : {}".format(sorted(missing)))
    manopth = {k for k in extra if ".mano_layer_" in k}  # manopth buffers of reference checkpoints: not parameters here
    if manopth:
        warnings.warn("Ignoring {} manopth layer buffers stored in the checkpoint".format(len(manopth)))
        state_dict = {k: v for k, v in state_dict.items() if k not in manopth}
    model.load_state_dict(state_dict, strict=strict)


def load_checkpoints(model, resume_paths, strict=True):
    dicts, epochs = [], []
    for path in resume_paths:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        dicts.append(_adapt_prefix(ckpt["state_dict"], model))
        epochs.append(ckpt["epoch"])
    mean = {}
    for key, val in dicts[0].items():
        if val.dtype.is_floating_point:
            mean[key] = torch.stack([d[key] for d in dicts]).mean(0)
        else:
            mean[key] = dicts[-1][key]
    _load_filtered(model, mean, strict)
    return max(epochs), None


def load_checkpoint(model, resume_path, optimizer=None, strict=True, load_atlas=False):
    if not os.path.isfile(resume_path):
        raise ValueError("=> no checkpoint found at '{}'".format(resume_path))
    ckpt = torch.load(resume_path, map_location="cpu", weights_only=False)
    state_dict = _adapt_prefix(ckpt["state_dict"], model)
    if load_atlas:
        state_dict = {k.replace("base_net", "atlas_base_net") if "base_net" in k and "atlas_base_net" not in k else k: v
                      for k, v in state_dict.items()}
    _load_filtered(model, state_dict, strict)
    if optimizer is not None:
        try:
            optimizer.load_state_dict(ckpt["optimizer"])
        except (ValueError, KeyError):
            traceback.print_exc()
            warnings.warn("Couldn' load optimizer from {}".format(resume_path))
    for key in ("best_auc", "best_acc", "best_score"):
        if key in ckpt:
            return ckpt["epoch"], ckpt[key]
    return ckpt["epoch"], None


def save_checkpoint(state, is_best, checkpoint="checkpoint", filename="checkpoint.pth.tar", snapshot=None):
    os.makedirs(checkpoint, exist_ok=True)
    filepath = os.path.join(checkpoint, filename)
    torch.save(state, filepath)
    if snapshot and state["epoch"] % snapshot == 0:
        shutil.copyfile(filepath, os.path.join(checkpoint, "checkpoint_{}.pth.tar".format(state["epoch"])))
    if is_best:
        shutil.copyfile(filepath, os.path.join(checkpoint, "model_best.pth.tar"))
