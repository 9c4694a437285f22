"""One pass over a loader: the caller of the hot path.  Mirror of ``mano_train/netscripts/epochpass3d.py:17-215`` for
what drives the model - mode switching incl. ``freeze_batchnorm`` (:48-52), forward -> ``zero_grad`` -> ``backward`` ->
``step`` (:80-91), per-loss running averages (:111-121), joint errors fed to the PCK evaluator (:141-151) and the
``(avg_meters, pck_info)`` return with ``pck_info = {auc, thres, pck_curve, epe_mean, epe_median, evaluator}`` (:168-196) -
without the reference's display / image / pickle side effects (matplotlib, MANO pickles, ``progress``).
Differences by design: losses are read back once per step in a single transfer (or, with ``log_freq`` > 1, the whole
window of steps in one transfer - every step still enters the running means, as in the reference), and an optional
``GradientBuckets`` averages gradients across ranks before the optimizer step."""
import time

import os

import numpy as np
import torch

from obman_train_amd.evaluation.evalutils import AverageMeters
from obman_train_amd.evaluation.zimeval import EvalUtil
from obman_train_amd.netscripts import savemano
from obman_train_amd.queries import TransQueries


def epoch_pass(loader, model, epoch, optimizer=None, debug=True, freeze_batchnorm=False, display=False, display_freq=10,
               save_path="checkpoints/debug", idxs=None, train=True, inspect_weights=False, fig=None, save_results=False,
               buckets=None, log_freq=1, verbose=False):
    avg_meters, time_meters = AverageMeters(), AverageMeters()
    if train:
        model.eval() if freeze_batchnorm else model.train()
    else:
        model.eval()
    net = model.module if hasattr(model, "module") else model
    end = time.time()
    window = []  # loss dicts of the steps since the last read-back (device tensors)
    idxs = list(range(21)) if idxs is None else idxs
    joint_errs, joint_vis = [], []  # device tensors; copied to the host once, after the loop
    for batch_idx, sample in enumerate(loader):
        time_meters.add_loss_value("data_time", time.time() - end)
        with torch.set_grad_enabled(train):
            model_loss, results, model_losses = net.forward(sample, return_features=inspect_weights)
        if train:
            if buckets is not None and buckets.enabled:
                buckets.zero_grad()
            else:
                optimizer.zero_grad(set_to_none=True)
            model_loss.backward()
            if buckets is not None:
                buckets.finish()
            optimizer.step()
        if save_results:  # epochpass3d.py:135-139: save_results/<train|val>/epoch_N/batch_XXXXXX.pkl
            split = "train" if train else "val"
            folder = os.path.join(save_path, "save_results", split, "epoch_{}".format(epoch))
            savemano.save_batch_info(os.path.join(folder, "batch_{:06d}.pkl".format(batch_idx)), results=results, sample=sample)
        if "joints" in results and TransQueries.joints3d in sample:
            pred = results["joints"].detach()[:, idxs]
            gt = sample[TransQueries.joints3d].to(pred.device)[:, idxs]
            joint_errs.append((pred - gt).pow(2).sum(2).sqrt())
            vis = sample.get("vis")
            joint_vis.append(None if vis is None else torch.as_tensor(vis)[:, idxs].bool())
        window.append({k: (v.detach() if torch.is_tensor(v) else v) for k, v in model_losses.items()})
        if (batch_idx + 1) % log_freq == 0:
            values = avg_meters.add_loss_window(window)
            window = []
            if verbose:
                print("epoch {} batch {} loss {:.4f}".format(epoch, batch_idx + 1, values.get("total_loss", float("nan"))))
        time_meters.add_loss_value("batch_time", time.time() - end)
        end = time.time()
    if window:
        avg_meters.add_loss_window(window)
    avg_meters.time_meters = time_meters
    pck_info = {}
    if joint_errs:
        evaluator = EvalUtil(num_kp=len(idxs))
        errs = torch.cat(joint_errs).cpu().numpy()
        vis = None
        if any(v is not None for v in joint_vis):
            vis = np.concatenate([np.ones(e.shape, dtype=bool) if v is None else v.cpu().numpy() for e, v in zip(joint_errs, joint_vis)])
        evaluator.feed_batch(errs, vis)
        epe_mean, _, epe_median, auc, curve, thresholds = evaluator.get_measures(0, 50, 20)
        pck_info = {"auc": auc, "thres": thresholds, "pck_curve": curve, "epe_mean": epe_mean, "epe_median": epe_median,
                    "evaluator": evaluator}
    return avg_meters, pck_info
