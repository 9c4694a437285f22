"""One pass over a loader: the caller of the hot path.  Mirror of ``mano_train/netscripts/epochpass3d.py:17-215`` for
what drives the model - mode switching incl. ``freeze_batchnorm`` (:48-52), forward -> ``zero_grad`` -> ``backward`` ->
``step`` (:80-91), per-loss running averages (:111-121), ``(avg_meters, pck_info)`` return - without the reference's
display / PCK / pickle side effects (those need matplotlib, the MANO pickles and ``progress``; SURVEY §2.1 #12,#17).
Differences by design: losses are read back once per step in a single transfer (or every ``log_freq`` steps), and an
optional ``GradientBuckets`` averages gradients across ranks before the optimizer step."""
import time

import torch

from obman_train_amd.evaluation.evalutils import AverageMeters


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
    pending = None
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
        pending = model_losses
        if (batch_idx + 1) % log_freq == 0:
            values = avg_meters.add_loss_dict(pending)
            pending = None
            if verbose:
                print("epoch {} batch {} loss {:.4f}".format(epoch, batch_idx + 1, values.get("total_loss", float("nan"))))
        time_meters.add_loss_value("batch_time", time.time() - end)
        end = time.time()
    if pending is not None:
        avg_meters.add_loss_dict(pending)
    avg_meters.time_meters = time_meters
    return avg_meters, {}
