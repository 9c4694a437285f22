"""3-D keypoint evaluation: per-joint end-point error, PCK curve and its AUC.

Same interface and results as the evaluator the reference feeds in ``epoch_pass`` (``mano_train/evaluation/zimeval.py:21-129``,
used at ``epochpass3d.py:141-151,168-209``): ``EvalUtil(num_kp)``, ``feed(gt [K,3], pred [K,3], keypoint_vis)``,
``get_measures(val_min, val_max, steps) -> (epe_mean_all, epe_mean_joint, epe_median_all, auc_all, pck_curve_all, thresholds)``.
Own implementation: distances are kept as per-joint arrays and every statistic is vectorised; ``feed_batch`` takes a whole
``[B,K]`` distance matrix computed on the device, so an epoch needs one device->host copy instead of one per sample."""
import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class EvalUtil:
    def __init__(self, num_kp=21):
        self.num_kp = num_kp
        self._chunks = [[] for _ in range(num_kp)]

    @property
    def data(self):  # reference attribute: list (per keypoint) of recorded distances
        return [list(np.concatenate(c)) if c else [] for c in self._chunks]

    def feed(self, keypoint_gt, keypoint_pred, keypoint_vis=None):
        gt, pred = np.squeeze(_np(keypoint_gt)), np.squeeze(_np(keypoint_pred))
        assert gt.ndim == 2 and pred.ndim == 2
        vis = np.ones(gt.shape[0], dtype=bool) if keypoint_vis is None else np.squeeze(_np(keypoint_vis)).astype(bool)
        assert vis.ndim == 1
        self.feed_batch(np.sqrt(np.square(gt - pred).sum(1))[None], vis[None])

    def feed_batch(self, dists, vis=None):
        """dists [B,K] Euclidean errors; vis [B,K] bool or None (all visible)."""
        dists = _np(dists).astype(np.float64).reshape(-1, self.num_kp) if _np(dists).shape[-1] == self.num_kp else _np(dists)
        vis = np.ones(dists.shape, dtype=bool) if vis is None else _np(vis).astype(bool)
        for k in range(min(self.num_kp, dists.shape[1])):
            sel = dists[vis[:, k], k]
            if sel.size:
                self._chunks[k].append(sel)

    def get_measures(self, val_min, val_max, steps):
        thresholds = np.linspace(val_min, val_max, steps)
        trapz = getattr(np, "trapezoid", None) or np.trapz
        norm = trapz(np.ones_like(thresholds), thresholds)
        means, medians, aucs, curves = [], [], [], []
        for chunks in self._chunks:
            if not chunks:
                continue  # no valid measurement for this keypoint
            d = np.concatenate(chunks)
            means.append(d.mean())
            medians.append(np.median(d))
            curve = (d[None, :] <= thresholds[:, None]).mean(1)
            curves.append(curve)
            aucs.append(trapz(curve, thresholds) / norm)
        return (np.mean(np.array(means)), means, np.mean(np.array(medians)), np.mean(np.array(aucs)),
                np.mean(np.array(curves), 0), thresholds)
