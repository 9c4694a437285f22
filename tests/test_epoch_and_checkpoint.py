"""CPU: the 'next' row of SURVEY §8f - epoch loop, meters and reference-compatible checkpoint I/O - driven through the
test-only oracle backend (tests/fake_ops.py)."""
import os

import numpy as np
import pytest
import torch

from tests import fake_ops
from tests.handnet_common import build_fixture_model, fixture_sample


def test_epoch_pass_trains_and_averages(golden, monkeypatch):
    fake_ops.install(monkeypatch)
    from obman_train_amd.netscripts.epochpass3d import epoch_pass
    from obman_train_amd.trainer import make_optimizer

    g = golden("handnet_eval")
    model, _ = build_fixture_model(g, monkeypatch, train_mode=True)
    opt = make_optimizer(model, "adam", lr=1e-3)
    loader = [fixture_sample(g) for _ in range(3)]
    before = model.mano_branch.pose_reg.weight.detach().clone()
    meters, pck = epoch_pass(loader, model, epoch=0, optimizer=opt, train=True, freeze_batchnorm=True)
    assert set(pck) == {"auc", "thres", "pck_curve", "epe_mean", "epe_median", "evaluator"}
    assert pck["pck_curve"].shape == (20,) and 0.0 <= pck["auc"] <= 1.0 and pck["epe_mean"] > 0
    assert not model.training  # freeze_batchnorm => eval-mode BN during training (epochpass3d.py:48-52)
    assert not torch.equal(before, model.mano_branch.pose_reg.weight)
    am = meters.average_meters
    assert am["total_loss"].count == 3 and am["contact_auc"].count == 3
    first = float(g["total"][0])
    assert am["total_loss"].sum / 3 == pytest.approx(am["total_loss"].avg)
    assert abs(am["mano_verts3d"].avg) > 0 and np.isfinite(am["total_loss"].avg)
    # first step sees the fixture's weights: the running sum starts from the reference's loss value
    meters2, _ = epoch_pass(loader[:1], build_fixture_model(g, monkeypatch, train_mode=False)[0], epoch=0, train=False)
    assert meters2.average_meters["total_loss"].avg == pytest.approx(first, rel=1e-4)


def test_log_freq_window_keeps_every_step_in_the_running_means(golden, monkeypatch):
    """log_freq > 1 reads the losses back once per window but every step must still enter the averages (the reference adds
    each step, epochpass3d.py:111-121): same sums and counts as log_freq = 1, incl. the trailing partial window."""
    fake_ops.install(monkeypatch)
    from obman_train_amd.netscripts.epochpass3d import epoch_pass
    from obman_train_amd.trainer import make_optimizer

    g = golden("handnet_eval")
    out = {}
    for log_freq in (1, 3):
        model, _ = build_fixture_model(g, monkeypatch, train_mode=True)
        opt = make_optimizer(model, "adam", lr=1e-3)
        meters, _ = epoch_pass([fixture_sample(g) for _ in range(5)], model, epoch=0, optimizer=opt, train=True,
                               freeze_batchnorm=True, log_freq=log_freq)
        out[log_freq] = {k: (m.sum, m.count) for k, m in meters.average_meters.items()}
    assert out[1]["total_loss"][1] == 5 and set(out[1]) == set(out[3])
    for k in out[1]:
        assert out[3][k][1] == out[1][k][1], k
        assert out[3][k][0] == pytest.approx(out[1][k][0], rel=1e-6), k
    assert len({round(v, 3) for v in [out[1]["total_loss"][0]]}) == 1


def test_strict_loading_is_honoured(golden, monkeypatch, tmp_path):
    """strict=True raises on a missing model key (as the reference does); only manopth's buffers are filtered out."""
    fake_ops.install(monkeypatch)
    from obman_train_amd.modelutils import modelio

    g = golden("handnet_eval")
    model, _ = build_fixture_model(g, monkeypatch, train_mode=False)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["mano_branch.mano_layer_left.th_weights"] = torch.zeros(778, 16)  # reference-only buffer: ignored, not an error
    torch.save({"epoch": 1, "state_dict": sd, "best_score": 0.0}, str(tmp_path / "full.pth.tar"))
    with pytest.warns(UserWarning):
        modelio.load_checkpoint(model, str(tmp_path / "full.pth.tar"), strict=True)
    del sd["mano_branch.pose_reg.bias"]
    torch.save({"epoch": 1, "state_dict": sd, "best_score": 0.0}, str(tmp_path / "partial.pth.tar"))
    with pytest.warns(UserWarning), pytest.raises(RuntimeError):
        modelio.load_checkpoint(model, str(tmp_path / "partial.pth.tar"), strict=True)
    with pytest.warns(UserWarning):
        modelio.load_checkpoint(model, str(tmp_path / "partial.pth.tar"), strict=False)  # what traineval.py passes


def test_checkpoint_roundtrip_and_reference_layout(golden, monkeypatch, tmp_path):
    fake_ops.install(monkeypatch)
    from obman_train_amd.modelutils import modelio
    from obman_train_amd.trainer import make_optimizer

    g = golden("handnet_eval")
    model, _ = build_fixture_model(g, monkeypatch, train_mode=False)
    opt = make_optimizer(model, "sgd", lr=0.1)
    # a reference-style checkpoint: DataParallel prefix + manopth buffers the HIP implementation does not hold
    sd = {"module." + k: v.clone() for k, v in model.state_dict().items()}
    sd["module.mano_branch.mano_layer_right.th_betas"] = torch.zeros(1, 10)
    state = {"epoch": 7, "network": "handnet", "state_dict": sd, "best_score": 0.25, "optimizer": opt.state_dict()}
    modelio.save_checkpoint(state, is_best=True, checkpoint=str(tmp_path), snapshot=7)
    for name in ("checkpoint.pth.tar", "checkpoint_7.pth.tar", "model_best.pth.tar"):
        assert os.path.exists(os.path.join(str(tmp_path), name))
    fresh, _ = build_fixture_model(g, monkeypatch, train_mode=False)
    with torch.no_grad():
        fresh.mano_branch.pose_reg.weight.zero_()
    with pytest.warns(UserWarning):
        epoch, best = modelio.load_checkpoint(fresh, os.path.join(str(tmp_path), "checkpoint.pth.tar"), optimizer=opt)
    assert (epoch, best) == (7, 0.25)
    assert torch.equal(fresh.mano_branch.pose_reg.weight, model.mano_branch.pose_reg.weight)
    total_a, _, _ = model.forward(fixture_sample(g))
    total_b, _, _ = fresh.forward(fixture_sample(g))
    assert float(total_a) == float(total_b)
    # averaging two checkpoints (modelio.load_checkpoints)
    sd2 = {k: (v + 2.0 if v.dtype.is_floating_point else v) for k, v in sd.items()}
    torch.save(dict(state, state_dict=sd2, epoch=9), os.path.join(str(tmp_path), "b.pth.tar"))
    with pytest.warns(UserWarning):
        epoch, _ = modelio.load_checkpoints(fresh, [os.path.join(str(tmp_path), "checkpoint.pth.tar"), os.path.join(str(tmp_path), "b.pth.tar")])
    assert epoch == 9
    torch.testing.assert_close(fresh.mano_branch.pose_reg.bias, model.mano_branch.pose_reg.bias + 1.0)
    with pytest.raises(ValueError):
        modelio.load_checkpoint(fresh, os.path.join(str(tmp_path), "nope.pth.tar"))


def test_evalutil_matches_reference_golden(golden):
    from obman_train_amd.evaluation.zimeval import EvalUtil

    g = golden("zimeval")
    a, b = EvalUtil(), EvalUtil()
    for gt, pred, vis in zip(g["gt"], g["pred"], g["vis"]):
        a.feed(torch.from_numpy(gt), torch.from_numpy(pred), keypoint_vis=vis)            # the reference's per-sample API
    b.feed_batch(np.sqrt(((g["gt"] - g["pred"]) ** 2).sum(2)), g["vis"])                   # batched distances
    for ev in (a, b):
        epe_mean, epe_joint, epe_median, auc, curve, thr = ev.get_measures(0, 50, 20)
        np.testing.assert_allclose(epe_mean, g["epe_mean"], rtol=1e-6)
        np.testing.assert_allclose(np.array(epe_joint), g["epe_joint"], rtol=1e-6)
        np.testing.assert_allclose(epe_median, g["epe_median"], rtol=1e-6)
        np.testing.assert_allclose(auc, g["auc"], rtol=1e-9)
        np.testing.assert_allclose(curve, g["curve"], rtol=1e-9)
        np.testing.assert_allclose(thr, g["thresholds"])
    assert len(a.data[5]) == 0 and len(a.data[0]) == int(g["vis"][:, 0].sum())


def test_save_results_writes_reference_layout(golden, monkeypatch, tmp_path):
    fake_ops.install(monkeypatch)
    from obman_train_amd.netscripts import savemano
    from obman_train_amd.netscripts.epochpass3d import epoch_pass

    g = golden("handnet_eval")
    model, _ = build_fixture_model(g, monkeypatch, train_mode=False)
    epoch_pass([fixture_sample(g)], model, epoch=3, train=False, save_results=True, save_path=str(tmp_path))
    path = os.path.join(str(tmp_path), "save_results", "val", "epoch_3", "batch_000000.pkl")
    data = savemano.load_batch(path)
    assert set(data) == {"sample", "results"}
    # what the reference's load_batch_info reads (savemano.py:13-17)
    assert data["results"]["verts"].shape == (3, 778, 3) and isinstance(data["results"]["verts"], np.ndarray)
    assert data["results"]["objfaces"].shape == (320, 3)
    assert data["results"]["contact_info"]["repulsion_masks"].shape == (3, 778)
    assert data["sample"]["sides"] == ["left", "left", "right"] and "images" in data["sample"]


def test_graphed_step_compares_host_entries_of_any_type():
    """ADVICE r03 (trainer.py): numpy-array and list entries of a sample must compare without 'truth value is ambiguous'."""
    import numpy as np

    from obman_train_amd.trainer import _same_entry

    assert _same_entry(["left", "right"], ["left", "right"]) and not _same_entry(["left", "right"], ["left", "left"])
    assert _same_entry(np.arange(4), np.arange(4)) and not _same_entry(np.arange(4), np.arange(4) + 1)
    assert _same_entry(np.arange(4), [0, 1, 2, 3]) and not _same_entry(np.zeros((2, 2)), np.zeros(4))
    assert _same_entry("wrist", "wrist") and not _same_entry("wrist", "palm")
    assert _same_entry(None, None) and not _same_entry(None, "wrist") and not _same_entry(3, None)
