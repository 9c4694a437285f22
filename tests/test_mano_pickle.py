"""MANO model files and the non-PCA pose path (MANO parity stays unpinned - manopth is not vendored).

* ``tools/make_mano_pickle.py`` writes a MANO-layout pickle (chumpy ``Ch`` shapedirs, SciPy-sparse joint regressor,
  python-2 protocol) from the synthetic pack; ``load_mano_pickle`` must read it back without chumpy, with and without
  ``flat_hand_mean``; ``get_mano_pack`` finds it under ``mano_root`` and raises when it is missing (as manopth does).
* GPU: kernel == oracle on the LOADED pack with a non-zero ``hands_mean``; the reference's ``mano_use_pca=False`` path
  (16 rotation matrices, manobranch.py:52-54,126-128) forward + backward vs the oracle; ``project_rotations``."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import mano as omano
from obman_train_amd import mano_params

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from make_mano_pickle import write_mano_pickle  # noqa: E402


def _files(tmp_path, with_mean=True):
    means = {}
    for side in ("right", "left"):
        rng = np.random.RandomState(5 if side == "right" else 6)
        means[side] = rng.normal(0, 0.15, size=45) if with_mean else None
        write_mano_pickle(str(tmp_path / ("MANO_%s.pkl" % side.upper())), mano_params.synthetic_mano(side), hands_mean=means[side])
    return means


def test_pickle_round_trip_without_chumpy(tmp_path):
    means = _files(tmp_path)
    assert "chumpy" not in sys.modules
    for side in ("right", "left"):
        want = mano_params.synthetic_mano(side)
        path = str(tmp_path / ("MANO_%s.pkl" % side.upper()))
        got = mano_params.load_mano_pickle(path, side=side)  # flat_hand_mean=True: manopth's default, what ManoBranch uses
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "hands_components"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
            assert got[k].dtype == np.float32 and got[k].flags["C_CONTIGUOUS"]
        np.testing.assert_array_equal(got["faces"], want["faces"])
        np.testing.assert_array_equal(got["hands_mean"], np.zeros(45, dtype=np.float32))
        np.testing.assert_array_equal(got["tips"], want["tips"])
        curled = mano_params.load_mano_pickle(path, side=side, flat_hand_mean=False)
        np.testing.assert_allclose(curled["hands_mean"], means[side].astype(np.float32))
    assert "chumpy" not in sys.modules  # the loader's stand-in, not an import


def test_get_mano_pack_resolution(tmp_path, monkeypatch):
    _files(tmp_path)
    pack = mano_params.get_mano_pack(str(tmp_path), "left")
    np.testing.assert_array_equal(pack["v_template"], mano_params.synthetic_mano("left")["v_template"])
    monkeypatch.delenv("OBMAN_MANO_SYNTHETIC", raising=False)
    with pytest.raises(FileNotFoundError):
        mano_params.get_mano_pack(str(tmp_path / "nowhere"), "right")  # a wrong mano_root must not train on a fake hand
    assert mano_params.get_mano_pack("synthetic", "right")["side"] == "right"  # asked for explicitly
    monkeypatch.setenv("OBMAN_MANO_SYNTHETIC", "1")
    assert mano_params.get_mano_pack(str(tmp_path / "nowhere"), "right")["side"] == "right"  # the test suite's fallback


def test_oracle_rotation_matrix_mode_equals_axis_angle_mode():
    """Feeding R = Rodrigues(axis-angle) as matrices must reproduce the 45-value axis-angle path (same formulas downstream)."""
    pk = omano.pack_to_torch(mano_params.synthetic_mano("right"), torch.float64)
    rng = np.random.RandomState(3)
    pose = torch.from_numpy(rng.normal(0, 0.5, size=(4, 48)))
    betas = torch.from_numpy(rng.normal(0, 1.0, size=(4, 10)))
    v0, j0 = omano.mano_lbs(pk, pose, betas, center_idx=9, use_pca=False)
    R = omano.axisang_to_rotmat(pose.reshape(-1, 3)).view(4, 16, 3, 3)
    v1, j1 = omano.mano_lbs(pk, R, betas, center_idx=9, use_pca=False)
    np.testing.assert_allclose(v1.numpy(), v0.numpy(), atol=1e-9)
    np.testing.assert_allclose(j1.numpy(), j0.numpy(), atol=1e-9)


def test_mano_branch_non_pca_initialisation_is_the_references():
    """pose_reg emits 144 values; zero bias, |weights| kept only on the diagonal entries of every 3x3 (manobranch.py:71-81)."""
    from obman_train_amd.networks.branches.manobranch import ManoBranch

    torch.manual_seed(0)
    br = ManoBranch(ncomps=30, base_neurons=[512, 1024, 256], center_idx=0, use_pca=False, mano_root="synthetic", adapt_skeleton=False)
    w = br.pose_reg.weight.detach()
    assert tuple(w.shape) == (144, 256) and float(br.pose_reg.bias.abs().max()) == 0.0
    mask = torch.eye(3).view(9).repeat(16).bool()
    assert float(w[~mask].abs().max()) == 0.0 and bool((w[mask] >= 0).all()) and float(w[mask].max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("flat", [False, True])
def test_kernel_matches_oracle_on_a_loaded_pickle(tmp_path, flat):
    from obman_train_amd import ops
    from obman_train_amd.mano_model import ManoModelBlob

    _files(tmp_path)
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(11)
    for side in ("right", "left"):
        pack = mano_params.load_mano_pickle(str(tmp_path / ("MANO_%s.pkl" % side.upper())), side=side, flat_hand_mean=flat)
        assert (np.abs(pack["hands_mean"]).max() > 0.05) != flat
        pk = omano.pack_to_torch(pack)
        blob = ManoModelBlob(pack).on(dev)
        pose = torch.from_numpy(rng.normal(0, 0.4, size=(5, 33)).astype(np.float32))
        betas = torch.from_numpy(rng.normal(0, 1.0, size=(5, 10)).astype(np.float32))
        cv = torch.from_numpy(rng.normal(size=(5, 778, 3)).astype(np.float32))
        cj = torch.from_numpy(rng.normal(size=(5, 21, 3)).astype(np.float32))
        p_o, b_o = pose.clone().requires_grad_(), betas.clone().requires_grad_()
        wv, wj = omano.mano_lbs(pk, p_o, b_o, ncomps=30, center_idx=9)
        ((wv * cv).sum() + (wj * cj).sum()).backward()
        p_g, b_g = pose.to(dev).requires_grad_(), betas.to(dev).requires_grad_()
        gv, gj = ops.mano_lbs(p_g, b_g, blob, ncomps=30, center_idx=9)
        ((gv * cv.to(dev)).sum() + (gj * cj.to(dev)).sum()).backward()
        scale = wv.abs().max().item()
        assert (gv.detach().cpu() - wv.detach()).abs().max().item() <= 1e-4 * scale
        assert (gj.detach().cpu() - wj.detach()).abs().max().item() <= 1e-4 * scale
        for got, want in ((p_g.grad, p_o.grad), (b_g.grad, b_o.grad)):
            assert (got.cpu() - want).abs().max().item() <= 1e-3 * want.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("center_idx,root_palm,mixed", [(9, False, True), (0, True, False), (None, False, True)])
def test_rotation_matrix_pose_mode_matches_oracle(center_idx, root_palm, mixed):
    """use_pca=False as the reference uses it: [B,16,3,3] matrices straight into the layer (not orthonormal in general -
    they come out of a Linear), forward and backward incl. the gradient with respect to every matrix entry."""
    from obman_train_amd import ops
    from obman_train_amd.mano_model import ManoModelBlob

    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(21)
    B = 6
    packs = {s: mano_params.synthetic_mano(s) for s in ("right", "left")}
    blobs = {s: ManoModelBlob(packs[s]).on(dev) for s in packs}
    R = torch.from_numpy((np.eye(3)[None, None] + rng.normal(0, 0.25, size=(B, 16, 3, 3))).astype(np.float32))
    betas = torch.from_numpy(rng.normal(0, 1.0, size=(B, 10)).astype(np.float32))
    cv = torch.from_numpy(rng.normal(size=(B, 778, 3)).astype(np.float32))
    cj = torch.from_numpy(rng.normal(size=(B, 21, 3)).astype(np.float32))
    sides = [("left" if (mixed and i % 2) else "right") for i in range(B)]
    R_o, b_o = R.clone().requires_grad_(), betas.clone().requires_grad_()
    wv, wj = torch.empty(B, 778, 3), torch.empty(B, 21, 3)
    for s in ("right", "left"):
        m = torch.tensor([x == s for x in sides])
        if int(m.sum()):
            v, j = omano.mano_lbs(omano.pack_to_torch(packs[s]), R_o[m], b_o[m], center_idx=center_idx, root_palm=root_palm, use_pca=False)
            wv[m], wj[m] = v, j
    ((wv * cv).sum() + (wj * cj).sum()).backward()
    R_g, b_g = R.to(dev).requires_grad_(), betas.to(dev).requires_grad_()
    side = torch.tensor([0 if s == "right" else 1 for s in sides], dtype=torch.int32, device=dev) if mixed else None
    gv, gj = ops.mano_lbs(R_g, b_g, blobs["right"], blobs["left"] if mixed else None, side, use_pca=False, center_idx=center_idx,
                          root_palm=root_palm)
    ((gv * cv.to(dev)).sum() + (gj * cj.to(dev)).sum()).backward()
    scale = wv.abs().max().item()
    assert (gv.detach().cpu() - wv.detach()).abs().max().item() <= 1e-4 * scale
    assert (gj.detach().cpu() - wj.detach()).abs().max().item() <= 1e-4 * scale
    assert tuple(R_g.grad.shape) == (B, 16, 3, 3)
    for got, want in ((R_g.grad, R_o.grad), (b_g.grad, b_o.grad)):
        assert (got.cpu() - want).abs().max().item() <= 1e-3 * want.abs().max().item()


@pytest.mark.gpu
def test_mano_branch_without_pca_matches_oracle_branch():
    from obman_train_amd.networks.branches.manobranch import ManoBranch
    from tests.golden.common import seeded_state

    dev = torch.device("cuda", 0)
    br = ManoBranch(ncomps=30, base_neurons=[512, 1024, 256], center_idx=0, use_shape=True, use_pca=False, mano_root="synthetic",
                    adapt_skeleton=False)
    sd = seeded_state({k: v.shape for k, v in br.state_dict().items()}, 77)
    sd["pose_reg.weight"] = sd["pose_reg.weight"] * 0.3
    sd["pose_reg.bias"] = torch.eye(3).view(9).repeat(16) + sd["pose_reg.bias"]  # near-identity matrices
    br.load_state_dict(sd)
    feats = torch.randn(5, 512)
    sides = ["left", "right", "left", "left", "right"]
    params = {k: v.clone().requires_grad_() for k, v in sd.items()}
    packs = {s: omano.pack_to_torch(mano_params.synthetic_mano(s)) for s in ("right", "left")}
    want = omano.mano_branch(params, feats, sides, packs, ncomps=30, center_idx=0, use_shape=True, use_pca=False)
    (want["verts"].square().mean() + want["joints"].square().mean()).backward()
    br.to(dev)
    got = br(feats.to(dev), sides)
    (got["verts"].square().mean() + got["joints"].square().mean()).backward()
    assert tuple(got["pose"].shape) == (5, 144)
    scale = want["verts"].abs().max().item()
    assert (got["verts"].detach().cpu() - want["verts"].detach()).abs().max().item() <= 1e-4 * scale
    assert (got["joints"].detach().cpu() - want["joints"].detach()).abs().max().item() <= 1e-4 * scale
    g, w = br.pose_reg.weight.grad.cpu(), params["pose_reg.weight"].grad
    assert (g - w).abs().max().item() <= 1e-3 * w.abs().max().item()


@pytest.mark.gpu
def test_project_rotations_is_the_closest_rotation():
    from obman_train_amd import ops

    rng = np.random.RandomState(2)
    M = torch.from_numpy((np.eye(3)[None] + rng.normal(0, 0.4, size=(64, 3, 3))).astype(np.float32)).cuda()
    R = ops.project_rotations(M)
    eye = torch.eye(3, device="cuda").expand_as(R)
    assert (R @ R.transpose(1, 2) - eye).abs().max().item() <= 1e-5
    assert (torch.det(R) - 1).abs().max().item() <= 1e-5
    assert (ops.project_rotations(R) - R).abs().max().item() <= 1e-5  # rotations are fixed points
