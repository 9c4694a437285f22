"""GPU parity: fused MANO LBS kernel (C-ABI) vs the oracle restatement (MANO parity with manopth is
unpinned - see oracle/mano.py).  Tolerances: verts/joints 1e-4 rel of the hand scale (north_star:
1e-4 rel fp32) i.e. atol 2e-3 mm on ~100 mm coordinates; gradients vs fp64 oracle autograd 1e-3 rel."""
import numpy as np
import pytest
import torch

from oracle import mano as omano
from obman_train_amd.mano_params import synthetic_mano

pytestmark = pytest.mark.gpu


def _inputs(B, ncomps, seed, scale=0.6):
    rng = np.random.RandomState(seed)
    pose = torch.from_numpy(rng.normal(0, scale, size=(B, 3 + ncomps)).astype(np.float32))
    betas = torch.from_numpy(rng.normal(0, 1.0, size=(B, 10)).astype(np.float32))
    return pose, betas


def _blobs():
    from obman_train_amd.mano_model import ManoModelBlob

    return {s: ManoModelBlob(synthetic_mano(s)) for s in ("right", "left")}


@pytest.mark.parametrize("ncomps,center_idx,root_palm,use_betas", [(30, 0, False, True), (6, 9, False, False),
                                                                   (45, 9, True, True), (30, None, False, True)])
def test_forward_matches_oracle(ncomps, center_idx, root_palm, use_betas):
    from obman_train_amd import ops

    blobs = _blobs()
    B = 5
    pose, betas = _inputs(B, ncomps, 3)
    sides = [0, 1, 0, 0, 1]
    side_t = torch.tensor(sides, dtype=torch.int32).cuda()
    v, j = ops.mano_lbs(pose.cuda(), betas.cuda() if use_betas else None, blobs["right"].on("cuda"),
                        blobs["left"].on("cuda"), side_t, ncomps=ncomps, center_idx=center_idx, root_palm=root_palm)
    for b in range(B):
        pk = omano.pack_to_torch(synthetic_mano("left" if sides[b] else "right"), torch.float64)
        vo, jo = omano.mano_lbs(pk, pose[b:b + 1].double(), betas[b:b + 1].double() if use_betas else None,
                                ncomps=ncomps, center_idx=center_idx, root_palm=root_palm)
        np.testing.assert_allclose(v[b].cpu().numpy(), vo[0].numpy(), rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(j[b].cpu().numpy(), jo[0].numpy(), rtol=1e-4, atol=2e-3)


def test_zero_pose_and_shape_gives_template():
    from obman_train_amd import ops

    blobs = _blobs()
    pose = torch.zeros(2, 33).cuda()
    v, j = ops.mano_lbs(pose, None, blobs["right"].on("cuda"), center_idx=None)
    tmpl = torch.from_numpy(synthetic_mano("right")["v_template"]) * 1000
    np.testing.assert_allclose(v[0].cpu().numpy(), tmpl.numpy(), rtol=0, atol=2e-3)


def test_root_rotation_is_rigid():
    """A pure global rotation rotates the zero-pose hand rigidly about joint 0 (centred on joint 0)."""
    from obman_train_amd import ops

    blobs = _blobs()
    aa = torch.tensor([[0.3, -0.7, 0.5]])
    pose = torch.cat([aa, torch.zeros(1, 30)], 1).cuda()
    v0, j0 = ops.mano_lbs(torch.zeros(1, 33).cuda(), None, blobs["right"].on("cuda"), center_idx=0)
    v1, j1 = ops.mano_lbs(pose, None, blobs["right"].on("cuda"), center_idx=0)
    R = omano.axisang_to_rotmat(aa.double())[0]
    np.testing.assert_allclose(v1[0].cpu().double().numpy(), (v0[0].cpu().double() @ R.T).numpy(), atol=5e-3)
    np.testing.assert_allclose(j1[0].cpu().double().numpy(), (j0[0].cpu().double() @ R.T).numpy(), atol=5e-3)


@pytest.mark.parametrize("ncomps,center_idx,root_palm", [(30, 0, False), (45, 9, True), (12, None, False)])
def test_backward_matches_fp64_oracle_autograd(ncomps, center_idx, root_palm):
    from obman_train_amd import ops

    blobs = _blobs()
    B = 4
    pose, betas = _inputs(B, ncomps, 5)
    rng = np.random.RandomState(6)
    wv = torch.from_numpy(rng.normal(size=(B, 778, 3)).astype(np.float32))
    wj = torch.from_numpy(rng.normal(size=(B, 21, 3)).astype(np.float32)) * 10
    pc, bc = pose.cuda().requires_grad_(), betas.cuda().requires_grad_()
    v, j = ops.mano_lbs(pc, bc, blobs["right"].on("cuda"), ncomps=ncomps, center_idx=center_idx, root_palm=root_palm)
    ((v * wv.cuda()).sum() + (j * wj.cuda()).sum()).backward()
    pk = omano.pack_to_torch(synthetic_mano("right"), torch.float64)
    po, bo = pose.double().requires_grad_(), betas.double().requires_grad_()
    vo, jo = omano.mano_lbs(pk, po, bo, ncomps=ncomps, center_idx=center_idx, root_palm=root_palm)
    ((vo * wv.double()).sum() + (jo * wj.double()).sum()).backward()
    for got, want in ((pc.grad, po.grad), (bc.grad, bo.grad)):
        want = want.numpy()
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-3 * np.abs(want).max(), (err, np.abs(want).max())


def test_backward_at_zero_pose_is_finite_and_correct():
    from obman_train_amd import ops

    blobs = _blobs()
    pc = torch.zeros(2, 33).cuda().requires_grad_()
    v, j = ops.mano_lbs(pc, None, blobs["right"].on("cuda"), center_idx=0)
    w = torch.linspace(-1, 1, 778 * 3).view(1, 778, 3).cuda()
    (v * w).sum().backward()
    assert torch.isfinite(pc.grad).all()
    pk = omano.pack_to_torch(synthetic_mano("right"), torch.float64)
    po = torch.zeros(2, 33, dtype=torch.float64).requires_grad_()
    vo, _ = omano.mano_lbs(pk, po, None, center_idx=0)
    (vo * w.cpu().double()).sum().backward()
    want = po.grad.numpy()
    assert np.abs(pc.grad.cpu().numpy() - want).max() <= 2e-3 * np.abs(want).max()


def test_runs_are_bitwise_deterministic():
    from obman_train_amd import ops

    blobs = _blobs()
    pose, betas = _inputs(16, 30, 8)
    outs = []
    for _ in range(2):
        pc, bc = pose.cuda().requires_grad_(), betas.cuda().requires_grad_()
        v, j = ops.mano_lbs(pc, bc, blobs["right"].on("cuda"))
        (v.sum() + (j * j).sum()).backward()
        outs.append((v.detach().clone(), pc.grad.clone(), bc.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
