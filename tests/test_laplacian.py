"""Laplacian regulariser: oracle vs the reference's golden (CPU), CSR builder vs the oracle's
SciPy matrix (CPU), HIP kernels vs both (GPU).  Values rtol 1e-5, gradients 1e-4 rel of the largest entry."""
import numpy as np
import pytest
import torch

from oracle import laplacian as olap
from obman_train_amd.icosphere import icosphere, multi_patch
from obman_train_amd.networks.branches.laplacianloss import LaplacianLoss, cotangent, template_laplacian_csr

T = torch.from_numpy


def test_oracle_matches_reference_golden(golden):
    g = golden("laplacian")
    L = olap.laplacian_matrix(T(g["template"]), g["faces"])
    V = T(g["V"]).requires_grad_()
    loss, Lx = olap.laplacian_loss(L, V)
    np.testing.assert_allclose(Lx.detach().numpy(), g["Lx"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(float(loss), float(g["loss"][0]), rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(V.grad.numpy(), g["grad"], rtol=1e-4, atol=1e-7)


def test_csr_builder_and_cotangent_match_oracle():
    for subdiv, patches in ((1, 1), (2, 1), (1, 3)):
        v, f = multi_patch(subdiv, patches)
        tmpl = T(v.astype(np.float32))
        rp, ci, va = template_laplacian_csr(tmpl, f)
        L = olap.laplacian_matrix(tmpl, f).toarray()
        dense = np.zeros_like(L)
        for i in range(len(rp) - 1):
            dense[i, ci[rp[i]:rp[i + 1]]] = va[rp[i]:rp[i + 1]]
        np.testing.assert_allclose(dense, L, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dense.sum(1), 0, atol=1e-4)  # rows of a Laplacian sum to zero
        assert (np.diff(rp) <= 7).all()  # icosphere degree <= 6 (+ diagonal)
        C = cotangent(tmpl.unsqueeze(0), T(f.astype(np.int64)).unsqueeze(0))[0]
        np.testing.assert_allclose(C.numpy(), olap.cotangent_weights(tmpl, f).numpy(), rtol=1e-6)


@pytest.mark.gpu
def test_hip_matches_reference_golden_and_oracle(golden):
    g = golden("laplacian")
    loss_mod = LaplacianLoss(g["faces"], T(g["template"]))
    V = T(g["V"]).cuda().requires_grad_()
    loss = loss_mod(V)
    np.testing.assert_allclose(float(loss), float(g["loss"][0]), rtol=1e-5)
    loss.backward()
    err = np.abs(V.grad.cpu().numpy() - g["grad"]).max()
    assert err <= 1e-4 * np.abs(g["grad"]).max()
    # bigger, multi-patch, weighted upstream gradient
    v, f = multi_patch(3, 2)
    tmpl = T(v.astype(np.float32))
    rng = np.random.RandomState(5)
    X = T((v[None] * 40 + rng.normal(0, 2, size=(5,) + v.shape)).astype(np.float32))
    mod = LaplacianLoss(f, tmpl)
    xg = X.cuda().requires_grad_()
    (mod(xg) * 2.5).backward()
    xo = X.double().requires_grad_()
    lo, _ = olap.laplacian_loss(olap.laplacian_matrix(tmpl, f), xo)
    (lo * 2.5).backward()
    np.testing.assert_allclose(float(mod(xg.detach())), float(lo), rtol=1e-5)
    ref = xo.grad.numpy()
    assert np.abs(xg.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.gpu
def test_atlas_loss_with_laplacian_term():
    from obman_train_amd.networks.branches.atlasbranch import AtlasLoss
    from obman_train_amd.queries import TransQueries

    v, f = icosphere(2)
    tmpl = T(v.astype(np.float32))
    loss = AtlasLoss(lambda_atlas=0.167, final_lambda_atlas=None, lambda_laplacian=0.1, laplacian_faces=f, laplacian_verts=tmpl)
    pts = (tmpl * 30).unsqueeze(0).repeat(2, 1, 1).cuda().requires_grad_()
    gt = torch.randn(2, 50, 3).cuda() * 30
    total, parts = loss.compute_loss({"objpoints3d": pts, "objfaces": f}, {TransQueries.objpoints3d: gt})
    assert "atlas_laplac" in parts and float(parts["atlas_laplac"]) > 0
    np.testing.assert_allclose(float(total), 0.167 * float(parts["atlas_objpoints3d"]) + 0.1 * float(parts["atlas_laplac"]), rtol=1e-5)
    total.backward()
    assert torch.isfinite(pts.grad).all()
