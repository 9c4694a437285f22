"""CPU: HandNet host logic (loss assembly, dict contracts, reference quirks) against the golden
vectors produced by the reference's own HandNet.forward, with the HIP ops replaced by the test-only
oracle backend (tests/fake_ops.py).  The same assertions run on the GPU through the real kernels
in tests/test_handnet_gpu.py."""
import numpy as np
import pytest
import torch

from tests import fake_ops
from tests.handnet_common import assert_matches_fixture, build_fixture_model, fixture_sample


@pytest.mark.parametrize("tag", ["train", "eval"])
def test_handnet_forward_backward_matches_reference(golden, monkeypatch, tag):
    fake_ops.install(monkeypatch)
    g = golden("handnet_" + tag)
    model, _ = build_fixture_model(g, monkeypatch, train_mode=(tag == "train"))
    total, results, losses = model.forward(fixture_sample(g))
    total.backward()
    assert_matches_fixture(g, total, results, losses, model)


def test_handnet_refuses_to_run_without_rocm(golden, monkeypatch):
    from obman_train_amd import _lib

    g = golden("handnet_eval")
    model, _ = build_fixture_model(g, monkeypatch, train_mode=False)
    with pytest.raises(_lib.ObmanHipError):
        model.forward(fixture_sample(g))


def test_handnet_contracts(golden, monkeypatch):
    fake_ops.install(monkeypatch)
    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.queries import BaseQueries, TransQueries

    g = golden("handnet_eval")
    with pytest.raises(TypeError):
        HandNet(no_such_kwarg=1)
    with pytest.raises(NotImplementedError):
        HandNet(resnet_version=34)
    model, cfg = build_fixture_model(g, monkeypatch, train_mode=False)
    # inference contract of image_demo.py:20-32: GT tensors are only presence flags when no_loss=True
    sample = {TransQueries.images: torch.from_numpy(g["images"])[:1], BaseQueries.sides: ["left"],
              TransQueries.joints3d: torch.ones(1, 21, 3), "root": "wrist", TransQueries.objpoints3d: torch.ones(1, 600, 3)}
    total, results, losses = model.forward(sample, no_loss=True, return_features=True)
    assert total is None and losses["total_loss"] is None
    for key in ("verts", "joints", "objpoints3d", "objfaces", "img_features", "contact_info"):
        assert key in results
    # no object GT -> no atlas branch, loss is the MANO loss only
    sample = fixture_sample(g)
    del sample[TransQueries.objpoints3d]
    total, results, losses = model.forward(sample)
    assert "objpoints3d" not in results and "atlas_objpoints3d" not in losses
    np.testing.assert_allclose(float(total), float(losses["mano_total_loss"]))
    # decay_regul scales the edge regulariser weight (traineval.py:403-404)
    before = model.atlas_loss.edge_regul_lambda
    model.decay_regul(0.5)
    assert model.atlas_loss.edge_regul_lambda == pytest.approx(0.5 * before)


def test_default_flags_crash_like_the_reference(monkeypatch):
    """App. C #3: atlas_lambda unset and no translation head leaves final_loss unassigned."""
    fake_ops.install(monkeypatch)
    from obman_train_amd.networks.branches.atlasbranch import AtlasLoss
    from obman_train_amd.queries import TransQueries

    loss = AtlasLoss(lambda_atlas=0, final_lambda_atlas=0.167)
    with pytest.raises(UnboundLocalError):
        loss.compute_loss({"objpoints3d": torch.zeros(1, 4, 3)}, {TransQueries.objpoints3d: torch.zeros(1, 5, 3)})


def test_weighted_terms_cpu_path_is_the_reference_composition():
    """Off the GPU ``ops.weighted_terms`` evaluates ``Tensor([0]); final += lambda * term`` (manobranch.py:252-318) and
    ``lambda_a * a + lambda_b * b + ...`` (atlasbranch.py:247-280) operation by operation: bit-identical values and gradients."""
    from obman_train_amd import ops

    torch.manual_seed(3)
    lambdas = [0.167, 1e-5, 0.5, 3.0]
    terms = [torch.rand((), requires_grad=True) for _ in lambdas]
    ref = torch.zeros(1)
    for lam, t in zip(lambdas, terms):
        ref += lam * t
    got = ops.weighted_terms(list(zip(lambdas, terms)), (1,))
    assert got.shape == (1,) and torch.equal(got, ref)
    ref_atlas = lambdas[0] * terms[0] + lambdas[1] * terms[1] + lambdas[2] * terms[2] + lambdas[3] * 0
    got_atlas = ops.weighted_terms([(lambdas[0], terms[0]), (lambdas[1], terms[1]), (lambdas[2], terms[2]), (lambdas[3], 0)], ())
    assert got_atlas.shape == () and torch.equal(got_atlas, ref_atlas)
    got += got_atlas  # handnet.py:367-383 accumulates in place into the aliased mano_total_loss
    got.backward()
    for lam, t in zip(lambdas[:3], terms[:3]):
        assert float(t.grad) == pytest.approx(2 * lam, rel=1e-6)
    with pytest.raises(TypeError):  # ``None * 0``: the reference's failure for an unset weight
        ops.weighted_terms([(0.167, terms[0]), (None, 0)], ())


def test_weighted_terms_fused_form_matches_the_sequential_one():
    """The function the ROCm path uses (stack, multiply, sum; one multiply backward) on CPU tensors: same value up to the summation
    order of a handful of fp32 terms, exact gradients ``lambda_i * g``, in-place accumulation into the result allowed."""
    from obman_train_amd import ops

    torch.manual_seed(4)
    lambdas = torch.tensor([0.167, 0.167, 1e-5, 0.5, 2.0, 7.0])
    terms = [(torch.rand(s) * 10).requires_grad_() for s in ((), (1,), (), (), (1,), ())]
    out = ops._WeightedTerms.apply(lambdas, *terms).reshape(1)
    ref = torch.zeros(1)
    for lam, t in zip(lambdas.tolist(), terms):
        ref += lam * t.detach().reshape(())
    torch.testing.assert_close(out.detach(), ref, rtol=3e-7, atol=0)
    out += 1.0
    (3.0 * out).sum().backward()
    for lam, t in zip(lambdas.tolist(), terms):
        assert t.grad.shape == t.shape
        torch.testing.assert_close(t.grad.reshape(()), torch.tensor(3.0 * lam), rtol=1e-7, atol=0)
