"""GPU parity: fused NHWC BatchNorm(+skip)(+ReLU) kernels (C-ABI obman_bnact_fwd/bwd) vs torch.nn.BatchNorm2d + add +
relu - the op sequence of the reference's ResNet blocks (bases/resnet.py:38-52).  Same math, different summation order:
outputs rtol 1e-5/atol 1e-5, gradients 1e-4 rel of the largest entry, running statistics rtol 1e-5."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _ref(bn, x, skip, relu):
    y = bn(x)
    if skip is not None:
        y = y + skip
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (2, 128, 9, 7), (3, 512, 4, 4), (64, 64, 32, 32)])
@pytest.mark.parametrize("relu,has_skip,training", [(True, False, True), (True, True, True), (False, False, True),
                                                    (True, True, False), (False, True, True)])
def test_bn_act_matches_torch(shape, relu, has_skip, training):
    from obman_train_amd import ops

    torch.manual_seed(0)
    B, C, H, W = shape
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    bn_ref = copy.deepcopy(bn)
    x = (torch.randn(shape, device="cuda") * 2 + 0.7).contiguous(memory_format=torch.channels_last)
    skip = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last) if has_skip else None
    w = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    sa = skip.clone().requires_grad_() if has_skip else None
    sb = skip.clone().requires_grad_() if has_skip else None
    ya = ops.bn_act(bn, xa, skip=sa, relu=relu)
    yb = _ref(bn_ref, xb, sb, relu)
    assert ya.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=2e-5)
    (ya * w).sum().backward()
    (yb * w).sum().backward()

    def check(a, b, name):
        err = (a - b).abs().max().item()
        assert err <= 2e-4 * b.abs().max().item() + 1e-6, (name, err, b.abs().max().item())

    check(xa.grad, xb.grad, "dx")
    check(bn.weight.grad, bn_ref.weight.grad, "dgamma")
    check(bn.bias.grad, bn_ref.bias.grad, "dbeta")
    if has_skip:
        check(sa.grad, sb.grad, "dskip")
    torch.testing.assert_close(bn.running_mean, bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (2, 128, 9, 7), (16, 64, 32, 32)])
@pytest.mark.parametrize("relu,has_skip,use", [(True, True, "both"), (True, True, "first"), (True, True, "second"), (True, False, "both"),
                                               (False, True, "both")])
def test_bn_act_dual_output_takes_one_gradient_per_consumer(shape, relu, has_skip, use, dtype):
    """``ops.bn_act(..., dual=True)``: the output as two tensors on one storage; the gradients of their consumers reach the fused
    backward separately (obman_bnact_bwd2 adds them in its statistics pass for the residual form, relu + skip; other forms add them
    with a torch op).  Same bounds as the single-output test against BatchNorm2d + add + relu with the consumers' gradients added by
    autograd; bf16 activations: bounds of test_bn_act_bf16_activations."""
    from obman_train_amd import ops

    torch.manual_seed(1)
    B, C, H, W = shape
    bn = nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    bn_ref = copy.deepcopy(bn)

    def nhwc(t):
        return t.contiguous(memory_format=torch.channels_last)

    x = nhwc((torch.randn(shape, device="cuda") * 2 + 0.7).to(dtype))
    skip = nhwc(torch.randn(shape, device="cuda").to(dtype)) if has_skip else None
    w1, w2 = nhwc(torch.randn(shape, device="cuda")), nhwc(torch.randn(shape, device="cuda"))
    xa, xb = x.clone().requires_grad_(), x.float().clone().requires_grad_()
    sa = skip.clone().requires_grad_() if has_skip else None
    sb = skip.float().clone().requires_grad_() if has_skip else None
    y1, y2 = ops.bn_act(bn, xa, skip=sa, relu=relu, dual=True)
    assert y1.data_ptr() == y2.data_ptr() and y1 is not y2 and y2.is_contiguous(memory_format=torch.channels_last)
    yb = _ref(bn_ref, xb, sb, relu)
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)
    torch.testing.assert_close(y1.float(), yb, **tol)
    la = (y1.float() * w1).sum() * (use != "second") + (y2.float() * w2).sum() * (use != "first")
    lb = (yb * w1).sum() * (use != "second") + (yb * w2).sum() * (use != "first")
    la.backward()
    lb.backward()
    rel = 2e-4 if dtype == torch.float32 else 2e-2

    def check(a, b, name):
        err = (a.float() - b).abs().max().item()
        assert err <= rel * b.abs().max().item() + 1e-6, (name, err, b.abs().max().item())

    check(xa.grad, xb.grad, "dx")
    check(bn.weight.grad, bn_ref.weight.grad, "dgamma")
    check(bn.bias.grad, bn_ref.bias.grad, "dbeta")
    if has_skip:
        check(sa.grad, sb.grad, "dskip")
    with torch.no_grad():  # no graph: one tensor
        assert torch.is_tensor(ops.bn_act(bn, x, skip=skip, relu=relu, dual=True))


@pytest.mark.parametrize("shape,training", [((2, 64, 16, 16), True), ((3, 64, 13, 9), True), ((2, 128, 8, 8), False), ((8, 64, 64, 64), True)])
def test_bn_relu_maxpool_matches_torch(shape, training):
    from obman_train_amd import ops

    torch.manual_seed(1)
    B, C, H, W = shape
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(-1.0, 1.5)  # negative scales too: the max is over relu(s*x+t), not over x
        bn.bias.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    bn_ref = copy.deepcopy(bn)
    pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    x = (torch.randn(shape, device="cuda") * 2 + 0.3).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = ops.bn_relu_maxpool(bn, xa, pool)
    yb = pool(torch.relu(bn_ref(xb)))
    assert ya.shape == yb.shape and ya.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=2e-5)
    w = torch.randn_like(yb)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    for a, b, name in ((xa.grad, xb.grad, "dx"), (bn.weight.grad, bn_ref.weight.grad, "dgamma"), (bn.bias.grad, bn_ref.bias.grad, "dbeta")):
        err = (a - b).abs().max().item()
        assert err <= 2e-4 * b.abs().max().item() + 1e-6, (name, err, b.abs().max().item())
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=1e-5, atol=1e-6)


def test_resnet18_with_fused_bn_matches_stock_blocks():
    """Whole encoder: fused path (channels_last on the GPU) vs the same module evaluated with stock ops on the CPU."""
    from obman_train_amd.networks.bases.resnet import resnet18

    torch.manual_seed(0)
    net = resnet18().train()
    ref = copy.deepcopy(net)
    x = torch.rand(4, 3, 64, 64) - 0.5
    f_cpu, _ = ref(x)
    f_cpu.square().sum().backward()
    net.cuda()
    f_gpu, _ = net(x.cuda())
    f_gpu.square().sum().backward()
    np.testing.assert_allclose(f_gpu.detach().cpu().numpy(), f_cpu.detach().numpy(), rtol=2e-3, atol=2e-4)
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None
            continue
        err = (p1.grad.cpu() - p2.grad).abs().max().item()
        assert err <= 2e-2 * p2.grad.abs().max().item() + 1e-6, (n1, err)
    np.testing.assert_allclose(net.layer3[0].bn1.running_var.cpu().numpy(), ref.layer3[0].bn1.running_var.numpy(), rtol=1e-3)


def _bf16_close(a, b, name, tol):
    """relative L2 error of a bf16 result against the fp32 reference"""
    a, b = a.float(), b.float()
    err = (a - b).norm().item() / max(b.norm().item(), 1e-12)
    assert err <= tol, (name, err)


@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (2, 128, 9, 7), (16, 64, 32, 32)])
@pytest.mark.parametrize("relu,has_skip,training", [(True, False, True), (True, True, True), (False, False, True), (True, True, False)])
def test_bn_act_bf16_activations(shape, relu, has_skip, training):
    """bf16 flavour (autocast encoder of BASELINE configs[2]): bf16 x / skip / dy in, bf16 y / dx / dskip out, fp32 parameters and
    statistics.  Reference: the fp32 op sequence on the SAME bf16-rounded inputs; the fused kernel rounds once at each store, so
    values agree to bf16 precision (2^-8 relative per element) and the fp32 outputs (dgamma, dbeta, running stats) much closer."""
    from obman_train_amd import ops

    torch.manual_seed(2)
    B, C, H, W = shape
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    bn_ref = copy.deepcopy(bn)
    nhwc = dict(memory_format=torch.channels_last)
    x = (torch.randn(shape, device="cuda") * 2 + 0.7).bfloat16().contiguous(**nhwc)
    skip = torch.randn(shape, device="cuda").bfloat16().contiguous(**nhwc) if has_skip else None
    w = torch.randn(shape, device="cuda").bfloat16().contiguous(**nhwc)
    xa, xb = x.clone().requires_grad_(), x.float().requires_grad_()
    sa = skip.clone().requires_grad_() if has_skip else None
    sb = skip.float().requires_grad_() if has_skip else None
    ya = ops.bn_act(bn, xa, skip=sa, relu=relu)
    yb = _ref(bn_ref, xb, sb, relu)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(**nhwc)
    torch.testing.assert_close(ya.float(), yb, rtol=2 ** -7, atol=1e-2)   # one bf16 rounding of the fp32 result
    (ya.float() * w.float()).sum().backward()
    (yb * w.float()).sum().backward()
    assert xa.grad.dtype == torch.bfloat16
    _bf16_close(xa.grad, xb.grad, "dx", 1e-2)
    if has_skip:
        _bf16_close(sa.grad, sb.grad, "dskip", 1e-2)
    # the reductions are fp32 / fp64 in both: they differ only through the bf16 rounding of y's ReLU mask inputs
    _bf16_close(bn.weight.grad, bn_ref.weight.grad, "dgamma", 2e-3)
    _bf16_close(bn.bias.grad, bn_ref.bias.grad, "dbeta", 2e-3)
    torch.testing.assert_close(bn.running_mean, bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape,training", [((2, 64, 16, 16), True), ((3, 64, 13, 9), True), ((8, 64, 64, 64), True), ((2, 128, 8, 8), False)])
def test_bn_relu_maxpool_bf16_activations(shape, training):
    from obman_train_amd import ops

    torch.manual_seed(3)
    B, C, H, W = shape
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(-1.0, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    bn_ref = copy.deepcopy(bn)
    pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    x = (torch.randn(shape, device="cuda") * 2 + 0.3).bfloat16().contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(), x.float().requires_grad_()
    ya = ops.bn_relu_maxpool(bn, xa, pool)
    yb = pool(torch.relu(bn_ref(xb)))
    assert ya.dtype == torch.bfloat16 and ya.shape == yb.shape
    torch.testing.assert_close(ya.float(), yb, rtol=2 ** -7, atol=1e-2)
    w = torch.randn_like(yb).bfloat16()
    (ya.float() * w.float()).sum().backward()
    (yb * w.float()).sum().backward()
    # arg-max routing can differ where two window taps round to the same bf16 value: compare in relative L2
    _bf16_close(xa.grad, xb.grad, "dx", 5e-2)
    _bf16_close(bn.weight.grad, bn_ref.weight.grad, "dgamma", 2e-2)
    _bf16_close(bn.bias.grad, bn_ref.bias.grad, "dbeta", 2e-2)
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=1e-5, atol=1e-6)
