"""Mirror of ``mano_train/networks/netutils.py:4-19``: BatchNorm freezing helpers used by
``traineval.py:91-101`` (momentum 0 = running statistics stop moving)."""
import torch


def _zero_bn_momentum(model):
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 0


def rec_freeze(model):
    _zero_bn_momentum(model)
    for p in model.parameters():
        p.requires_grad = False


def freeze_batchnorm_stats(model):
    _zero_bn_momentum(model)
