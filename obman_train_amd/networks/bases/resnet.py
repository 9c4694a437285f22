"""Image encoder: ResNet-18/50 feature extractor (stock PyTorch-ROCm / MIOpen convolutions).

Same contract as ``mano_train/networks/bases/resnet.py:99-248`` (reference):
``forward(x) -> (features [B, 512|2048], {})`` where the features are the global
mean over the last stage's spatial map (``x.mean(3).mean(2)``, ``resnet.py:179-185``;
resolution-agnostic, the 7x7 avgpool is unused) and the state-dict key names are
torchvision's (``conv1, bn1, layer{1..4}.{i}.conv{1,2,3}/bn{1,2,3}/downsample.{0,1}, fc``)
so reference / ImageNet checkpoints load.  Per north_star the conv backbone is
not a HIP-kernel target; it runs in channels_last so MIOpen picks its NHWC kernels.

No network in this environment: ``pretrained=True`` loads
``$OBMAN_RESNET_WEIGHTS/resnet{18,50}.pth`` if present and otherwise keeps the
seeded default init (a warning is printed once).
"""
import os
import warnings

import torch
from torch import nn

from obman_train_amd import ops

__all__ = ["ResNet", "resnet18", "resnet50"]


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    expansion = 1
    _count = True  # False while ResNet.forward bumps every num_batches_tracked in one foreach launch
    _dual = False  # True while ResNet.forward runs a block that is followed by another one: output as a pair, see ops.bn_act

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(cin, planes, 3, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        x, xs = x if isinstance(x, tuple) else (x, x)  # the previous block's output, once per consumer (ops.bn_act, dual)
        if self.downsample is None:
            skip = xs
        else:
            skip = ops.bn_act(self.downsample[1], ops.shadow_conv2d(self.downsample[0], xs), relu=False, count=self._count)
        y = ops.bn_act(self.bn1, ops.shadow_conv2d(self.conv1, x), count=self._count)
        return ops.bn_act(self.bn2, ops.shadow_conv2d(self.conv2, y), skip=skip, count=self._count, dual=self._dual)


class Bottleneck(nn.Module):
    expansion = 4
    _count = True
    _dual = False

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(cin, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        x, xs = x if isinstance(x, tuple) else (x, x)
        if self.downsample is None:
            skip = xs
        else:
            skip = ops.bn_act(self.downsample[1], ops.shadow_conv2d(self.downsample[0], xs), relu=False, count=self._count)
        y = ops.bn_act(self.bn1, ops.shadow_conv2d(self.conv1, x), count=self._count)
        y = ops.bn_act(self.bn2, ops.shadow_conv2d(self.conv2, y), count=self._count)
        return ops.bn_act(self.bn3, ops.shadow_conv2d(self.conv3, y), skip=skip, count=self._count, dual=self._dual)


class ResNet(nn.Module):
    def __init__(self, block, depths, num_classes=1000, features=True):
        super().__init__()
        self.features = features
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        width = 64
        stages = []
        for s, (planes, n) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for i in range(n):
                stride = 2 if (i == 0 and s > 0) else 1
                down = None
                if stride != 1 or width != planes * block.expansion:
                    down = nn.Sequential(
                        nn.Conv2d(width, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                        nn.BatchNorm2d(planes * block.expansion),
                    )
                blocks.append(block(width, planes, stride, down))
                width = planes * block.expansion
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        # never reached with features=True (resnet.py:184-186) but part of the checkpoint layout
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if x.is_cuda:
            if not getattr(self, "_nhwc_weights", False):
                # keep the filters in NHWC too, once: otherwise every conv call re-lays them out (52 copies / step)
                self.to(memory_format=torch.channels_last)
                self._nhwc_weights = True
            x = x.contiguous(memory_format=torch.channels_last)
        dtype = getattr(self, "autocast_dtype", None)
        if dtype is not None and x.is_cuda and not torch.is_autocast_enabled():
            # opt-in reduced-precision encoder (BASELINE configs[2] "bf16"): convolutions on the bf16 MFMA path, features
            # handed to the fp32 heads; the fused BN kernels take the bf16 activations (fp32 statistics and parameters).
            with torch.autocast("cuda", dtype=dtype):
                feats, extra = self.forward(x)
            return feats.float(), extra
        batched = x.is_cuda and self.training
        dual = x.is_cuda and torch.is_grad_enabled()  # block outputs as (conv consumer, skip consumer) pairs: ops.bn_act
        if not hasattr(self, "_blocks"):
            self._blocks = [m for m in self.modules() if isinstance(m, (BasicBlock, Bottleneck))]
        if batched:  # one multi-tensor add instead of one tiny kernel per BatchNorm layer
            if not hasattr(self, "_nbt") or self._nbt[0] is not self.bn1.num_batches_tracked:  # rebuilt after .to()/.cuda()
                self._nbt = [m.num_batches_tracked for m in self.modules()
                             if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
            with torch.no_grad():
                torch._foreach_add_(self._nbt, 1)
        for blk in self._blocks:
            blk._count = not batched
            blk._dual = dual and blk is not self._blocks[-1]
        try:
            x = ops.bn_relu_maxpool(self.bn1, ops.shadow_conv2d(self.conv1, x), self.maxpool, count=not batched)
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        finally:
            for blk in self._blocks:
                blk._count, blk._dual = True, False
        if x.is_cuda and x.is_contiguous(memory_format=torch.channels_last):
            # global average pool: the NHWC tensor as [B, H W, C], ONE coalesced reduction, and a backward that writes the
            # NHWC gradient directly (the two-step form of resnet.py:179 below costs two strided reductions, two divisions
            # and a layout copy of the expanded gradient: 45 us per step against 12; same value up to fp32 summation order)
            B, C, H, W = x.shape
            x = x.permute(0, 2, 3, 1).reshape(B, H * W, C).mean(1)
        else:
            x = x.mean(3).mean(2)
        x = x.view(x.size(0), -1)
        if self.features:
            return x, {}
        return self.fc(x)


def _maybe_pretrained(model, name, pretrained):
    if not pretrained:
        return model
    root = os.environ.get("OBMAN_RESNET_WEIGHTS", "")
    path = os.path.join(root, name + ".pth")
    if root and os.path.exists(path):
        model.load_state_dict(torch.load(path, map_location="cpu"))
    else:
        warnings.warn(
            "%s: no ImageNet weights available offline (set OBMAN_RESNET_WEIGHTS); using seeded random init" % name
        )
    return model


def resnet18(pretrained=False, **kw):
    return _maybe_pretrained(ResNet(BasicBlock, (2, 2, 2, 2), **kw), "resnet18", pretrained)


def resnet50(pretrained=False, **kw):
    return _maybe_pretrained(ResNet(Bottleneck, (3, 4, 6, 3), **kw), "resnet50", pretrained)
