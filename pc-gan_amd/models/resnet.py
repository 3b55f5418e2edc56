"""ResNet trunk of the Elo rating encoder on the HIP layer set.

Mirrors the module/attribute names of the reference's models/resnet.py (conv1, bn1, layerN.M.
conv1/drop1/bn1/conv2/drop2/bn2/downsample.0/.1) so that state_dict keys match
(SURVEY.md appendix B).  One reference quirk is kept on purpose: the optional Dropout2d sits
BETWEEN the convolution and its BatchNorm (reference models/resnet.py:58-65).
"""
import torch.nn as tnn

from ..hip import nn as hnn
from ..hip.lib import ACT_NONE, ACT_RELU


def _drop_layer(p):
    return hnn.Dropout2d(p) if p > 0 else hnn.IdentityMapping()


def _conv3x3(cin, cout, stride=1):
    return hnn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _conv1x1(cin, cout, stride=1):
    return hnn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class BasicBlock(tnn.Module):
    """reference models/resnet.py:31-73"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dropout=0.):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.drop1 = _drop_layer(dropout)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.relu = tnn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.drop2 = _drop_layer(dropout)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        if self.downsample is not None:
            identity = self.downsample[1](self.downsample[0](x))
        out = self.bn1(self.drop1(self.conv1(x)), ACT_RELU)
        # relu(bn2(.) + identity): residual add and activation fused into the normalise pass
        return self.bn2(self.drop2(self.conv2(out)), ACT_RELU, 0.0, identity)


class Bottleneck(tnn.Module):
    """reference models/resnet.py:76-122"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dropout=0.):
        super().__init__()
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)
        self.drop2 = _drop_layer(dropout)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * self.expansion)
        self.drop3 = _drop_layer(dropout)
        self.bn3 = hnn.BatchNorm2d(planes * self.expansion)
        self.relu = tnn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        if self.downsample is not None:
            identity = self.downsample[1](self.downsample[0](x))
        out = self.bn1(self.conv1(x), ACT_RELU)
        out = self.bn2(self.drop2(self.conv2(out)), ACT_RELU)
        return self.bn3(self.drop3(self.conv3(out)), ACT_RELU, 0.0, identity)


class ResNet(tnn.Module):
    """Trunk only is used on the hot path (ResNetFeature deletes `fc`); reference models/resnet.py:125-196."""

    def __init__(self, block, layers, num_classes=1000, dropout=0.):
        super().__init__()
        widths = [64, 128, 256, 512]
        self.inplanes = widths[0]
        self.conv1 = hnn.Conv2d(3, widths[0], kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = hnn.BatchNorm2d(widths[0])
        self.relu = tnn.ReLU(inplace=True)
        self.maxpool = hnn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, widths[0], layers[0], 1, dropout)
        self.layer2 = self._stage(block, widths[1], layers[1], 2, dropout)
        self.layer3 = self._stage(block, widths[2], layers[2], 2, dropout)
        self.layer4 = self._stage(block, widths[3], layers[3], 2, dropout)
        self.avgpool = tnn.AdaptiveAvgPool2d((1, 1))
        self.fc = tnn.Linear(widths[3] * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, tnn.Conv2d):
                tnn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, tnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)

    def _stage(self, block, planes, n, stride, dropout):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = tnn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                  hnn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down, dropout=dropout)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes, dropout=dropout) for _ in range(1, n)]
        return tnn.Sequential(*blocks)

    def features(self, x):
        x = self.bn1(self.conv1(x), ACT_RELU)
        x = self.maxpool(x)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                x = blk(x)
        return x

    def forward(self, x):
        raise NotImplementedError('pcgan_amd: the classifier head (avgpool+fc) is outside the hot path; '
                                  'use ResNetFeature')


def resnet18(pretrained=False, **kw):
    assert not pretrained, 'no network access: load weights through load_pretrained()'
    return ResNet(BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(pretrained=False, **kw):
    assert not pretrained, 'no network access: load weights through load_pretrained()'
    return ResNet(BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(pretrained=False, **kw):
    assert not pretrained, 'no network access: load weights through load_pretrained()'
    return ResNet(Bottleneck, [3, 4, 6, 3], **kw)
