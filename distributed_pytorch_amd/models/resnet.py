"""ResNet-50 (bottleneck v1.5: stride on the 3x3 convolution), defined from scratch for the
"ResNet-50, synthetic 224 x 224, bf16" data-parallel configuration.  The 7x7 stem, the strided
and the narrow (64 / 128 channel) 3x3 convolutions run on the vendor library through torch (MIOpen)
in channels-last layout; the 1x1 stride-1 convolutions (a GEMM over channels-last rows), the 3x3
stride-1 convolutions of stages 3 and 4 (implicit GEMMs, ops/conv.py), batch norm (with the ReLU and
the residual add of a block folded into it), the classifier head, the loss (fused softmax
cross-entropy + accuracy), the optimizer and the gradient reducer are this framework's
kernels.  BatchNorm running statistics are module buffers, which the DDP wrapper keeps in one
flat tensor and re-broadcasts from rank 0 each step.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops.batchnorm import BatchNorm2d
from ..ops.conv import Conv3x3, PointwiseConv2d
from ..ops.linear import Linear


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None):
        super().__init__()
        self.conv1 = PointwiseConv2d(inplanes, planes)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv3x3(planes, planes, stride=stride)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = PointwiseConv2d(planes, planes * self.expansion)
        self.bn3 = BatchNorm2d(planes * self.expansion)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=identity, relu=True)  # relu(bn3(.) + identity) in one pass


class ResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), num_classes: int = 1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = Linear(512 * Bottleneck.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                nn.init.zeros_(m.bn3.weight)  # each residual branch starts as identity

    def _make_layer(self, planes: int, blocks: int, stride: int = 1) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                PointwiseConv2d(self.inplanes, planes * Bottleneck.expansion, stride=stride),
                BatchNorm2d(planes * Bottleneck.expansion),
            )
        seq = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        seq += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x), relu=True))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet50(num_classes: int = 1000) -> ResNet:
    return ResNet((3, 4, 6, 3), num_classes)
