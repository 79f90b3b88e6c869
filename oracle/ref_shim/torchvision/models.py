"""Minimal ResNet-50 v1.5 constructor (architecture only, random init).

Restates the published torchvision 0.8.2 architecture that the reference pins
(`/root/reference/README.md:41`): Bottleneck with expansion 4, stride on the 3x3
conv, all convolutions bias-free and followed by BatchNorm, stem 7x7/2 + maxpool
3/2/1, stage depths [3, 4, 6, 3].  Only attribute names matter to the reference
(`conv1, bn1, relu, maxpool, layer1..layer4`), see modules.py:70-78.
"""
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, depths=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        stages = []
        for i, (width, n) in enumerate(zip((64, 128, 256, 512), depths)):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                ds = None
                if s != 1 or cin != width * 4:
                    ds = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=s, bias=False),
                                       nn.BatchNorm2d(width * 4))
                blocks.append(Bottleneck(cin, width, s, ds))
                cin = width * 4
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages


def resnet50(pretrained=False, **kwargs):
    return ResNet()
