"""ResNet-101 trunk in front of Encoder.conv1 (reference: geo-aware/models.py:24-31,42-43
`nn.Sequential(*list(torchvision.models.resnet101().children())[:-2])` + `AdaptiveAvgPool2d(14)`).

Outside the hot path (BASELINE configs start at the 14x14x2048 feature map), so this is stock torch.nn: the
convolutions run through MIOpen.  torchvision is not needed: the module tree below reproduces its layout, so the
state_dict keys under `Encoder.resnet` are torchvision's (`0.weight` = conv1, `1.*` = bn1, `4.0.conv1.weight` =
layer1 block 0, ... `7.2.bn3.*`) and pretrained weights load with `load_state_dict` / `load_torchvision_state_dict`.
There is no network here: a freshly built trunk is randomly initialised (torchvision's default init)."""
import torch
from torch import nn


class Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck: 1x1 -> 3x3 (carries the stride) -> 1x1 x4, identity / projected shortcut."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


def _stage(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, down)]
    layers += [Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def resnet101_trunk():
    """children()[:-2] of torchvision's resnet101: conv1, bn1, relu, maxpool, layer1..layer4 (3, 4, 23, 3 blocks).
    (B, 3, H, W) -> (B, 2048, H/32, W/32)."""
    trunk = nn.Sequential(
        nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
        nn.MaxPool2d(3, stride=2, padding=1),
        _stage(64, 64, 3, 1), _stage(256, 128, 4, 2), _stage(512, 256, 23, 2), _stage(1024, 512, 3, 2))
    for m in trunk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
    return trunk


_TV_CHILD = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}


def load_torchvision_state_dict(trunk, sd):
    """Load a torchvision `resnet101().state_dict()` (keys conv1.weight, layer3.7.bn2.running_mean, fc.*) into the
    trunk; the classifier (fc.*) is dropped like the reference drops the last two children."""
    mapped = {}
    for k, v in sd.items():
        head, _, rest = k.partition(".")
        if head in _TV_CHILD:
            mapped[_TV_CHILD[head] + "." + rest] = v
    return trunk.load_state_dict(mapped, strict=True)
