"""Plain ``torch.nn`` definitions of the benchmark networks (LeNet-5, ImageNet ResNet-18 / ResNet-50, DenseNet-121 / 161).

torchvision is not available on the target image, and the estimators select layers by class NAME
(``Conv2d`` / ``Linear``; curvature/curvatures.py:121, :298), so these are ordinary torch modules with
the same registration order as the reference's networks (curvature/lenet5.py:11-24,
curvature/resnet.py:24-201 with the 7x7/stride-2 ImageNet stem) -- ``model.modules()`` order is the
layer-indexing contract (tests/golden/g11_layer_tables.json).
PyTorch-ROCm runs their forward/backward; nothing here is on the accelerated path.
"""
from typing import List, Sequence, Tuple

import torch
from torch import nn


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


def lenet5() -> nn.Sequential:
    """LeNet-5 variant of the reference (curvature/lenet5.py:11-24): 28x28 input, all layers biased."""
    return nn.Sequential(
        nn.Conv2d(1, 6, 5, padding=2), nn.ReLU(), nn.MaxPool2d(2, 2),
        nn.Conv2d(6, 16, 5), nn.ReLU(), nn.MaxPool2d(2, 2),
        Flatten(),
        nn.Linear(400, 120), nn.ReLU(),
        nn.Linear(120, 84), nn.ReLU(),
        nn.Linear(84, 10))


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class _Residual(nn.Module):
    """Basic (two 3x3) or bottleneck (1x1, 3x3, 1x1) residual unit; stride sits on the 3x3 of the
    bottleneck (curvature/resnet.py:76) and on the first 3x3 of the basic block (:38)."""

    def __init__(self, cin: int, width: int, stride: int, bottleneck: bool):
        super().__init__()
        cout = width * (4 if bottleneck else 1)
        if bottleneck:
            self.conv1, self.bn1 = _conv(cin, width, 1), nn.BatchNorm2d(width)
            self.conv2, self.bn2 = _conv(width, width, 3, stride), nn.BatchNorm2d(width)
            self.conv3, self.bn3 = _conv(width, cout, 1), nn.BatchNorm2d(cout)
        else:
            self.conv1, self.bn1 = _conv(cin, width, 3, stride), nn.BatchNorm2d(width)
            self.conv2, self.bn2 = _conv(width, cout, 3), nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))
        self.bottleneck = bottleneck

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        if self.bottleneck:
            out = self.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
        else:
            out = self.bn2(self.conv2(out))
        out = out + identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, depths: Sequence[int], bottleneck: bool, num_classes: int = 1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for stage, (depth, width) in enumerate(zip(depths, (64, 128, 256, 512))):
            units = []
            for u in range(depth):
                units.append(_Residual(cin, width, 2 if (u == 0 and stage > 0) else 1, bottleneck))
                cin = width * (4 if bottleneck else 1)
            setattr(self, f"layer{stage + 1}", nn.Sequential(*units))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, num_classes)
        for mod in self.modules():          # curvature/resnet.py:145-150
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def resnet18(num_classes: int = 1000) -> ResNet:
    return ResNet((2, 2, 2, 2), bottleneck=False, num_classes=num_classes)


def resnet50(num_classes: int = 1000) -> ResNet:
    return ResNet((3, 4, 6, 3), bottleneck=True, num_classes=num_classes)


class _DenseUnit(nn.Module):
    """BN-ReLU-Conv1x1 (to bn_size * growth channels) then BN-ReLU-Conv3x3 (to growth channels); the output is appended
    to the unit's input (Huang et al. 2017; the README's DenseNet-121 / 161 rows, README.rst:259-267)."""

    def __init__(self, cin: int, growth: int, bn_size: int):
        super().__init__()
        self.norm1, self.conv1 = nn.BatchNorm2d(cin), nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2, self.conv2 = nn.BatchNorm2d(bn_size * growth), nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, x):
        y = self.conv1(torch.relu(self.norm1(x)))
        y = self.conv2(torch.relu(self.norm2(y)))
        return torch.cat([x, y], 1)


class DenseNet(nn.Module):
    """ImageNet DenseNet-BC: 7x7 / stride-2 stem, dense blocks of `_DenseUnit`s joined by 1x1 transitions that halve the
    channels and the resolution.  Its layer geometries differ from a ResNet's in what matters to the factor build:
    1x1 convolutions whose input width grows by `growth` per unit (64 + 32 k, 96 + 48 k: most of them no multiple of
    128) and 3x3 convolutions with 128 / 192 input channels."""

    def __init__(self, growth: int, blocks: Sequence[int], init_features: int, bn_size: int = 4, num_classes: int = 1000):
        super().__init__()
        feats = [nn.Conv2d(3, init_features, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(init_features),
                 nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1)]
        c = init_features
        for bi, depth in enumerate(blocks):
            for _ in range(depth):
                feats.append(_DenseUnit(c, growth, bn_size))
                c += growth
            if bi + 1 < len(blocks):
                feats += [nn.BatchNorm2d(c), nn.ReLU(inplace=True), nn.Conv2d(c, c // 2, 1, bias=False), nn.AvgPool2d(2, 2)]
                c //= 2
        feats.append(nn.BatchNorm2d(c))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Linear(c, num_classes)
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight)
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)

    def forward(self, x):
        x = torch.relu(self.features(x))
        return self.classifier(torch.nn.functional.adaptive_avg_pool2d(x, (1, 1)).flatten(1))


def densenet121(num_classes: int = 1000) -> DenseNet:
    return DenseNet(32, (6, 12, 24, 16), 64, num_classes=num_classes)


def densenet161(num_classes: int = 1000) -> DenseNet:
    return DenseNet(48, (6, 12, 36, 24), 96, num_classes=num_classes)


def layer_table(model: nn.Module, input_chw: Tuple[int, int, int]) -> List[dict]:
    """[(index, name, kind, n, m, L, has_bias)] in ``model.modules()`` order for the selected layers.

    n = Cin*kh*kw (+1 with bias), m = Cout, L = output positions per sample."""
    names = {m: n for n, m in model.named_modules()}
    shapes = {}
    handles = []
    for mod in model.modules():
        if mod.__class__.__name__ in ("Conv2d", "Linear"):
            handles.append(mod.register_forward_hook(lambda m, i, o: shapes.__setitem__(m, tuple(o.shape))))
    was_training = model.training
    model.eval()
    with torch.no_grad():
        p = next(model.parameters())
        model(torch.zeros(1, *input_chw, dtype=p.dtype, device=p.device))
    model.train(was_training)
    for h in handles:
        h.remove()
    rows = []
    for mod in model.modules():
        kind = mod.__class__.__name__
        if kind not in ("Conv2d", "Linear"):
            continue
        bias = mod.bias is not None
        if kind == "Conv2d":
            n = mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1] + int(bias)
            m, L = mod.out_channels, shapes[mod][2] * shapes[mod][3]
        else:
            n, m, L = mod.in_features + int(bias), mod.out_features, 1
        rows.append({"index": len(rows), "name": names[mod], "kind": kind, "n": n, "m": m, "L": L,
                     "has_bias": bias})
    return rows
