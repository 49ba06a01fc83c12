"""Import shim for running the reference (`/root/reference`, Python/PyTorch) on CPU in the
BUILD CONTAINER ONLY (it does not exist on the GPU box; nothing under tests/ that runs there
imports this file).  See SURVEY.md Appendix B.

 * never write __pycache__ into the read-only reference tree
 * cv2 / imageio / skimage are imported by every reference file but unused on the tensor path
 * torchvision is absent: `models.resnet.resnet18(weights=...)` and `transforms.GaussianBlur` are
   written out HERE from torchvision 0.14.1's published definitions (environment.yml:358) --
   `BasicBlock` / `_make_layer` of models/resnet.py, `_get_gaussian_kernel1d/2d` + reflect pad +
   depthwise conv2d of transforms/functional_tensor.py -- and deliberately import nothing from
   `oracle/`: the fixtures pin the oracle, so the stand-in must not be the thing it pins
   (weights always come from the checkpoint; only the architecture / arithmetic is restated)
 * the reference calls `.cuda()` unconditionally in a few places -> identity
"""
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference/Full_model_inference/Codes'


def install():
    sys.dont_write_bytecode = True
    for name in ('cv2', 'imageio', 'skimage'):
        sys.modules.setdefault(name, types.ModuleType(name))

    import torch.nn.functional as F

    # ---- torchvision/models/resnet.py (0.14.1): conv3x3 / conv1x1 / BasicBlock / ResNet._make_layer, resnet18 = [2, 2, 2, 2]
    def conv3x3(in_planes, out_planes, stride=1):
        return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)

    def conv1x1(in_planes, out_planes, stride=1):
        return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1 = conv3x3(inplanes, planes, stride)
            self.bn1 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = conv3x3(planes, planes)
            self.bn2 = nn.BatchNorm2d(planes)
            self.downsample = downsample
            self.stride = stride

        def forward(self, x):
            identity = x
            out = self.conv1(x)
            out = self.bn1(out)
            out = self.relu(out)
            out = self.conv2(out)
            out = self.bn2(out)
            if self.downsample is not None:
                identity = self.downsample(x)
            out += identity
            out = self.relu(out)
            return out

    class _ResNet18(nn.Module):
        """conv1 .. layer3 of torchvision's ResNet(BasicBlock, [2, 2, 2, 2]) -- the attributes the reference reads
        (spatial_network.py:123-139, temporal_network.py:43-59); layer4 / avgpool / fc are never touched by it."""

        def __init__(self):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
            self.bn1 = nn.BatchNorm2d(self.inplanes)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
            self.layer1 = self._make_layer(64, 2)
            self.layer2 = self._make_layer(128, 2, stride=2)
            self.layer3 = self._make_layer(256, 2, stride=2)

        def _make_layer(self, planes, blocks, stride=1):
            downsample = None
            if stride != 1 or self.inplanes != planes * BasicBlock.expansion:
                downsample = nn.Sequential(conv1x1(self.inplanes, planes * BasicBlock.expansion, stride),
                                           nn.BatchNorm2d(planes * BasicBlock.expansion))
            layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
            self.inplanes = planes * BasicBlock.expansion
            for _ in range(1, blocks):
                layers.append(BasicBlock(self.inplanes, planes))
            return nn.Sequential(*layers)

    # ---- torchvision/transforms/functional_tensor.py (0.14.1): gaussian_blur
    def _get_gaussian_kernel1d(kernel_size, sigma):
        ksize_half = (kernel_size - 1) * 0.5
        x = torch.linspace(-ksize_half, ksize_half, steps=kernel_size)
        pdf = torch.exp(-0.5 * (x / sigma).pow(2))
        return pdf / pdf.sum()

    def _gaussian_blur(img, kernel_size, sigma):
        kx = _get_gaussian_kernel1d(kernel_size[0], sigma[0]).to(img.device, dtype=img.dtype)
        ky = _get_gaussian_kernel1d(kernel_size[1], sigma[1]).to(img.device, dtype=img.dtype)
        kernel = torch.mm(ky[:, None], kx[None, :])
        kernel = kernel.expand(img.shape[-3], 1, kernel.shape[0], kernel.shape[1])
        padding = [kernel_size[0] // 2, kernel_size[0] // 2, kernel_size[1] // 2, kernel_size[1] // 2]
        img = F.pad(img, padding, mode='reflect')
        return F.conv2d(img, kernel, groups=img.shape[-3])

    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvr = types.ModuleType('torchvision.models.resnet')
    tvt = types.ModuleType('torchvision.transforms')
    tvr.resnet18 = lambda *a, **k: _ResNet18()
    tvm.resnet = tvr
    tvm.resnet18 = tvr.resnet18

    class GaussianBlur:
        """transforms.GaussianBlur(kernel_size, sigma) with a scalar sigma: sigma range (s, s), so every call blurs with s"""

        def __init__(self, kernel_size, sigma):
            self.kernel_size = tuple(kernel_size)
            self.sigma = (float(sigma), float(sigma))

        def __call__(self, x):
            return _gaussian_blur(x, self.kernel_size, self.sigma)

    tvt.GaussianBlur = GaussianBlur
    tv.models = tvm
    tv.transforms = tvt
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm,
                        'torchvision.models.resnet': tvr, 'torchvision.transforms': tvt})

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
