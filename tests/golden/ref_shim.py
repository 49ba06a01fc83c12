"""Import shim for running the reference (`/root/reference`, Python/PyTorch) on CPU in the
BUILD CONTAINER ONLY (it does not exist on the GPU box; nothing under tests/ that runs there
imports this file).  See SURVEY.md Appendix B.

 * never write __pycache__ into the read-only reference tree
 * cv2 / imageio / skimage are imported by every reference file but unused on the tensor path
 * torchvision is absent: `models.resnet.resnet18(weights=...)` is replaced by a local ResNet-18
   with torchvision's attribute names (architecture restated, weights always come from the
   checkpoint), `transforms.GaussianBlur` by the restatement in oracle.pipeline
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

    from oracle import nets as ON
    from oracle import pipeline as OP

    class _ResNet18(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = ON._layer(64, 64, 1)
            self.layer2 = ON._layer(64, 128, 2)
            self.layer3 = ON._layer(128, 256, 2)

    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvr = types.ModuleType('torchvision.models.resnet')
    tvt = types.ModuleType('torchvision.transforms')
    tvr.resnet18 = lambda *a, **k: _ResNet18()
    tvm.resnet = tvr
    tvm.resnet18 = tvr.resnet18

    class GaussianBlur:
        def __init__(self, kernel_size, sigma):
            assert tuple(kernel_size) == (21, 21) and float(sigma) == 20.0

        def __call__(self, x):
            return OP.gaussian_blur_21_20(x)

    tvt.GaussianBlur = GaussianBlur
    tv.models = tvm
    tv.transforms = tvt
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm,
                        'torchvision.models.resnet': tvr, 'torchvision.transforms': tvt})

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
