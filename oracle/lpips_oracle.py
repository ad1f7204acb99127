"""CPU ORACLE for the LPIPS perceptual term — test infrastructure, NOT product code (only tests/ may import it).

lpips 0.1.4 (the reference's pinned dependency, requirements.txt:2; call sites enhancing/losses/vqperceptual.py:29,43,74,115) is NOT vendored under
/root/reference and cannot be installed here (no network), and neither can its pretrained weights.  This file restates its PUBLISHED algorithm for
``lpips.LPIPS(net="vgg")`` in plain fp32 PyTorch:

  lpips/lpips.py            LPIPS.forward:   in0/in1 -> scaling_layer -> net.forward -> per slice: normalize_tensor, squared difference,
                                             lins[k] (NetLinLayer: Dropout + 1x1 Conv2d(C, 1, bias=False)), spatial_average(keepdim=True); summed over the 5 slices
                            normalize_tensor(x, eps=1e-10) = x / (sqrt(sum_c x^2) + eps) ;  ScalingLayer: (inp - shift) / scale,
                                             shift = (-.030, -.088, -.188), scale = (.458, .448, .450) ; ``normalize=True`` maps [0,1] inputs with 2x - 1
  lpips/pretrained_networks.py  vgg16:       torchvision vgg16.features split at [0,4) [4,9) [9,16) [16,23) [23,30) -> relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
                                             (conv 3x3 pad 1 + ReLU, MaxPool2d(2, 2) opening slices 2..5)

PARITY UNPINNED: with neither the package nor its weights there is no golden vector to check this restatement against; it is anchored on the
reference's call sites (inputs*2-1 in, ``.mean()`` of the [B,1,1,1] result out) and on the published source.  What the tests pin is that the HIP
path equals THIS restatement on identical (random) weights.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SLICES = [[0, 2], [5, 7], [10, 12, 14], [17, 19, 21], [24, 26, 28]]
SHIFT = torch.tensor([-.030, -.088, -.188])[None, :, None, None]
SCALE = torch.tensor([.458, .448, .450])[None, :, None, None]


def vgg_features(x: torch.Tensor, sd: Dict[str, torch.Tensor]):
    """pretrained_networks.vgg16.forward: the five slice outputs."""
    outs = []
    for k, idxs in enumerate(SLICES):
        if k > 0:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        for i in idxs:
            x = F.relu(F.conv2d(x, sd[f"net.slice{k + 1}.{i}.weight"], sd[f"net.slice{k + 1}.{i}.bias"], padding=1))
        outs.append(x)
    return outs


def normalize_tensor(x: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def lpips_distance(in0: torch.Tensor, in1: torch.Tensor, sd: Dict[str, torch.Tensor], normalize: bool = False) -> torch.Tensor:
    """LPIPS.forward(in0, in1, normalize) -> [B,1,1,1]"""
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    shift, scale = sd.get("scaling_layer.shift", SHIFT), sd.get("scaling_layer.scale", SCALE)
    f0, f1 = vgg_features((in0 - shift) / scale, sd), vgg_features((in1 - shift) / scale, sd)
    val = 0
    for k in range(5):
        d = (normalize_tensor(f0[k]) - normalize_tensor(f1[k])) ** 2
        val = val + F.conv2d(d, sd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return val
