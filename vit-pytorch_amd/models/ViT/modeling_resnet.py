"""BiT ResNetV2 stem of the R50+ViT hybrid (drop-in for ``/root/reference/models/ViT/modeling_resnet.py``).

``StdConv2d`` (:35-42), ``PreActBottleneck`` (:55-127) and ``ResNetV2`` (:129-164) keep their
constructor arguments, attribute names and ``state_dict`` keys.  The stem is convolution /
GroupNorm work on image-shaped tensors, outside the token path the HIP kernel library covers: it
runs on PyTorch-ROCm's own operators (MIOpen convolutions) in fp32 and hands its [B, 16 width, H/16,
W/16] feature map to the patch projection of ``modeling.Embeddings``, which is a GEMM on the kernel
library again.  Caveat measured on MI355X with PyTorch 2.10 + ROCm 7.0 (tools/stem_check.py probes
it): the stem's fp32 parameter gradients from the stock convolution backward differ from the CPU's
by 1-8 % in roughly every second process (MIOpen picks its algorithms by timing), and always with
MIOpen disabled; every operator alone, the forward pass, and the gradient handed to the stem agree
to a few 1e-6.  As in this reference (which differs from upstream ViT-pytorch here), the root pool
has no padding and the body has three stages, so the map is 1/16 of the image.
"""
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.nn as nn
import torch.nn.functional as F


def np2th(weights, conv=False):
    """HWIO -> OIHW for convolution kernels."""
    if conv:
        weights = weights.transpose([3, 2, 0, 1])
    return torch.from_numpy(weights)


class StdConv2d(nn.Conv2d):
    """Convolution with weight standardisation over (in, kh, kw) per output channel, eps 1e-5."""

    def forward(self, x):
        w = self.weight
        v, m = torch.var_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        return F.conv2d(x, (w - m) / torch.sqrt(v + 1e-5), self.bias, self.stride, self.padding,
                        self.dilation, self.groups)


def conv3x3(cin, cout, stride=1, groups=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=bias, groups=groups)


def conv1x1(cin, cout, stride=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=1, stride=stride, padding=0, bias=bias)


class PreActBottleneck(nn.Module):
    """Bottleneck unit: 1x1 -> 3x3 (carries the stride) -> 1x1, GroupNorm(32, eps=1e-6) + ReLU after
    the first two, GroupNorm after the third, projection shortcut (1x1 + per-channel GroupNorm) when
    the shape changes, ReLU on the sum."""

    def __init__(self, cin, cout=None, cmid=None, stride=1):
        super().__init__()
        cout = cout or cin
        cmid = cmid or cout // 4
        self.gn1 = nn.GroupNorm(32, cmid, eps=1e-6)
        self.conv1 = conv1x1(cin, cmid, bias=False)
        self.gn2 = nn.GroupNorm(32, cmid, eps=1e-6)
        self.conv2 = conv3x3(cmid, cmid, stride, bias=False)
        self.gn3 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv3 = conv1x1(cmid, cout, bias=False)
        self.relu = nn.ReLU(inplace=True)
        if stride != 1 or cin != cout:
            self.downsample = conv1x1(cin, cout, stride, bias=False)
            self.gn_proj = nn.GroupNorm(cout, cout)

    def forward(self, x):
        residual = x
        if hasattr(self, "downsample"):
            residual = self.gn_proj(self.downsample(x))
        y = self.relu(self.gn1(self.conv1(x)))
        y = self.relu(self.gn2(self.conv2(y)))
        y = self.gn3(self.conv3(y))
        return self.relu(residual + y)

    def load_from(self, weights, n_block, n_unit):
        with torch.no_grad():
            for i in (1, 2, 3):
                getattr(self, f"conv{i}").weight.copy_(np2th(weights[pjoin(n_block, n_unit, f"conv{i}/kernel")], conv=True))
                gn = getattr(self, f"gn{i}")
                gn.weight.copy_(np2th(weights[pjoin(n_block, n_unit, f"gn{i}/scale")]).view(-1))
                gn.bias.copy_(np2th(weights[pjoin(n_block, n_unit, f"gn{i}/bias")]).view(-1))
            if hasattr(self, "downsample"):
                self.downsample.weight.copy_(np2th(weights[pjoin(n_block, n_unit, "conv_proj/kernel")], conv=True))
                self.gn_proj.weight.copy_(np2th(weights[pjoin(n_block, n_unit, "gn_proj/scale")]).view(-1))
                self.gn_proj.bias.copy_(np2th(weights[pjoin(n_block, n_unit, "gn_proj/bias")]).view(-1))


class ResNetV2(nn.Module):
    """Root (7x7/2 conv, GroupNorm, ReLU, 3x3/2 max-pool without padding) and three stages of
    bottleneck units with widths 4, 8, 16 x ``width`` (strides 1, 2, 2)."""

    def __init__(self, block_units, width_factor):
        super().__init__()
        width = int(64 * width_factor)
        self.width = width
        self.root = nn.Sequential(OrderedDict([
            ("conv", StdConv2d(3, width, kernel_size=7, stride=2, bias=False, padding=3)),
            ("gn", nn.GroupNorm(32, width, eps=1e-6)),
            ("relu", nn.ReLU(inplace=True)),
            ("pool", nn.MaxPool2d(kernel_size=3, stride=2, padding=0)),
        ]))

        def stage(cin, cout, cmid, stride, units):
            first = [("unit1", PreActBottleneck(cin=cin, cout=cout, cmid=cmid, stride=stride))]
            rest = [(f"unit{i:d}", PreActBottleneck(cin=cout, cout=cout, cmid=cmid)) for i in range(2, units + 1)]
            return nn.Sequential(OrderedDict(first + rest))

        self.body = nn.Sequential(OrderedDict([
            ("block1", stage(width, width * 4, width, 1, block_units[0])),
            ("block2", stage(width * 4, width * 8, width * 2, 2, block_units[1])),
            ("block3", stage(width * 8, width * 16, width * 4, 2, block_units[2])),
        ]))

    def forward(self, x):
        return self.body(self.root(x))
