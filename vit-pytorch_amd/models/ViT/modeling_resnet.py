"""BiT ResNetV2 stem of the R50+ViT hybrid (drop-in for ``/root/reference/models/ViT/modeling_resnet.py``).

``StdConv2d`` (:35-42), ``PreActBottleneck`` (:55-127) and ``ResNetV2`` (:129-164) keep their
constructor arguments, attribute names and ``state_dict`` keys; the containers' own ``forward`` is not
used.  On the GPU the stem runs channels-last (``[B, H, W, C]``: the layout in which a 1x1 convolution is
a GEMM over the ``B*H*W`` token rows) on the kernel library (``vitk.stem_ops``):

* weight standardisation and its backward: ``vitk_wstd_fwd/bwd``;
* every 1x1 convolution (two of the three per unit, and the projection shortcuts): the bf16 / fp32 MFMA
  GEMMs of the transformer path, with the weight-gradient GEMM;
* GroupNorm + ReLU, GroupNorm, and the residual add + ReLU: fused HBM-bound kernels;
* the 3x3 and the 7x7 root convolution: ``im2col`` / ``col2im`` kernels around the same GEMMs (forward, input
  gradient and weight gradient), so the activations never leave the channels-last layout;
* the 3x3 max-pool of the root: PyTorch.

The fp32 residual stream of the stem is kept in fp32; GEMM operands are in ``vitk.get_compute_dtype()``.
As in this reference (which differs from upstream ViT-pytorch here), the root pool has no padding and the
body has three stages, so the map is 1/16 of the image.  There is no CPU path.
"""
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.nn as nn
import torch.nn.functional as F

import vitk
from vitk import ops, stem_ops


def np2th(weights, conv=False):
    """HWIO -> OIHW for convolution kernels."""
    if conv:
        weights = weights.transpose([3, 2, 0, 1])
    return torch.from_numpy(weights)


class StdConv2d(nn.Conv2d):
    """Convolution with weight standardisation over (in, kh, kw) per output channel, eps 1e-5.
    ``forward_cl`` takes and returns channels-last tensors ``[B, H, W, C]``: ``x`` is the fp32 stream,
    ``x_lp`` the same values in the compute dtype (what a 1x1 convolution, a GEMM, consumes)."""

    def forward(self, x):
        return self.forward_cl(x.permute(0, 2, 3, 1).contiguous(), None).permute(0, 3, 1, 2)

    def rows_cl(self, x, x_lp, nchw=False):
        """The convolution as a GEMM: (rows [B*Ho*Wo, K] in the compute dtype, standardised weight [O, K] fp32,
        (B, Ho, Wo)).  A 1x1 convolution reads ``x_lp`` (strided rows when it carries a stride), a KxK one the
        im2col of the fp32 ``x``."""
        if self.bias is not None or self.groups != 1 or self.dilation != (1, 1):
            raise vitk.VitkError("StdConv2d: only bias-free, ungrouped, undilated convolutions are implemented")
        w_hat = stem_ops.weight_std(self.weight, 1e-5)
        cdt = vitk.get_compute_dtype()
        k, s, pad = self.kernel_size[0], self.stride[0], self.padding[0]
        if k == 1:
            src = x_lp if x_lp is not None else (x if x.dtype == cdt else x.to(cdt))
            if s != 1:
                src = src[:, ::s, ::s, :]
            B, H, W, C = src.shape
            return src.reshape(B * H * W, C), w_hat.view(self.out_channels, C), (B, H, W)
        return stem_ops.conv_rows(x, w_hat, s, pad, cdt, nchw)

    def forward_cl(self, x, x_lp):
        rows, w, (B, H, W) = self.rows_cl(x, x_lp)
        zero = torch.zeros(self.out_channels, dtype=torch.float32, device=rows.device)
        y = ops.linear(rows, (w,), (zero,), out_dtype=torch.float32)
        return y.view(B, H, W, self.out_channels)

    def gn_cl(self, gn, x, x_lp, relu, out_dtype, residual=None, lp_dtype=None, nchw=False):
        """``gn(conv(x))`` [+ ReLU] as one graph node, channels-last ``[B, Ho, Wo, O]`` in ``out_dtype``; with
        ``residual``: both copies of ``relu(residual + gn(conv(x)))`` (fp32, ``lp_dtype``)."""
        rows, w, (B, H, W) = self.rows_cl(x, x_lp, nchw)
        shape = (B, H, W, self.out_channels)
        r = stem_ops.conv_gn(rows, w, gn, B, H * W, relu, out_dtype, residual, lp_dtype)
        if residual is None:
            return r.view(shape)
        return r[0].view(shape), (r[1].view(shape) if r[1] is not None else None)


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
        x = x.permute(0, 2, 3, 1).contiguous()
        out, _ = self.forward_cl(x, None)
        return out.permute(0, 3, 1, 2)

    def forward_cl(self, x, x_lp):
        """x: [B, H, W, C] fp32 stream; x_lp: the same in the compute dtype (or None).  Returns both forms
        of relu(residual + branch)."""
        cdt = vitk.get_compute_dtype()
        if x_lp is None and cdt != torch.float32:
            x_lp = x.to(cdt)
        residual = x
        if hasattr(self, "downsample"):
            residual = self.downsample.gn_cl(self.gn_proj, x, x_lp, False, torch.float32)
        y = self.conv1.gn_cl(self.gn1, x, x_lp, True, torch.float32)        # fp32: the 3x3's im2col casts it
        y = self.conv2.gn_cl(self.gn2, y, None, True, cdt)                  # compute dtype: feeds a GEMM
        return self.conv3.gn_cl(self.gn3, None, y, False, torch.float32, residual=residual, lp_dtype=cdt)

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
        """x: [B, 3, H, W] image -> [B, 16 width, H/16, W/16] feature map (an NCHW-shaped view of channels-last
        memory)."""
        if not x.is_cuda:
            raise vitk.VitkError("the hybrid stem runs on the GPU only: got a tensor on " + str(x.device))
        cdt = vitk.get_compute_dtype()
        root = self.root
        y = root.conv.gn_cl(root.gn, x.float(), None, True, torch.float32, nchw=True)
        y = root.pool(y.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
        y_lp = None
        for block in self.body.children():
            for unit in block.children():
                y, y_lp = unit.forward_cl(y, y_lp)
        return y.permute(0, 3, 1, 2)
