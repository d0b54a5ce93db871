"""Host side of the R50 hybrid stem (reference models/ViT/modeling_resnet.py): one ``autograd.Function``
per operator group, on channels-last activations ``[B, H, W, C]``.

* weight standardisation of ``StdConv2d`` (:35-42) and its backward: ``vitk_wstd_fwd/bwd``;
* ``GroupNorm`` (+ ``ReLU``) (:55-91): ``vitk_groupnorm_fwd/bwd``;
* the 1x1 convolutions are GEMMs over the ``B*H*W`` token rows on the MFMA kernels (``ops.linear``);
* the 3x3 and 7x7 convolutions keep MIOpen for the forward pass and the input gradient, but their WEIGHT
  gradient is ``im2col`` (``vitk_im2col_conv``) + the library's own weight-gradient GEMM: on this stack
  MIOpen's deterministic / untimed backward-weights algorithm returns 3x3 gradients that are 23 % off, and
  its timed search picks that algorithm in some processes;
* the residual add + ReLU that ends a unit (:91): ``vitk_add_relu_fwd/bwd``.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import F32, VitkError, check, dtype_code, lib, ops
from .ops import _f32, _need_gpu, _ptr, _stream


# ----------------------------------------------------------------------------- weight standardisation
class WeightStdFn(torch.autograd.Function):
    """w_hat = (w - mean) / sqrt(var + eps) per output channel over (in, kh, kw), biased variance."""

    @staticmethod
    def forward(ctx, w, eps):
        _need_gpu(w)
        w2 = _f32(w).reshape(w.shape[0], -1)
        O, Fdim = w2.shape
        w_hat = torch.empty_like(w2)
        rstd = torch.empty(O, dtype=torch.float32, device=w.device)
        check(lib.vitk_wstd_fwd(_ptr(w2), _ptr(w_hat), _ptr(rstd), O, Fdim, float(eps), _stream()), "vitk_wstd_fwd")
        ctx.save_for_backward(w_hat, rstd)
        ctx.wshape = w.shape
        return w_hat.view(w.shape)

    @staticmethod
    def backward(ctx, g):
        w_hat, rstd = ctx.saved_tensors
        O, Fdim = w_hat.shape
        g2 = g.reshape(O, Fdim)
        if g2.dtype != torch.float32:
            g2 = g2.float()
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        dw = torch.empty_like(w_hat)
        check(lib.vitk_wstd_bwd(_ptr(g2), _ptr(w_hat), _ptr(rstd), _ptr(dw), O, Fdim, _stream()), "vitk_wstd_bwd")
        return dw.view(ctx.wshape), None


def weight_std(w, eps=1e-5):
    return WeightStdFn.apply(w, eps)


# ----------------------------------------------------------------------------- GroupNorm (+ ReLU)
class GroupNormFn(torch.autograd.Function):
    """y = [relu](GroupNorm(x)) on x [B, H, W, C] fp32 (channels last); y in ``out_dtype``."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu, out_dtype):
        _need_gpu(x, gamma, beta)
        if x.dtype != torch.float32:
            raise VitkError("GroupNorm input must be the fp32 stream")
        x = x if x.is_contiguous() else x.contiguous()
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        g, b = _f32(gamma), _f32(beta)
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        nws = lib.vitk_groupnorm_workspace(B, HW, C, groups)
        if nws == 0:
            raise VitkError(f"GroupNorm: unsupported shape B={B} HW={HW} C={C} groups={groups}")
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        check(lib.vitk_groupnorm_fwd(_ptr(x), _ptr(g), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(ws), B, HW, C, groups,
                                     float(eps), int(bool(relu)), dtype_code(out_dtype), _stream()), "vitk_groupnorm_fwd")
        ctx.save_for_backward(x, g, b, mean, rstd)
        ctx.meta = (B, HW, C, groups, bool(relu), nws)
        ctx.gparams = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, rstd = ctx.saved_tensors
        B, HW, C, groups, relu, nws = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(x)
        pg, pb = ctx.gparams
        dgamma, dbeta = ops.grad_buf([pg], C, x.device), ops.grad_buf([pb], C, x.device)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        check(lib.vitk_groupnorm_bwd(_ptr(dy), _ptr(x), _ptr(g), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dgamma),
                                     _ptr(dbeta), _ptr(ws), B, HW, C, groups, int(relu), dtype_code(dy.dtype), _stream()),
              "vitk_groupnorm_bwd")
        return dx, dgamma, dbeta, None, None, None, None


def group_norm(x, gamma, beta, groups, eps, relu, out_dtype):
    return GroupNormFn.apply(x, gamma, beta, groups, eps, relu, out_dtype)


# ----------------------------------------------------------------------------- residual add + ReLU
class AddReluFn(torch.autograd.Function):
    """(out fp32, out_lp) = relu(a + b); out_lp is a second copy in ``lp_dtype`` (the next unit's GEMM
    operand), or None when ``lp_dtype`` is None / fp32."""

    @staticmethod
    def forward(ctx, a, b, lp_dtype):
        _need_gpu(a, b)
        if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape:
            raise VitkError("add_relu: two fp32 tensors of one shape")
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        out = torch.empty_like(a)
        want_lp = lp_dtype is not None and lp_dtype != torch.float32
        out_lp = torch.empty(a.shape, dtype=lp_dtype, device=a.device) if want_lp else None
        check(lib.vitk_add_relu_fwd(_ptr(a), _ptr(b), _ptr(out), _ptr(out_lp), a.numel(),
                                    dtype_code(lp_dtype) if want_lp else F32, _stream()), "vitk_add_relu_fwd")
        ctx.save_for_backward(out)
        return out, out_lp

    @staticmethod
    def backward(ctx, g1, g2):
        (out,) = ctx.saved_tensors
        if g1 is None and g2 is None:
            return None, None, None
        if g1 is None:                      # only the low-precision copy was used downstream
            g1, g2 = g2.float(), None
        elif g1.dtype != torch.float32:
            g1 = g1.float()
        g1 = g1 if g1.is_contiguous() else g1.contiguous()
        if g2 is not None and not g2.is_contiguous():
            g2 = g2.contiguous()
        d = torch.empty_like(out)
        check(lib.vitk_add_relu_bwd(_ptr(g1), _ptr(g2), _ptr(out), _ptr(d), out.numel(),
                                    dtype_code(g2.dtype) if g2 is not None else F32, _stream()), "vitk_add_relu_bwd")
        return d, d, None


def add_relu(a, b, lp_dtype):
    return AddReluFn.apply(a, b, lp_dtype)


# ----------------------------------------------------------------------------- KxK convolution
class ConvFn(torch.autograd.Function):
    """conv2d(x, w, stride, padding) without bias.  x: NCHW-shaped fp32 (any memory format), w: [O, C, KH, KW] fp32.
    Forward and input gradient: MIOpen.  Weight gradient: im2col + the library's weight-gradient GEMM in the
    compute dtype (fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, compute_dtype):
        _need_gpu(x, w)
        y = F.conv2d(x, w, None, stride, padding)
        ctx.save_for_backward(x, w)
        ctx.meta = (stride, padding, compute_dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, cdt = ctx.meta
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [padding, padding], [1, 1], False,
                                                     [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            O, C, KH, KW = w.shape
            B, _, H, W = x.shape
            Ho, Wo = dy.shape[2], dy.shape[3]
            K = KH * KW * C
            Kp = (K + 7) // 8 * 8
            cols = torch.empty((B * Ho * Wo, Kp), dtype=cdt, device=x.device)
            sb, sc, sh, sw = x.stride()
            check(lib.vitk_im2col_conv(_ptr(x), _ptr(cols), B, H, W, C, KH, KW, stride, padding, sb, sh, sw, sc, Kp,
                                       dtype_code(cdt), _stream()), "vitk_im2col_conv")
            dyt = dy.permute(0, 2, 3, 1).reshape(B * Ho * Wo, O)          # a view when dy is channels-last
            dyt = ops.convert(dyt.contiguous(), cdt)
            dwk = ops.zeros_f32(O * Kp, x.device).view(O, Kp)
            ops.gemm_tn_acc(dyt, cols, dwk)
            dw = dwk[:, :K].reshape(O, KH, KW, C).permute(0, 3, 1, 2).contiguous()
        return dx, dw, None, None, None


def conv2d(x, w, stride, padding, compute_dtype):
    return ConvFn.apply(x, w, stride, padding, compute_dtype)
