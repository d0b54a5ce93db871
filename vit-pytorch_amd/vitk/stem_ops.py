"""Host side of the R50 hybrid stem (reference models/ViT/modeling_resnet.py): one ``autograd.Function``
per operator group, on channels-last activations ``[B, H, W, C]``.

* weight standardisation of ``StdConv2d`` (:35-42) and its backward: ``vitk_wstd_fwd/bwd``;
* ``GroupNorm`` (+ ``ReLU``) (:55-91): ``vitk_groupnorm_fwd/bwd``;
* the 1x1 convolutions are GEMMs over the ``B*H*W`` token rows on the MFMA kernels (``ops.linear``);
* the 3x3 and 7x7 convolutions are ``im2col`` (``vitk_im2col_conv``) + the same GEMMs, with ``col2im``
  (``vitk_col2im_conv``) for the input gradient;
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
class Im2colFn(torch.autograd.Function):
    """cols[(b, oy, ox)][(kh, kw, c)] of a KH x KW window walk over x, in the compute dtype, and its adjoint.
    ``x``: fp32, either channels-last ``[B, H, W, C]`` (``nchw=False``) or an NCHW image (``nchw=True``: the root)."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, cdt, nchw):
        _need_gpu(x)
        if x.dtype != torch.float32:
            raise VitkError("im2col input must be fp32")
        if nchw:
            B, C, H, W = x.shape
            sb, sc, sh, sw = x.stride()
        else:
            x = x if x.is_contiguous() else x.contiguous()
            B, H, W, C = x.shape
            sb, sh, sw, sc = x.stride()
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        K = k * k * C
        Kp = (K + 63) // 64 * 64 if cdt == torch.bfloat16 else (K + 7) // 8 * 8      # the bf16 GEMMs reduce 64 deep
        cols = torch.empty((B * Ho * Wo, Kp), dtype=cdt, device=x.device)
        check(lib.vitk_im2col_conv(_ptr(x), _ptr(cols), B, H, W, C, k, k, stride, pad, sb, sh, sw, sc, Kp,
                                   dtype_code(cdt), _stream()), "vitk_im2col_conv")
        ctx.meta = (B, H, W, C, k, stride, pad, Kp, nchw)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        B, H, W, C, k, stride, pad, Kp, nchw = ctx.meta
        if nchw:
            raise VitkError("im2col of an NCHW image has no input gradient")
        dcols = dcols if dcols.is_contiguous() else dcols.contiguous()
        dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dcols.device)
        check(lib.vitk_col2im_conv(_ptr(dcols), _ptr(dx), B, H, W, C, k, k, stride, pad, Kp, dtype_code(dcols.dtype),
                                   _stream()), "vitk_col2im_conv")
        return dx, None, None, None, None, None


def conv2d_cl(x, w, stride, pad, cdt, nchw=False):
    """KxK convolution without bias as im2col + GEMM.  x: ``[B, H, W, C]`` fp32 (or an NCHW image when ``nchw``),
    w: ``[O, C, K, K]`` fp32 -> ``[B, Ho, Wo, O]`` fp32.  The weight is reordered to ``[O, (kh, kw, c)]`` with
    torch ops on the weight-shaped tensor (their autograd carries dW back)."""
    O, C, k, _ = w.shape
    cols = Im2colFn.apply(x, k, stride, pad, cdt, nchw)
    Kp = cols.shape[1]
    wk = w.permute(0, 2, 3, 1).reshape(O, k * k * C)
    if Kp != wk.shape[1]:
        wk = F.pad(wk, (0, Kp - wk.shape[1]))
    zero = torch.zeros(O, dtype=torch.float32, device=w.device)
    y = ops.linear(cols, (wk,), (zero,), out_dtype=torch.float32)
    if nchw:
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
    else:
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return y.view(B, Ho, Wo, O)
