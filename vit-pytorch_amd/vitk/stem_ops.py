"""Host side of the R50 hybrid stem (reference models/ViT/modeling_resnet.py): one ``autograd.Function``
per operator group, on channels-last activations ``[B, H, W, C]``.

* weight standardisation of ``StdConv2d`` (:35-42) and its backward: ``vitk_wstd_fwd/bwd``;
* ``GroupNorm`` (+ ``ReLU``) (:55-91): ``vitk_groupnorm_fwd/bwd``;
* the 1x1 convolutions are GEMMs over the ``B*H*W`` token rows on the MFMA kernels (``ops.linear``);
* the 3x3 and 7x7 convolutions are ``im2col`` (``vitk_im2col_conv``) + the same GEMMs, with ``col2im``
  (``vitk_col2im_conv``) for the input gradient;
* the residual add + ReLU that ends a unit (:91): ``vitk_add_relu_fwd/bwd``;
* a convolution GEMM and the GroupNorm behind it as ONE graph node (``conv_gn``), so that the gradient between the
  two is written once, in the operand type of the backward GEMMs, and the unit's last GroupNorm writes
  ``relu(residual + branch)`` directly (``vitk_groupnorm_add_relu_fwd``).
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
                                     _ptr(dbeta), _ptr(ws), B, HW, C, groups, int(relu), dtype_code(dy.dtype), F32,
                                     _stream()), "vitk_groupnorm_bwd")
        return dx, dgamma, dbeta, None, None, None, None


def group_norm(x, gamma, beta, groups, eps, relu, out_dtype):
    return GroupNormFn.apply(x, gamma, beta, groups, eps, relu, out_dtype)


# ----------------------------------------------------------------------------- convolution GEMM + GroupNorm
_ZERO_BIAS = {}     # (device, O) -> fp32 zeros, read only: the fp32-output GEMMs of the library all take a bias


def _zero_bias(O, device):
    key = (device, O)
    if key not in _ZERO_BIAS:
        _ZERO_BIAS[key] = torch.zeros(O, dtype=torch.float32, device=device)
    return _ZERO_BIAS[key]


class ConvGnFn(torch.autograd.Function):
    """``GroupNorm(src @ w^T)`` [+ ReLU], or ``relu(residual + GroupNorm(src @ w^T))`` when ``residual`` is given
    (reference PreActBottleneck.forward, models/ViT/modeling_resnet.py:78-91: ``gn(conv(x))`` pairs and the sum that
    ends the unit).  ``src``: ``[B*HW, K]`` rows in the compute dtype (channels-last activations for a 1x1
    convolution, im2col rows for a KxK one); ``w``: ``[O, K]`` fp32 (the standardised weight).

    One node instead of a linear and a GroupNorm node: the GEMM result stays fp32 (GroupNorm's statistics are taken
    from it as before), but its gradient, whose only readers are the two backward GEMMs, is written by GroupNorm's
    backward directly in their operand type instead of in fp32 followed by a conversion pass; same values."""

    @staticmethod
    def forward(ctx, src, w, gamma, beta, residual, B, HW, groups, eps, relu, out_dtype, lp_dtype):
        _need_gpu(src, w, gamma, beta, residual)
        src = src if src.is_contiguous() else src.contiguous()
        M, K = src.shape
        O = w.shape[0]
        if M != B * HW or w.numel() != O * K:
            raise VitkError(f"conv_gn: {tuple(src.shape)} rows against B*HW={B * HW}, weight {tuple(w.shape)}")
        need_t = ctx.needs_input_grad[0]
        wo, wt = ops.weight_operands([w.reshape(O, K)], src.dtype, True, need_t)
        y = torch.empty((M, O), dtype=torch.float32, device=src.device)
        ops.gemm_nt(src, wo, y, ops.EPI_BIAS, bias=_zero_bias(O, src.device))
        g, b = _f32(gamma), _f32(beta)
        mean = torch.empty(B * groups, dtype=torch.float32, device=src.device)
        rstd = torch.empty(B * groups, dtype=torch.float32, device=src.device)
        nws = lib.vitk_groupnorm_workspace(B, HW, O, groups)
        if nws == 0:
            raise VitkError(f"GroupNorm: unsupported shape B={B} HW={HW} C={O} groups={groups}")
        ws = torch.empty(nws, dtype=torch.uint8, device=src.device)
        if residual is None:
            out = torch.empty((M, O), dtype=out_dtype, device=src.device)
            check(lib.vitk_groupnorm_fwd(_ptr(y), _ptr(g), _ptr(b), _ptr(out), _ptr(mean), _ptr(rstd), _ptr(ws), B, HW, O,
                                         groups, float(eps), int(bool(relu)), dtype_code(out_dtype), _stream()),
                  "vitk_groupnorm_fwd")
            out_lp = None
            ctx.save_for_backward(src, wt, y, g, b, mean, rstd)
        else:
            if residual.dtype != torch.float32 or residual.numel() != M * O:
                raise VitkError("conv_gn: the residual is the fp32 stream, one value per output element")
            residual = residual if residual.is_contiguous() else residual.contiguous()
            out = torch.empty((M, O), dtype=torch.float32, device=src.device)
            want_lp = lp_dtype is not None and lp_dtype != torch.float32
            out_lp = torch.empty((M, O), dtype=lp_dtype, device=src.device) if want_lp else None
            check(lib.vitk_groupnorm_add_relu_fwd(_ptr(y), _ptr(g), _ptr(b), _ptr(residual), _ptr(out), _ptr(out_lp),
                                                  _ptr(mean), _ptr(rstd), _ptr(ws), B, HW, O, groups, float(eps),
                                                  dtype_code(lp_dtype) if want_lp else F32, _stream()),
                  "vitk_groupnorm_add_relu_fwd")
            ctx.save_for_backward(src, wt, y, g, b, mean, rstd, out)
        ctx.meta = (B, HW, O, K, groups, bool(relu) and residual is None, nws, residual is not None)
        ctx.gparams = (gamma, beta)
        return out if residual is None else (out, out_lp)

    @staticmethod
    def backward(ctx, g1, g2=None):
        B, HW, O, K, groups, relu, nws, fused_sum = ctx.meta
        if fused_sum and g1 is None and g2 is None:
            return (None,) * 12
        src = ctx.saved_tensors[0]
        M = B * HW
        dz = torch.empty((M, O), dtype=src.dtype, device=src.device)
        pg, pb = ctx.gparams
        ws = torch.empty(nws, dtype=torch.uint8, device=src.device)
        if fused_sum:
            src, wt, y, g, b, mean, rstd, out = ctx.saved_tensors
            if g1 is None:
                g1, g2 = g2.float(), None
            elif g1.dtype != torch.float32:
                g1 = g1.float()
            g1 = g1 if g1.is_contiguous() else g1.contiguous()
            if g2 is not None and not g2.is_contiguous():
                g2 = g2.contiguous()
            dgamma, dbeta = ops.grad_buf([pg], O, src.device), ops.grad_buf([pb], O, src.device)
            dres = torch.empty_like(out)            # gradient of both addends: the residual's, and GroupNorm's input
            check(lib.vitk_groupnorm_add_relu_bwd(_ptr(g1), _ptr(g2), _ptr(out), _ptr(y), _ptr(g), _ptr(b), _ptr(mean),
                                                  _ptr(rstd), _ptr(dres), _ptr(dz), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                                                  B, HW, O, groups, dtype_code(g2.dtype) if g2 is not None else F32,
                                                  dtype_code(dz.dtype), _stream()), "vitk_groupnorm_add_relu_bwd")
        else:
            src, wt, y, g, b, mean, rstd = ctx.saved_tensors
            dy = g1 if g1.is_contiguous() else g1.contiguous()
            dres = None
            dgamma, dbeta = ops.grad_buf([pg], O, src.device), ops.grad_buf([pb], O, src.device)
            check(lib.vitk_groupnorm_bwd(_ptr(dy), _ptr(y), _ptr(g), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(dz), _ptr(dgamma),
                                         _ptr(dbeta), _ptr(ws), B, HW, O, groups, int(relu), dtype_code(dy.dtype),
                                         dtype_code(dz.dtype), _stream()), "vitk_groupnorm_bwd")
        dsrc = dw = None
        if ctx.needs_input_grad[0]:
            dsrc = torch.empty((M, K), dtype=src.dtype, device=src.device)
            ops.gemm_nt(dz, wt, dsrc, ops.EPI_NONE)
        if ctx.needs_input_grad[1]:
            dw = ops.zeros_f32(O * K, src.device).view(O, K)
            ops.gemm_tn_acc(dz, src, dw)
        return dsrc, dw, dgamma, dbeta, dres, None, None, None, None, None, None, None


def conv_gn(src, w, gn, B, HW, relu, out_dtype, residual=None, lp_dtype=None):
    """``src`` rows ``[B*HW, K]`` x standardised weight ``w`` ``[O, ...]`` -> GroupNorm ``gn`` (an ``nn.GroupNorm``)
    [+ ReLU] -> ``[B*HW, O]`` in ``out_dtype``; with ``residual`` (``[B*HW, O]`` fp32):
    ``(relu(residual + GroupNorm(...)) fp32, the same in lp_dtype or None)``."""
    if residual is not None:
        residual = residual.reshape(B * HW, -1)
    return ConvGnFn.apply(src, w, gn.weight, gn.bias, residual, B, HW, gn.num_groups, gn.eps, relu, out_dtype, lp_dtype)


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


def conv_rows(x, w, stride, pad, cdt, nchw=False):
    """The two GEMM operands of a KxK convolution without bias: im2col rows ``[B*Ho*Wo, Kp]`` in ``cdt`` and the
    weight reordered to ``[O, Kp]`` = ``[O, (kh, kw, c)]`` + zero padding (torch ops on the weight-shaped tensor;
    their autograd carries dW back).  x: ``[B, H, W, C]`` fp32 (or an NCHW image when ``nchw``), w: ``[O, C, K, K]``
    fp32.  Returns (cols, wk, (B, Ho, Wo))."""
    O, C, k, _ = w.shape
    cols = Im2colFn.apply(x, k, stride, pad, cdt, nchw)
    Kp = cols.shape[1]
    wk = w.permute(0, 2, 3, 1).reshape(O, k * k * C)
    if Kp != wk.shape[1]:
        wk = F.pad(wk, (0, Kp - wk.shape[1]))
    if nchw:
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
    else:
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return cols, wk, (B, Ho, Wo)


def conv2d_cl(x, w, stride, pad, cdt, nchw=False):
    """KxK convolution without bias as im2col + GEMM -> ``[B, Ho, Wo, O]`` fp32 (see conv_rows)."""
    cols, wk, (B, Ho, Wo) = conv_rows(x, w, stride, pad, cdt, nchw)
    zero = torch.zeros(wk.shape[0], dtype=torch.float32, device=w.device)
    y = ops.linear(cols, (wk,), (zero,), out_dtype=torch.float32)
    return y.view(B, Ho, Wo, wk.shape[0])
