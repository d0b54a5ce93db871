"""Piggyback mask layers on the vitk HIP kernels (MI355X / gfx950).

Drop-in for the part of ``/root/reference/models/ViT/layers_PB.py`` that ``modeling_PB.py`` uses:
``Binarizer`` (:14-33), ``Ternarizer`` (:36-54) and ``ElementWiseLinear`` (:56-135) keep their
names, constructor arguments, ``forward(input, task)`` and ``state_dict`` keys (``weight``, ``bias``,
``mask_reals.<t>``).  ``ElementWiseRG`` / ``ElementWiseSFG`` (the RKR+piggyback combination of
``modeling_RKRPB.py``) are not provided.

The reference multiplies the thresholded mask into the weight and calls ``F.linear``.  The mask acts
on the weight side only, so here too the product ``mask_t * W`` is the operand handed to the fused
GEMM / attention / MLP operators (``effective``), built once per forward on the [out, in] weight;
the gradient of the real-valued mask is ``dW_t * W`` (straight-through estimator: the thresholding
passes the gradient unchanged) and comes out of autograd on that weight-shaped product.
"""
import copy

import torch
import torch.nn as nn

from vitk import ops

DEFAULT_THRESHOLD = 5e-3


class Binarizer(torch.autograd.Function):
    """{0, 1} mask of a real-valued tensor: 1 where it exceeds the threshold; identity gradient."""

    @staticmethod
    def forward(ctx, inputs):
        return (inputs > DEFAULT_THRESHOLD).to(inputs.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class Ternarizer(torch.autograd.Function):
    """{-1, 0, 1} mask: -1 below zero, 1 above the threshold, 0 in between; identity gradient."""

    @staticmethod
    def forward(ctx, inputs):
        return (inputs > DEFAULT_THRESHOLD).to(inputs.dtype) - (inputs < 0).to(inputs.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class ElementWiseLinear(nn.Module):
    """Linear layer whose (usually frozen) weight is masked by a per-task learned binary / ternary
    mask.  As in the reference, ``weight`` and ``bias`` are left uninitialised: they come from a
    checkpoint."""

    def __init__(self, in_features, out_features, config, bias=True, threshold=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.mask_scale, self.mask_init = config.mask_scale, config.mask_init
        self.threshold = DEFAULT_THRESHOLD if threshold is None else threshold   # the reference always applies the default
        self.info = {"threshold_fn": config.threshold_fn, "threshold": self.threshold}
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        mask_real = torch.empty(out_features, in_features)
        if self.mask_init == "1s":
            mask_real.fill_(self.mask_scale)
        elif self.mask_init == "uniform":
            mask_real.uniform_(-self.mask_scale, self.mask_scale)
        self.mask_reals = nn.ParameterList(
            [copy.deepcopy(nn.Parameter(mask_real)) for _ in range(config.task_num)])
        if config.threshold_fn == "binarizer":
            self.threshold_fn = Binarizer
        elif config.threshold_fn == "ternarizer":
            self.threshold_fn = Ternarizer
        else:
            raise ValueError(f"unknown threshold_fn {config.threshold_fn!r}")

    def effective(self, task, sfg=None):
        """(mask_t * W [out, in], bias) - what the GEMM multiplies by for this task."""
        if sfg is not None:
            raise ValueError("ElementWiseLinear takes no SFG scale")
        return self.threshold_fn.apply(self.mask_reals[task]) * self.weight, self.bias

    def forward(self, input, task):
        from . import modeling as base
        w, b = self.effective(task)
        shape = input.shape
        y = ops.linear(base._to_compute(input).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.out_features)

    def __repr__(self):
        return f"{self.__class__.__name__}(in_features={self.in_features}, out_features={self.out_features})"
