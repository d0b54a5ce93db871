"""Piggyback mask layers on the vitk HIP kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/layers_PB.py``: ``Binarizer`` (:14-33), ``Ternarizer``
(:36-54), ``ElementWiseLinear`` (:56-135, used by ``modeling_PB.py``), ``ElementWiseRG`` (:138-302)
and ``ElementWiseSFG`` (:306-368, the RKR+piggyback combination of ``modeling_RKRPB.py``) keep
their names, constructor arguments, ``forward(input, task)`` and ``state_dict`` keys (``weight``,
``bias``, ``mask_reals.<t>``; ``LM_base``, ``RM_base``, ``LM_list.<t>``, ``RM_list.<t>``,
``mask_reals_LM.<t-1>``, ``mask_reals_RM.<t-1>``, ``F_base``, ``F_list.<t>``, ``mask_reals_F.<t-1>``).

The reference multiplies the thresholded mask into the weight and calls ``F.linear``.  The mask acts
on the weight side only, so here too the product ``mask_t * W`` is the operand handed to the fused
GEMM / attention / MLP operators (``effective``), built once per forward on the [out, in] weight by
``vitk_mask_fold_fwd`` (threshold and product in one launch); the gradient of the real-valued mask is
``dW_t * W`` (straight-through estimator: the thresholding passes the gradient unchanged), that of the
masked tensor ``dW_t * mask_t``, both from one launch of ``vitk_mask_fold_bwd``.  The same pair masks the
rank-K factors and the feature scales of the RKR combination.
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
        return _masked(self, self.mask_reals[task], self.weight), self.bias

    def forward(self, input, task):
        from . import modeling as base
        w, b = self.effective(task)
        shape = input.shape
        if b is None:
            b = torch.zeros(self.out_features, device=w.device)
        y = ops.linear(base._to_compute(input).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.out_features)

    def __repr__(self):
        return f"{self.__class__.__name__}(in_features={self.in_features}, out_features={self.out_features})"


def _masked(layer, mask_real, base):
    """hard(mask_real) * base: one kernel launch on the GPU, the two autograd functions above elsewhere."""
    if base.is_cuda:
        return ops.mask_fold(mask_real, base, layer.threshold_fn is Ternarizer, DEFAULT_THRESHOLD)
    return layer.threshold_fn.apply(mask_real) * base


def _threshold_fn(config):
    if config.threshold_fn == "binarizer":
        return Binarizer
    if config.threshold_fn == "ternarizer":
        return Ternarizer
    raise ValueError(f"unknown threshold_fn {config.threshold_fn!r}")


def _mask_list(like, config, count):
    """``count`` real-valued masks shaped like ``like``, initialised per ``config.mask_init``."""
    mask_real = torch.empty(like.shape)
    if config.mask_init == "1s":
        mask_real.fill_(config.mask_scale)
    elif config.mask_init == "uniform":
        mask_real.uniform_(-config.mask_scale, config.mask_scale)
    return nn.ParameterList([copy.deepcopy(nn.Parameter(mask_real)) for _ in range(count)])


class ElementWiseRG(nn.Module):
    """Linear layer with a rank-K weight correction whose factors are shared by all tasks and
    specialised per task by piggyback masks (reference :138-302).  Task 0 uses the shared factors
    as they are; task t > 0 uses mask ``t - 1``.  The configuration flags pick which factor is
    masked and which is a plain per-task list:

    ============  =======================================  ================================
    flag          left factor [in, K]                      right factor [K, out]
    ============  =======================================  ================================
    ``RGnoPB``    ``LM_list[t]``                           ``RM_list[t]``
    ``LMnoPB``    ``LM_list[t]``                           ``mask_RM[t-1] * RM_base``
    ``RMnoPB``    ``mask_LM[t-1] * LM_base``               ``RM_list[t]``
    ``PBwR``      ``LM_base`` (product masked: ``mask[t-1] * (LM_base . RM_base)``)
    none          ``mask_LM[t-1] * LM_base``               ``mask_RM[t-1] * RM_base``
    ============  =======================================  ================================

    As everywhere in this package the correction is folded into the weight operand of the fused
    GEMM (``effective``); no activation-sized tensor is touched."""

    def __init__(self, in_features, out_features, config, bias=True, threshold=None, task=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.mask_scale, self.mask_init = config.mask_scale, config.mask_init
        self.threshold = DEFAULT_THRESHOLD if threshold is None else threshold   # the reference always applies the default
        self.info = {"threshold_fn": config.threshold_fn, "threshold": self.threshold}
        self.threshold_fn = _threshold_fn(config)
        self.weight = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(out_features, in_features)))
        if bias:
            self.bias = nn.Parameter(nn.init.normal_(torch.empty(out_features)))
        else:
            self.register_parameter("bias", None)
        self.scale = config.rkr_scale
        K, T = config.K, config.task_num
        self.RGnoPB, self.RMnoPB, self.LMnoPB, self.PBwR = config.RGnoPB, config.RMnoPB, config.LMnoPB, config.PBwR

        def lm():
            return nn.Parameter(nn.init.kaiming_uniform_(torch.empty(in_features, K)) * self.scale)

        def rm():
            return nn.Parameter(nn.init.kaiming_uniform_(torch.empty(K, out_features)) * self.scale)

        if self.RGnoPB:
            self.LM_list = nn.ParameterList([lm() for _ in range(T)])
            self.RM_list = nn.ParameterList([rm() for _ in range(T)])
        elif self.LMnoPB:
            self.LM_list = nn.ParameterList([lm() for _ in range(T)])
            self.RM_base = rm()
            self.mask_reals_RM = _mask_list(self.RM_base, config, T - 1)
        elif self.RMnoPB:
            self.RM_list = nn.ParameterList([rm() for _ in range(T)])
            self.LM_base = lm()
            self.mask_reals_LM = _mask_list(self.LM_base, config, T - 1)
        else:
            self.LM_base, self.RM_base = lm(), rm()
            if self.PBwR:
                self.mask_reals = _mask_list(torch.empty(in_features, out_features), config, T - 1)
            else:
                self.mask_reals_LM = _mask_list(self.LM_base, config, T - 1)
                self.mask_reals_RM = _mask_list(self.RM_base, config, T - 1)

    def factors(self, task):
        """(LM_t [in, K], RM_t [K, out], real-valued mask of their product or None) (reference forward, :253-297)."""
        if self.RGnoPB:
            return self.LM_list[task], self.RM_list[task], None
        if self.LMnoPB:
            rm = self.RM_base if task == 0 else _masked(self, self.mask_reals_RM[task - 1], self.RM_base)
            return self.LM_list[task], rm, None
        if self.RMnoPB:
            lm = self.LM_base if task == 0 else _masked(self, self.mask_reals_LM[task - 1], self.LM_base)
            return lm, self.RM_list[task], None
        if task == 0:
            return self.LM_base, self.RM_base, None
        if self.PBwR:
            return self.LM_base, self.RM_base, self.mask_reals[task - 1]
        return (_masked(self, self.mask_reals_LM[task - 1], self.LM_base),
                _masked(self, self.mask_reals_RM[task - 1], self.RM_base), None)

    def rank_term(self, task):
        """R_t [in, out]."""
        lm, rm, mask_real = self.factors(task)
        R = ops.matmul_f32(lm, rm) if lm.is_cuda else torch.matmul(lm, rm)
        return R if mask_real is None else _masked(self, mask_real, R)

    def effective(self, task, sfg=None):
        """(W + R_t^T [out, in], bias), with the output-feature scale of ``sfg`` folded in when given."""
        lm, rm, mask = self.factors(task)
        if mask is None and self.bias is not None and self.weight.is_cuda and self.in_features % 4 == 0 and lm.shape[1] <= 8:
            # one fused launch (two in backward) for the fold and its gradients
            return ops.rkr_fold(self.weight, self.bias, lm, rm, sfg.scale(task) if sfg is not None else None)
        R = ops.matmul_f32(lm, rm) if lm.is_cuda else torch.matmul(lm, rm)
        w, b = self.weight + (R if mask is None else _masked(self, mask, R)).t(), self.bias
        if sfg is not None:
            f = sfg.scale(task)
            w = w * f.unsqueeze(1)
            b = b * f if b is not None else None
        return w, b

    def forward(self, input, task):
        from . import modeling as base
        w, b = self.effective(task)
        shape = input.shape
        if b is None:
            b = torch.zeros(self.out_features, device=w.device)
        y = ops.linear(base._to_compute(input).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.out_features)

    __repr__ = ElementWiseLinear.__repr__


class ElementWiseSFG(nn.Module):
    """Per-task scale of every output feature (reference :306-368): a plain per-task list under
    ``SFGnoPB``, otherwise a shared vector, used as it is by task 0 and masked for task t > 0."""

    def __init__(self, out_features, config, threshold=None, task=None):
        super().__init__()
        self.out_features = out_features
        self.mask_scale, self.mask_init = config.mask_scale, config.mask_init
        self.threshold = DEFAULT_THRESHOLD if threshold is None else threshold
        self.info = {"threshold_fn": config.threshold_fn, "threshold": self.threshold}
        self.threshold_fn = _threshold_fn(config)
        self.SFGnoPB = config.SFGnoPB
        self.F_base = nn.Parameter(torch.ones(out_features))
        if self.SFGnoPB:
            self.F_list = nn.ParameterList([nn.Parameter(torch.ones(out_features)) for _ in range(config.task_num)])
        else:
            self.mask_reals_F = _mask_list(self.F_base, config, config.task_num - 1)

    def scale(self, task):
        if self.SFGnoPB:
            return self.F_list[task]
        if task == 0:
            return self.F_base
        return _masked(self, self.mask_reals_F[task - 1], self.F_base)

    def forward(self, x, task):
        return x * self.scale(task).to(x.dtype)
