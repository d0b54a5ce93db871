"""ViT whose linears draw most of their filters from a bank of earlier tasks' filters, on the vitk HIP
kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/modeling_PBG.py``: ``PiggybackFC`` (:36-74), ``Attention``
(:92-142), ``Mlp`` (:145-173), ``Block`` (:222-286), ``Encoder`` (:289-307), ``Transformer`` (:310-319) and
``VisionTransformer`` (:322-398) keep their constructor signatures (the model is BUILT for one ``task``),
``forward(x, task)``, attribute names and ``state_dict`` keys (``unc_filt``, ``weights_mat``, ``bias`` and the
buffer ``concat_unc_filter``).  One classifier head, as in the reference.

For task 0 a layer is a plain linear on ``unc_filt``.  For a later task it owns ``ceil(lamb * out)``
unconstrained filters; the remaining output filters are mixtures of the bank: the bank ``[C, in]`` is
RESHAPED (not transposed: the reference reinterprets the row-major memory) to ``[in, C]``, multiplied by
``weights_mat [C, out - lamb_num]`` and reshaped to ``[out - lamb_num, in]``.  As everywhere in this package
that weight-shaped arithmetic forms the operand of the fused GEMM / attention / MLP operators
(``effective``); no activation-sized tensor is touched.  The mixing product is a library GEMM on
weight-sized fp32 matrices.
"""
import math
from os.path import join as pjoin

import torch
import torch.nn as nn

from vitk import ops

from . import modeling as base
from . import modeling_RKR as rkr
from .modeling import (ATTENTION_K, ATTENTION_NORM, ATTENTION_OUT, ATTENTION_Q, ATTENTION_V, FC_0, FC_1,  # noqa: F401
                       MLP_NORM, Embeddings, np2th)

LN_EPS = rkr.LN_EPS


def bank_filters(bank, weights_mat, h_in):
    """Filters mixed out of the bank (reference :67-69): reshape ``[C, in] -> [in, C]``, times
    ``weights_mat [C, n]``, reshape ``[in, n] -> [n, in]``."""
    mm = ops.matmul_f32 if bank.is_cuda else torch.matmul          # (CPU: construction-time checks only)
    mixed = mm(torch.reshape(bank, (h_in, bank.shape[0])), weights_mat)
    return torch.reshape(mixed, (mixed.shape[1], h_in))


class PiggybackFC(nn.Module):
    def __init__(self, in_channels, out_channels, config, bias=True, task=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))        # from a checkpoint, as in the reference
        else:
            self.register_parameter("bias", None)
        self.task = task
        self.lambdas = config.lamb
        self.lamb_num = math.ceil(self.lambdas * out_channels)
        self.lamb_rem_num = out_channels - self.lamb_num
        if task == 0:
            self.unc_filt = nn.Parameter(nn.init.normal_(torch.empty(out_channels, in_channels), 0.0, 0.02))
        else:
            rows = out_channels + (task - 1) * self.lamb_num
            self.unc_filt = nn.Parameter(nn.init.normal_(torch.empty(self.lamb_num, in_channels), 0.0, 0.02))
            self.weights_mat = nn.Parameter(nn.init.normal_(torch.empty(rows, self.lamb_rem_num), 0.0, 0.02))
            self.register_buffer("concat_unc_filter", torch.empty(rows, in_channels))

    def effective(self, task=None, sfg=None):
        """(weight [out, in], bias) of the task the layer was built for (the ``task`` argument is ignored, as
        the reference's forward ignores it)."""
        if sfg is not None:
            raise ValueError("PiggybackFC takes no SFG scale")
        if self.task == 0:
            return self.unc_filt, self.bias
        w = torch.cat((self.unc_filt, bank_filters(self.concat_unc_filter, self.weights_mat, self.in_channels)), dim=0)
        return w, self.bias

    def forward(self, x, task=None):
        w, b = self.effective()
        shape = x.shape
        if b is None:
            b = torch.zeros(self.out_channels, device=w.device)
        y = ops.linear(base._to_compute(x).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.out_channels)


class Attention(rkr.Attention):
    def __init__(self, config, vis, task):
        nn.Module.__init__(self)
        self.vis = vis
        self.num_attention_heads = config.transformer["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        rate = config.transformer["attention_dropout_rate"]
        self.attn_dropout = nn.Dropout(rate)
        self.proj_dropout = nn.Dropout(rate)
        self.softmax = nn.Softmax(dim=-1)
        self.RG = self.SFG = False
        self.query = PiggybackFC(config.hidden_size, self.all_head_size, config, task=task)
        self.key = PiggybackFC(config.hidden_size, self.all_head_size, config, task=task)
        self.value = PiggybackFC(config.hidden_size, self.all_head_size, config, task=task)
        self.out = PiggybackFC(config.hidden_size, config.hidden_size, config, task=task)


class Mlp(rkr.Mlp):
    def __init__(self, config, task):
        nn.Module.__init__(self)
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self.RG = self.SFG = False
        self.fc1 = PiggybackFC(config.hidden_size, config.transformer["mlp_dim"], config, task=task)
        self.fc2 = PiggybackFC(config.transformer["mlp_dim"], config.hidden_size, config, task=task)
        self._init_weights()

    def _init_weights(self):
        nn.init.xavier_uniform_(self.fc1.unc_filt)
        nn.init.xavier_uniform_(self.fc2.unc_filt)


class Block(rkr.Block):
    def __init__(self, config, vis, task):
        nn.Module.__init__(self)
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config, task)
        self.attn = Attention(config, vis, task)

    def load_from(self, weights, n_block):
        """One encoder block out of a JAX ``.npz`` checkpoint into the task-0 layout (reference :242-286)."""
        root = f"Transformer/encoderblock_{n_block}"
        hs = self.hidden_size

        def arr(sub, leaf):
            return np2th(weights[pjoin(root, sub, leaf)])

        with torch.no_grad():
            for sub, lin in ((ATTENTION_Q, self.attn.query), (ATTENTION_K, self.attn.key), (ATTENTION_V, self.attn.value),
                             (ATTENTION_OUT, self.attn.out)):
                lin.unc_filt.copy_(arr(sub, "kernel").view(hs, hs).t())
                lin.bias.copy_(arr(sub, "bias").view(-1))
            for sub, lin in ((FC_0, self.ffn.fc1), (FC_1, self.ffn.fc2)):
                lin.unc_filt.copy_(arr(sub, "kernel").t())
                lin.bias.copy_(arr(sub, "bias").t())
            for sub, ln in ((ATTENTION_NORM, self.attention_norm), (MLP_NORM, self.ffn_norm)):
                ln.weight.copy_(arr(sub, "scale"))
                ln.bias.copy_(arr(sub, "bias"))


class Encoder(rkr.Encoder):
    def __init__(self, config, vis, task):
        nn.Module.__init__(self)
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(Block(config, vis, task))


class Transformer(rkr.Transformer):
    def __init__(self, config, img_size, vis, task):
        nn.Module.__init__(self)
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis, task)


class VisionTransformer(rkr.VisionTransformer):
    """Classifier built for one task (reference :322-349): a single head of ``num_classes`` outputs."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False, task=0):
        nn.Module.__init__(self)
        self.num_classes = num_classes
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.multi_head = False                       # the reference keeps config.multi_head but builds one head
        self.transformer = Transformer(config, img_size, vis, task)
        self.head = nn.Linear(config.hidden_size, num_classes)
