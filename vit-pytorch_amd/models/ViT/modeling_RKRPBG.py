"""ViT with a rank-K weight correction whose filters come partly from a bank of earlier tasks' filters, on
the vitk HIP kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/modeling_RKRPBG.py``: ``RG_FC`` (:40-138), ``SFG_FC`` (:141-152),
``Attention`` (:167-235), ``Mlp`` (:238-276), ``Block`` (:325-389), ``Encoder``, ``Transformer`` and
``VisionTransformer`` (:411-500) keep their constructor signatures (the model is BUILT for one ``task``),
``forward(x, task)``, attribute names and ``state_dict`` keys (``weight``, ``bias``, ``unc_filt_LM``,
``unc_filt_RM``, ``weights_mat_LM``, ``weights_mat_RM``, the buffer ``concat_unc_filter``, ``SFG_F``).

The correction R that is added to the layer's weight is, for task 0 or ``lamb == 1``, the rank-K product
``unc_filt_LM . unc_filt_RM`` ``[out, in]``; otherwise its first ``ceil(lamb * out)`` rows are that product
and the rest are mixtures of the bank with the rank-K mixing matrix ``weights_mat_LM . weights_mat_RM``
(the reshapes of modeling_PBG.py).  Unlike modeling_RKRPB.py, this file scales query, key and value BEFORE
the heads are split off (:211-214), so all four attention scales act.  The weight-shaped arithmetic forms
the operand of the fused operators (``effective``).
"""
import math

import torch
import torch.nn as nn

from vitk import ops

from . import modeling as base
from . import modeling_RKR as rkr
from .modeling import (ATTENTION_K, ATTENTION_NORM, ATTENTION_OUT, ATTENTION_Q, ATTENTION_V, FC_0, FC_1,  # noqa: F401
                       MLP_NORM, Embeddings, np2th)
from .modeling_PBG import bank_filters

LN_EPS = rkr.LN_EPS
TASK_NUM = 10


class RG_FC(nn.Module):
    def __init__(self, RG, K, task_num, h_in, h_out, lambdas, scale, bias=True, task=0):
        super().__init__()
        self.RG, self.task, self.h_in, self.h_out = RG, task, h_in, h_out
        self.weight = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(h_out, h_in)))
        if bias:
            self.bias = nn.Parameter(nn.init.normal_(torch.empty(h_out)))
        else:
            self.register_parameter("bias", None)
        if not RG:
            return
        self.out_filt = h_out
        self.lambdas = lambdas
        self.lamb_num = math.ceil(lambdas * h_out)
        self.lamb_rem_num = h_out - self.lamb_num
        self.scale = scale
        self.full = task == 0 or lambdas == 1.0
        self.unc_filt_RM = None
        if self.full:
            self.unc_filt_LM = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(h_out, K)) * scale)
            self.unc_filt_RM = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(K, h_in)) * scale)
            self.register_parameter("weights_mat_LM", None)
            self.register_parameter("weights_mat_RM", None)
        else:
            rows = h_out + (task - 1) * self.lamb_num
            self.unc_filt_LM = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(self.lamb_num, K)) * scale)
            self.unc_filt_RM = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(K, h_in)) * scale)
            self.weights_mat_LM = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(rows, K)))
            self.weights_mat_RM = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(K, self.lamb_rem_num)))
            self.register_buffer("concat_unc_filter", torch.empty(rows, h_in))

    def effective(self, task=None, sfg=None):
        """(W + R [out, in], bias) of the task the layer was built for, with the output-feature scale of
        ``sfg`` folded in when given."""
        w, b = self.weight, self.bias
        if self.RG:
            if self.full and w.is_cuda and b is not None and self.h_in % 4 == 0 and self.unc_filt_LM.shape[1] <= 8:
                # W + LM . RM is the fold of modeling_RKR.py with the factors in the other orientation
                return ops.rkr_fold(w, b, self.unc_filt_RM.t(), self.unc_filt_LM.t(),
                                    sfg.scale(task) if sfg is not None else None)
            mm = ops.matmul_f32 if w.is_cuda else torch.matmul
            R = mm(self.unc_filt_LM, self.unc_filt_RM)
            if not self.full:
                mat = mm(self.weights_mat_LM, self.weights_mat_RM)
                R = torch.cat((R, bank_filters(self.concat_unc_filter, mat, self.h_in)), dim=0)
            w = w + R
        if sfg is not None:
            f = sfg.scale(task)
            w = w * f.unsqueeze(1)
            b = b * f if b is not None else None
        return w, b

    def forward(self, x, task=None):
        w, b = self.effective()
        shape = x.shape
        if b is None:
            b = torch.zeros(self.h_out, device=w.device)
        y = ops.linear(base._to_compute(x).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.h_out)


class SFG_FC(nn.Module):
    """One scale per output feature, the same for every task (reference :141-152)."""

    def __init__(self, c_out, task_num: int):
        super().__init__()
        self.SFG_F = nn.Parameter(torch.ones(c_out))

    def scale(self, task=None):
        return self.SFG_F

    def forward(self, x, task=None):
        return x * self.SFG_F.to(x.dtype)


def _fc(config, h_in, h_out, task):
    return RG_FC(config.RG, config.K, config.task_num, h_in, h_out, config.lamb, config.rkr_scale, task=task)


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
        self.RG, self.SFG = config.RG, config.SFG
        self.query = _fc(config, config.hidden_size, self.all_head_size, task)
        self.key = _fc(config, config.hidden_size, self.all_head_size, task)
        self.value = _fc(config, config.hidden_size, self.all_head_size, task)
        self.out = _fc(config, config.hidden_size, config.hidden_size, task)
        if self.SFG:
            self.SFG_query = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_key = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_value = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_out = SFG_FC(self.all_head_size, config.task_num)


class Mlp(rkr.Mlp):
    def __init__(self, config, task):
        nn.Module.__init__(self)
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self.RG, self.SFG = config.RG, config.SFG
        self.fc1 = _fc(config, config.hidden_size, config.transformer["mlp_dim"], task)
        self.fc2 = _fc(config, config.transformer["mlp_dim"], config.hidden_size, task)
        self._init_weights()
        if self.SFG:
            self.SFG1 = SFG_FC(config.transformer["mlp_dim"], config.task_num)
            self.SFG2 = SFG_FC(config.hidden_size, config.task_num)


class Block(rkr.Block):
    def __init__(self, config, vis, task):
        nn.Module.__init__(self)
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config, task)
        self.attn = Attention(config, vis, task)


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
    """Classifier built for one task (reference :411-428): ``num_classes`` is the list over tasks, the single
    head has ``num_classes[task]`` outputs."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False, task=0):
        nn.Module.__init__(self)
        self.num_classes = num_classes
        self.num_tasks = len(num_classes)
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.multi_head = False                       # the reference keeps config.multi_head but builds one head
        self.transformer = Transformer(config, img_size, vis, task)
        self.head = nn.Linear(config.hidden_size, num_classes[task])
