"""ViT with RKR task-conditioned linears on the vitk HIP kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/modeling_RKR.py``: ``RG_FC`` (:35-65), ``SFG_FC`` (:67-78),
``Attention`` (:99-160), ``Mlp`` (:163-200), ``Block`` (:247-266), ``Encoder`` (:306-323),
``Transformer`` (:326-335) and ``VisionTransformer`` (:338-432) keep their constructor / ``forward``
signatures (``forward(x, task)``), attribute names and ``state_dict`` keys (``weight``, ``bias``,
``LM_list.<t>``, ``RM_list.<t>``, ``F_list.<t>``, ``head.<t>``), which the training driver filters by
name (reference train.py:400-468).

How the task conditioning reaches the kernels.  The reference evaluates, per token,

    y = (x . (W + (LM_t . RM_t)^T)^T + b) * F_t                      (RG_FC.forward, SFG_FC.forward)

Both corrections act on the weight side only, so they are folded into the operand the GEMM
already re-reads every step:

    W_t = diag(F_t) . (W + RM_t^T . LM_t^T),      b_t = F_t * b

``W_t`` / ``b_t`` are built once per forward on the [out, in] weight (2.4 M elements for 768x3072:
microseconds) and handed to the same fused GEMM / attention / MLP operators as the plain model; not
one extra pass over the [tokens, features] activations exists, and nothing task-specific sits in
a GEMM epilogue.  Gradients of ``W``, ``LM_t``, ``RM_t``, ``F_t`` and ``b`` follow from dW_t / db_t
(produced by the weight-gradient kernel) through autograd on those small weight-shaped tensors.
"""
import torch
import torch.nn as nn

import vitk
from vitk import ops

from . import modeling as base
from .modeling import (ATTENTION_K, ATTENTION_NORM, ATTENTION_OUT, ATTENTION_Q, ATTENTION_V, FC_0, FC_1,  # noqa: F401
                       MLP_NORM, Embeddings, np2th)

LN_EPS = base.LN_EPS


class RG_FC(nn.Module):
    """Linear layer whose weight gets a per-task rank-K correction (reference :35-65)."""

    def __init__(self, config, h_in, h_out):
        super().__init__()
        self.RG = config.RG
        K = config.K
        task_num = config.task_num
        self.h_in, self.h_out = h_in, h_out
        self.weight = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(h_out, h_in)))
        self.bias = nn.Parameter(nn.init.normal_(torch.empty(h_out)))
        if self.RG:
            self.scale = config.rkr_scale
            self.LM_list = nn.ParameterList(
                [nn.Parameter(nn.init.kaiming_uniform_(torch.empty(h_in, K)) * self.scale) for _ in range(task_num)])
            self.RM_list = nn.ParameterList(
                [nn.Parameter(nn.init.kaiming_uniform_(torch.empty(K, h_out)) * self.scale) for _ in range(task_num)])

    def effective(self, task, sfg=None):
        """(W_t [out, in], b_t [out]) with the rank-K term and, when given, the SFG_FC column scale
        folded in."""
        w, b = self.weight, self.bias
        if self.RG and w.is_cuda and self.h_in % 4 == 0 and self.LM_list[task].shape[1] <= 8:
            # one fused launch (and two in backward) instead of a dozen weight-shaped tensor ops
            return ops.rkr_fold(w, b, self.LM_list[task], self.RM_list[task], sfg.scale(task) if sfg is not None else None)
        if self.RG:
            w = torch.addmm(w, self.RM_list[task].t(), self.LM_list[task].t())
        if sfg is not None:
            f = sfg.scale(task)
            w = w * f.unsqueeze(1)
            b = b * f
        return w, b

    def forward(self, x, task: int):
        w, b = self.effective(task)
        shape = x.shape
        y = ops.linear(base._to_compute(x).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.h_out)


class SFG_FC(nn.Module):
    """Per-task scale of every output feature (reference :67-78)."""

    def __init__(self, c_out, task_num: int):
        super().__init__()
        self.F_list = nn.ParameterList([nn.Parameter(torch.ones(c_out)) for _ in range(task_num)])

    def scale(self, task: int):
        return self.F_list[task]

    def forward(self, x, task: int):
        return x * self.F_list[task].to(x.dtype)


class Attention(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.num_attention_heads = config.transformer["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        rate = config.transformer["attention_dropout_rate"]
        self.attn_dropout = nn.Dropout(rate)
        self.proj_dropout = nn.Dropout(rate)
        self.softmax = nn.Softmax(dim=-1)
        self.RG, self.SFG = config.RG, config.SFG
        self.query = RG_FC(config, config.hidden_size, self.all_head_size)
        self.key = RG_FC(config, config.hidden_size, self.all_head_size)
        self.value = RG_FC(config, config.hidden_size, self.all_head_size)
        self.out = RG_FC(config, config.hidden_size, config.hidden_size)
        if self.SFG:
            self.SFG_query = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_key = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_value = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_out = SFG_FC(self.all_head_size, config.task_num)

    def _sfgs(self):
        """The SFG modules that act on query, key, value and output projection."""
        if not self.SFG:
            return None, None, None, None
        return self.SFG_query, self.SFG_key, self.SFG_value, self.SFG_out

    def forward(self, hidden_states, task, residual=None):
        """hidden_states [B,N,D] -> (attention_output, probs or None); ``residual`` (not in the
        reference signature) is added in the output projection's epilogue."""
        x = base._to_compute(hidden_states)
        B, N, D = x.shape
        H = self.num_attention_heads
        sq, sk, sv, so = self._sfgs()
        wq, bq = self.query.effective(task, sq)
        wk, bk = self.key.effective(task, sk)
        wv, bv = self.value.effective(task, sv)
        wo, bo = self.out.effective(task, so)
        qkv = ops.linear(x.reshape(B * N, D), (wq, wk, wv), (bq, bk, bv))
        ctx, lse = ops.attention(qkv, B, N, H, dropout_p=self.attn_dropout.p if self.training else 0.0)
        weights = ops.attention_probs(qkv.detach(), lse, B, N, H) if self.vis else None
        p = self.proj_dropout.p if self.training else 0.0
        out = ops.linear(ctx, (wo,), (bo,), residual=residual, dropout_p=p)
        return out.view(B, N, D), weights


class Mlp(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self.RG, self.SFG = config.RG, config.SFG
        self.fc1 = RG_FC(config, config.hidden_size, config.transformer["mlp_dim"])
        self.fc2 = RG_FC(config, config.transformer["mlp_dim"], config.hidden_size)
        self._init_weights()
        if self.SFG:
            self.SFG1 = SFG_FC(config.transformer["mlp_dim"], config.task_num)
            self.SFG2 = SFG_FC(config.hidden_size, config.task_num)

    def _init_weights(self):
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)
        nn.init.normal_(self.fc1.bias, std=1e-6)
        nn.init.normal_(self.fc2.bias, std=1e-6)

    def forward(self, x, task, residual=None):
        w1, b1 = self.fc1.effective(task, self.SFG1 if self.SFG else None)
        w2, b2 = self.fc2.effective(task, self.SFG2 if self.SFG else None)
        p = self.dropout.p if self.training else 0.0
        return ops.mlp(base._to_compute(x), w1, b1, w2, b2, residual=residual, dropout_p=p)


class Block(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis)

    def forward(self, x, task):
        cdt = vitk.get_compute_dtype()
        x = base._to_stream(x)
        h, y = ops.add_norm(x, self.attention_norm.weight, self.attention_norm.bias, self.attention_norm.eps, cdt)
        x, weights = self.attn(y, task, residual=h)
        h, y = ops.add_norm(x, self.ffn_norm.weight, self.ffn_norm.bias, self.ffn_norm.eps, cdt)
        x = self.ffn(y, task, residual=h)
        return x, weights

    load_from = base.Block.load_from      # same checkpoint keys, same attribute names (reference :268-303)


class Encoder(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(Block(config, vis))

    def forward(self, hidden_states, task):
        attn_weights = []
        for layer_block in self.layer:
            hidden_states, weights = layer_block(hidden_states, task)
            if self.vis:
                attn_weights.append(weights)
        encoded = ops.layer_norm(base._to_stream(hidden_states), self.encoder_norm.weight, self.encoder_norm.bias,
                                 self.encoder_norm.eps, out_dtype=torch.float32)
        return encoded, attn_weights


class Transformer(nn.Module):
    def __init__(self, config, img_size, vis):
        super().__init__()
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)

    def forward(self, input_ids, task):
        return self.encoder(self.embeddings(input_ids), task)


class VisionTransformer(nn.Module):
    """Multi-task classifier (reference :338-370): ``num_classes`` is a list, one head per task when
    ``config.multi_head``; heads and the loss stay on stock PyTorch ops."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        super().__init__()
        self.num_classes = num_classes
        self.num_tasks = len(num_classes)
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.multi_head = config.multi_head
        self.transformer = Transformer(config, img_size, vis)
        if self.multi_head:
            self.head = nn.ModuleList([nn.Linear(config.hidden_size, n) for n in num_classes])
        else:
            self.head = nn.Linear(config.hidden_size, num_classes)

    def forward(self, x, labels=None, task=None):
        x, attn_weights = self.transformer(x, task)
        logits = self.head[task](x[:, 0]) if self.multi_head else self.head(x[:, 0])
        num_class = self.num_classes[task] if isinstance(self.num_classes, list) else self.num_classes
        if labels is None:
            return logits, attn_weights
        return nn.functional.cross_entropy(logits.view(-1, num_class), labels.view(-1))

    _fit_posemb = base.VisionTransformer._fit_posemb

    def load_from(self, weights):
        import numpy as np
        emb, enc = self.transformer.embeddings, self.transformer.encoder
        with torch.no_grad():
            if self.zero_head:
                for head in (self.head if self.multi_head else [self.head]):
                    nn.init.zeros_(head.weight)
                    nn.init.zeros_(head.bias)
            else:
                self.head.weight.copy_(np2th(weights["head/kernel"]).t())
                self.head.bias.copy_(np2th(weights["head/bias"]).t())
            emb.patch_embeddings.weight.copy_(np2th(weights["embedding/kernel"], conv=True))
            emb.patch_embeddings.bias.copy_(np2th(weights["embedding/bias"]))
            emb.cls_token.copy_(np2th(weights["cls"]))
            enc.encoder_norm.weight.copy_(np2th(weights["Transformer/encoder_norm/scale"]))
            enc.encoder_norm.bias.copy_(np2th(weights["Transformer/encoder_norm/bias"]))
            emb.position_embeddings.copy_(
                self._fit_posemb(np.asarray(weights["Transformer/posembed_input/pos_embedding"]),
                                 emb.position_embeddings.shape[1]))
            for i, block in enumerate(enc.layer):
                block.load_from(weights, n_block=str(i))
