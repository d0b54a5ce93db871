"""ViT whose RKR factors come from a task-embedding hyper-network, on the vitk HIP kernels
(MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/modeling_RKRTSN.py``: ``TaskEmbedding`` (:39-57), ``RG_FC``
(:59-85), ``SFG_FC`` (:87-98), ``Attention`` (:114-179), ``Mlp`` (:182-217), ``Block`` (:267-286),
``Encoder`` (:326-343), ``Transformer`` (:346-355) and ``VisionTransformer`` (:358-405) keep their
constructor signatures, ``forward(..., task, task_emb_vec)`` / ``forward(x, labels, task, task_vecs)``,
attribute names and ``state_dict`` keys (``LM_fc``, ``RM_fc``, ``F_list.<t>``, ``task_embedding.fc.<i>``,
``head.<t>``).

A stack of eight small linear layers maps every task's descriptor vector to an embedding; two more
linear layers per ``RG_FC`` turn the current task's embedding into the rank-K factors
``LM [in, K]`` and ``RM [K, out]``.  From there it is the RKR model: the correction is folded into
the weight operand of the fused GEMM (``effective``), and the gradients of the hyper-network follow
from the weight gradient through autograd on those small tensors.  The classes below are the RKR
ones; the pair ``(task, task_emb_vec)`` travels through their ``forward`` in place of the task index.
"""
import torch
import torch.nn as nn

from . import modeling_RKR as rkr
from .modeling import (ATTENTION_K, ATTENTION_NORM, ATTENTION_OUT, ATTENTION_Q, ATTENTION_V, FC_0, FC_1,  # noqa: F401
                       MLP_NORM, Embeddings, np2th)

LN_EPS = rkr.LN_EPS
TASK_NUM = 10


def _join(task, task_emb_vec):
    return task if isinstance(task, tuple) else (task, task_emb_vec)


class TaskEmbedding(nn.Module):
    """Task descriptor [tasks, d] -> task embedding [tasks, task_num] (reference :39-57); stock
    PyTorch operators: a handful of d x d products per step."""

    def __init__(self, d, task_num):
        super().__init__()
        self.fc = nn.Sequential(*([nn.Linear(d, d) for _ in range(7)] + [nn.Linear(d, task_num)]))

    def forward(self, task_vec):
        return self.fc(task_vec)


class RG_FC(nn.Module):
    """Linear layer whose rank-K weight correction is generated from the task embedding (reference :59-85)."""

    def __init__(self, config, h_in, h_out):
        super().__init__()
        self.RG = config.RG
        self.K = config.K
        self.h_in, self.h_out = h_in, h_out
        self.weight = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(h_out, h_in)))
        self.bias = nn.Parameter(nn.init.normal_(torch.empty(h_out)))
        if self.RG:
            self.LM_fc = nn.Linear(config.task_num, h_in * self.K)
            self.RM_fc = nn.Linear(config.task_num, h_out * self.K)

    def effective(self, task, sfg=None):
        """(W_t [out, in], b_t [out]) for ``task = (index, embedding)``, the SFG scale folded in when given."""
        _, emb = task
        w, b = self.weight, self.bias
        if self.RG:
            lm = self.LM_fc(emb).reshape(self.h_in, self.K)
            rm = self.RM_fc(emb).reshape(self.K, self.h_out)
            if w.is_cuda and self.h_in % 4 == 0 and self.K <= 8:
                from vitk import ops
                return ops.rkr_fold(w, b, lm, rm, sfg.scale(task) if sfg is not None else None)
            w = torch.addmm(w, rm.t(), lm.t())
        if sfg is not None:
            f = sfg.scale(task)
            w = w * f.unsqueeze(1)
            b = b * f
        return w, b

    def forward(self, x, task, task_emb_vec=None):
        from vitk import ops
        from . import modeling as base
        w, b = self.effective(_join(task, task_emb_vec))
        shape = x.shape
        y = ops.linear(base._to_compute(x).reshape(-1, shape[-1]), (w,), (b,))
        return y.view(*shape[:-1], self.h_out)


class SFG_FC(nn.Module):
    """Per-task scale of every output feature (reference :87-98; the embedding is not used)."""

    def __init__(self, c_out, task_num: int):
        super().__init__()
        self.F_list = nn.ParameterList([nn.Parameter(torch.ones(c_out)) for _ in range(task_num)])

    def scale(self, task):
        return self.F_list[task[0] if isinstance(task, tuple) else task]

    def forward(self, x, task, task_emb_vec=None):
        return x * self.scale(task).to(x.dtype)


class Attention(rkr.Attention):
    def __init__(self, config, vis):
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
        self.query = RG_FC(config, config.hidden_size, self.all_head_size)
        self.key = RG_FC(config, config.hidden_size, self.all_head_size)
        self.value = RG_FC(config, config.hidden_size, self.all_head_size)
        self.out = RG_FC(config, config.hidden_size, config.hidden_size)
        if self.SFG:
            self.SFG_query = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_key = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_value = SFG_FC(self.all_head_size, config.task_num)
            self.SFG_out = SFG_FC(self.all_head_size, config.task_num)

    def forward(self, hidden_states, task, task_emb_vec=None, residual=None):
        return super().forward(hidden_states, _join(task, task_emb_vec), residual=residual)


class Mlp(rkr.Mlp):
    def __init__(self, config):
        nn.Module.__init__(self)
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self.RG, self.SFG = config.RG, config.SFG
        self.fc1 = RG_FC(config, config.hidden_size, config.transformer["mlp_dim"])
        self.fc2 = RG_FC(config, config.transformer["mlp_dim"], config.hidden_size)
        self._init_weights()
        if self.SFG:
            self.SFG1 = SFG_FC(config.transformer["mlp_dim"], config.task_num)
            self.SFG2 = SFG_FC(config.hidden_size, config.task_num)

    def forward(self, x, task, task_emb_vec=None, residual=None):
        return super().forward(x, _join(task, task_emb_vec), residual=residual)


class Block(rkr.Block):
    def __init__(self, config, vis):
        nn.Module.__init__(self)
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis)

    def forward(self, x, task, task_emb_vec=None):
        return super().forward(x, _join(task, task_emb_vec))


class Encoder(rkr.Encoder):
    def __init__(self, config, vis):
        nn.Module.__init__(self)
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(Block(config, vis))

    def forward(self, hidden_states, task, task_emb_vec=None):
        return super().forward(hidden_states, _join(task, task_emb_vec))


class Transformer(rkr.Transformer):
    def __init__(self, config, img_size, vis):
        nn.Module.__init__(self)
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)

    def forward(self, input_ids, task, task_emb_vec=None):
        return super().forward(input_ids, _join(task, task_emb_vec))


class VisionTransformer(rkr.VisionTransformer):
    """Multi-task classifier with the task-embedding network (reference :358-405).  With labels it
    returns ``(loss, loss_distill)``: the distillation term holds the embeddings of the tasks seen so
    far to the values recorded when each was last the current task."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        nn.Module.__init__(self)
        self.num_classes = num_classes
        self.num_tasks = len(num_classes)
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.multi_head = config.multi_head
        self.task_emb_labels = [None] * config.task_num
        self.task_embedding = TaskEmbedding(d=config.task_emb_d, task_num=self.num_tasks)
        self.transformer = Transformer(config, img_size, vis)
        if self.multi_head:
            self.head = nn.ModuleList([nn.Linear(config.hidden_size, n) for n in num_classes])
        else:
            self.head = nn.Linear(config.hidden_size, num_classes)

    def forward(self, x, labels=None, task=None, task_vecs=None):
        task_emb_vecs = self.task_embedding(task_vecs)
        task_emb_vec = task_emb_vecs[task]
        self.task_emb_labels[task] = task_emb_vec.detach()
        x, attn_weights = self.transformer(x, task, task_emb_vec)
        logits = self.head[task](x[:, 0]) if self.multi_head else self.head(x[:, 0])
        num_class = self.num_classes[task] if isinstance(self.num_classes, list) else self.num_classes
        if labels is None:
            return logits, attn_weights
        loss = nn.functional.cross_entropy(logits.view(-1, num_class), labels.view(-1))
        if task != 0:
            loss_distill = nn.functional.l1_loss(task_emb_vecs[:task + 1],
                                                 torch.stack(self.task_emb_labels[:task + 1], dim=0))
        else:
            loss_distill = torch.tensor(0)
        return loss, loss_distill
