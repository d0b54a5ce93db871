"""ViT with RKR factors shared across tasks and specialised by piggyback masks, on the vitk HIP
kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/modeling_RKRPB.py``: ``Attention`` (:49-113), ``Mlp``
(:116-150), ``Block`` (:200-256), ``Encoder`` (:259-278), ``Transformer`` (:281-290) and
``VisionTransformer`` (:293-386) keep their constructor signatures (each takes the extra ``task``
argument of the reference and, like it, ignores it beyond passing it down), ``forward(x, task)``,
attribute names and ``state_dict`` keys.  It is the RKR file with ``ElementWiseRG`` /
``ElementWiseSFG`` in place of ``RG_FC`` / ``SFG_FC``: only the way the per-task weight operand is
formed differs (``effective``), everything after that is the shared fused path.

One behaviour of the reference is kept on purpose: its attention scales query, key and value with
``SFG_query/key/value`` only AFTER the heads have been split off the unscaled projections
(:90-97), so those three scales never reach the output and their parameters receive no gradient;
only ``SFG_out`` acts.
"""
import torch
import torch.nn as nn

from . import modeling_RKR as rkr
from .layers_PB import ElementWiseRG, ElementWiseSFG
from .modeling import (ATTENTION_K, ATTENTION_NORM, ATTENTION_OUT, ATTENTION_Q, ATTENTION_V, FC_0, FC_1,  # noqa: F401
                       MLP_NORM, Embeddings, np2th)

LN_EPS = rkr.LN_EPS


class Attention(rkr.Attention):
    def __init__(self, config, vis, task=None):
        nn.Module.__init__(self)
        self.vis = vis
        self.num_attention_heads = config.transformer["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.SFG = config.SFG
        rate = config.transformer["attention_dropout_rate"]
        self.attn_dropout = nn.Dropout(rate)
        self.proj_dropout = nn.Dropout(rate)
        self.softmax = nn.Softmax(dim=-1)
        self.query = ElementWiseRG(config.hidden_size, self.all_head_size, config, task=task)
        self.key = ElementWiseRG(config.hidden_size, self.all_head_size, config, task=task)
        self.value = ElementWiseRG(config.hidden_size, self.all_head_size, config, task=task)
        self.out = ElementWiseRG(config.hidden_size, config.hidden_size, config, task=task)
        if self.SFG:
            self.SFG_query = ElementWiseSFG(self.all_head_size, config)
            self.SFG_key = ElementWiseSFG(self.all_head_size, config)
            self.SFG_value = ElementWiseSFG(self.all_head_size, config)
            self.SFG_out = ElementWiseSFG(self.all_head_size, config)

    def _sfgs(self):
        return None, None, None, (self.SFG_out if self.SFG else None)


class Mlp(rkr.Mlp):
    def __init__(self, config, task=None):
        nn.Module.__init__(self)
        self.SFG = config.SFG
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self.fc1 = ElementWiseRG(config.hidden_size, config.transformer["mlp_dim"], config, task=task)
        self.fc2 = ElementWiseRG(config.transformer["mlp_dim"], config.hidden_size, config, task=task)
        if self.SFG:
            self.SFG1 = ElementWiseSFG(config.transformer["mlp_dim"], config)
            self.SFG2 = ElementWiseSFG(config.hidden_size, config)
        self._init_weights()


class Block(rkr.Block):
    def __init__(self, config, vis, task=None):
        nn.Module.__init__(self)
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config, task)
        self.attn = Attention(config, vis, task)


class Encoder(rkr.Encoder):
    def __init__(self, config, vis, task=None):
        nn.Module.__init__(self)
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(Block(config, vis, task))


class Transformer(rkr.Transformer):
    def __init__(self, config, img_size, vis, task=None):
        nn.Module.__init__(self)
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis, task)


class VisionTransformer(rkr.VisionTransformer):
    """Multi-task classifier (reference :293-325): one head per task when ``config.multi_head``."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False, task=None):
        nn.Module.__init__(self)
        self.num_classes = num_classes
        self.num_tasks = len(num_classes)
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.multi_head = config.multi_head
        self.transformer = Transformer(config, img_size, vis, task)
        if self.multi_head:
            self.head = nn.ModuleList([nn.Linear(config.hidden_size, n) for n in num_classes])
        else:
            self.head = nn.Linear(config.hidden_size, num_classes)
