"""ViT with piggyback-masked linears on the vitk HIP kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/ViT/modeling_PB.py``: ``Attention`` (:58-105), ``Mlp`` (:108-133),
``Block`` (:180-199), ``Encoder`` (:238-256), ``Transformer`` (:259-268) and ``VisionTransformer``
(:271-365) keep their constructor / ``forward(x, task)`` signatures, attribute names and
``state_dict`` keys.  It is the RKR file with ``ElementWiseLinear`` in place of ``RG_FC`` and no SFG
scales, and is built the same way here: the layers only differ in how the per-task weight operand
is formed (``effective``), everything after that is the shared fused path.
"""
import torch.nn as nn

from . import modeling_RKR as rkr
from .layers_PB import ElementWiseLinear
from .modeling import (ATTENTION_K, ATTENTION_NORM, ATTENTION_OUT, ATTENTION_Q, ATTENTION_V, FC_0, FC_1,  # noqa: F401
                       MLP_NORM, Embeddings, np2th)

LN_EPS = rkr.LN_EPS


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
        self.RG = self.SFG = False
        self.query = ElementWiseLinear(config.hidden_size, self.all_head_size, config)
        self.key = ElementWiseLinear(config.hidden_size, self.all_head_size, config)
        self.value = ElementWiseLinear(config.hidden_size, self.all_head_size, config)
        self.out = ElementWiseLinear(config.hidden_size, config.hidden_size, config)


class Mlp(rkr.Mlp):
    def __init__(self, config):
        nn.Module.__init__(self)
        import torch
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self.RG = self.SFG = False
        self.fc1 = ElementWiseLinear(config.hidden_size, config.transformer["mlp_dim"], config)
        self.fc2 = ElementWiseLinear(config.transformer["mlp_dim"], config.hidden_size, config)
        self._init_weights()


class Block(rkr.Block):
    def __init__(self, config, vis):
        nn.Module.__init__(self)
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis)


class Encoder(rkr.Encoder):
    def __init__(self, config, vis):
        nn.Module.__init__(self)
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(Block(config, vis))


class Transformer(rkr.Transformer):
    def __init__(self, config, img_size, vis):
        nn.Module.__init__(self)
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)


class VisionTransformer(rkr.VisionTransformer):
    """Multi-task classifier (reference :271-303): one head per task when ``config.multi_head``."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        nn.Module.__init__(self)
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
