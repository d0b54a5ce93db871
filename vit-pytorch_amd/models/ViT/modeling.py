"""ViT modules backed by the vitk HIP kernels (MI355X / gfx950).

Drop-in for the five hot-path classes of the reference (``/root/reference/models/ViT/modeling.py``):
``Embeddings`` (:122-166), ``Attention`` (:50-94), ``Mlp`` (:97-119), ``Block`` (:169-225) and
``Encoder`` (:228-245) keep their constructor / ``forward`` signatures, attribute names and
``state_dict`` keys; ``Transformer`` / ``VisionTransformer`` (:248-334) are provided so the
training drivers can build the whole model from this package.

Parameters are ordinary fp32 ``nn.Parameter``s held by stock ``nn.Linear`` / ``nn.LayerNorm`` /
``nn.Conv2d`` containers (so optimisers, ``state_dict`` and name-based ``requires_grad``
filters behave as before); the containers' own ``forward`` is never used.  GEMM operands and
intermediate activations are in ``vitk.get_compute_dtype()`` (bf16 by default), the residual
stream between blocks is fp32, and every FLOP of the five modules executes in libvitk.so --
there is no PyTorch fallback, CPU tensors raise.
"""
import logging
from os.path import join as pjoin

import numpy as np
import torch
import torch.nn as nn

import vitk
from vitk import ops

logger = logging.getLogger(__name__)

# JAX checkpoint sub-keys consumed by load_from (reference modeling.py:26-33).
_JAX_BLOCK_KEYS = {
    "q": "MultiHeadDotProductAttention_1/query",
    "k": "MultiHeadDotProductAttention_1/key",
    "v": "MultiHeadDotProductAttention_1/value",
    "o": "MultiHeadDotProductAttention_1/out",
    "fc1": "MlpBlock_3/Dense_0",
    "fc2": "MlpBlock_3/Dense_1",
    "ln1": "LayerNorm_0",
    "ln2": "LayerNorm_2",
}
ATTENTION_Q, ATTENTION_K = _JAX_BLOCK_KEYS["q"], _JAX_BLOCK_KEYS["k"]
ATTENTION_V, ATTENTION_OUT = _JAX_BLOCK_KEYS["v"], _JAX_BLOCK_KEYS["o"]
FC_0, FC_1 = _JAX_BLOCK_KEYS["fc1"], _JAX_BLOCK_KEYS["fc2"]
ATTENTION_NORM, MLP_NORM = _JAX_BLOCK_KEYS["ln1"], _JAX_BLOCK_KEYS["ln2"]

LN_EPS = 1e-6


def np2th(weights, conv=False):
    """numpy -> torch; ``conv=True`` reorders a HWIO kernel to OIHW."""
    a = np.asarray(weights)
    if conv:
        a = np.transpose(a, (3, 2, 0, 1))
    return torch.from_numpy(np.ascontiguousarray(a))


def _to_compute(x):
    dt = vitk.get_compute_dtype()
    return x if x.dtype == dt else x.to(dt)


def _to_stream(x):
    return x if x.dtype == vitk.STREAM_DTYPE else x.to(vitk.STREAM_DTYPE)


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Attention(nn.Module):
    """Multi-head self-attention: one fused QKV GEMM, fused online-softmax attention that reads
    and writes heads in place (no transpose_for_scores copies), output projection GEMM."""

    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.num_attention_heads = config.transformer["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        hidden = config.hidden_size
        self.query = nn.Linear(hidden, self.all_head_size)
        self.key = nn.Linear(hidden, self.all_head_size)
        self.value = nn.Linear(hidden, self.all_head_size)
        self.out = nn.Linear(hidden, hidden)
        # one GEMM produces the three weight gradients (and one column sum the three bias gradients):
        # a gradient bucketer keeps their slices adjacent so that it can do so in place
        ops.colocate_grads(self.query.weight, self.key.weight, self.value.weight)
        ops.colocate_grads(self.query.bias, self.key.bias, self.value.bias)
        rate = config.transformer["attention_dropout_rate"]
        # The reference applies nn.Dropout(attention_dropout_rate) to the probabilities and to
        # the projection (modeling.py:63-64); every preset sets the rate to 0.0.
        self.attn_dropout = nn.Dropout(rate)
        self.proj_dropout = nn.Dropout(rate)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, hidden_states, residual=None):
        """hidden_states [B,N,D] -> (attention_output [B,N,D], probs [B,H,N,N] or None).
        ``residual`` (optional, not in the reference signature) is added inside the output
        projection's epilogue; ``Block`` uses it to fuse `x + h`."""
        x = _to_compute(hidden_states)
        B, N, D = x.shape
        H = self.num_attention_heads
        qkv = ops.linear(x.reshape(B * N, D),
                         (self.query.weight, self.key.weight, self.value.weight),
                         (self.query.bias, self.key.bias, self.value.bias))
        ctx, lse = ops.attention(qkv, B, N, H, dropout_p=self.attn_dropout.p if self.training else 0.0)
        weights = ops.attention_probs(qkv.detach(), lse, B, N, H) if self.vis else None
        p = self.proj_dropout.p if self.training else 0.0
        out = ops.linear(ctx, (self.out.weight,), (self.out.bias,), residual=residual, dropout_p=p)
        return out.view(B, N, D), weights


class Mlp(nn.Module):
    """fc1 -> exact GELU -> dropout -> fc2 -> dropout as two GEMMs with fused epilogues."""

    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.transformer["mlp_dim"])
        self.fc2 = nn.Linear(config.transformer["mlp_dim"], config.hidden_size)
        self.act_fn = torch.nn.functional.gelu
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        self._init_weights()

    def _init_weights(self):
        for fc in (self.fc1, self.fc2):
            nn.init.xavier_uniform_(fc.weight)
        for fc in (self.fc1, self.fc2):
            nn.init.normal_(fc.bias, std=1e-6)

    def forward(self, x, residual=None):
        p = self.dropout.p if self.training else 0.0
        return ops.mlp(_to_compute(x), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                       residual=residual, dropout_p=p)


class Embeddings(nn.Module):
    """Patch + position embeddings: im2col + MFMA GEMM in place of the strided Conv2d."""

    def __init__(self, config, img_size, in_channels=3):
        super().__init__()
        self.hybrid = None
        img_size = _pair(img_size)
        grid = config.patches.get("grid")
        if grid is not None:
            patch_size = (img_size[0] // 16 // grid[0], img_size[1] // 16 // grid[1])
            n_patches = (img_size[0] // 16) * (img_size[1] // 16)
            self.hybrid = True
        else:
            patch_size = _pair(config.patches["size"])
            n_patches = (img_size[0] // patch_size[0]) * (img_size[1] // patch_size[1])
            self.hybrid = False
        if self.hybrid:
            # the convolutional stem (kernel library, channels-last); its feature map is projected to tokens by
            # the GEMM below (reference :140-147)
            from .modeling_resnet import ResNetV2
            self.hybrid_model = ResNetV2(block_units=config.resnet.num_layers,
                                         width_factor=config.resnet.width_factor)
            in_channels = self.hybrid_model.width * 16
        self.patch_embeddings = nn.Conv2d(in_channels=in_channels, out_channels=config.hidden_size,
                                          kernel_size=patch_size, stride=patch_size)
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches + 1, config.hidden_size))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])

    def forward(self, x):
        p = self.dropout.p if self.training else 0.0
        if self.hybrid:
            return self._forward_hybrid(x, p)
        return ops.patch_embed(x, self.patch_embeddings.weight, self.patch_embeddings.bias,
                               self.cls_token, self.position_embeddings, p, vitk.get_compute_dtype())

    def _forward_hybrid(self, x, p):
        """Stem feature map -> tokens (reference :157-165).  The map needs a gradient, unlike an
        image, so the patch projection is the generic linear operator (with its dX GEMM) on the
        unfolded map instead of the image-only patch-embedding operator; class token, position
        embeddings and dropout are a handful of elementwise passes over [B, N, D] once per step."""
        feat = self.hybrid_model(x.float())
        B, C = feat.shape[:2]
        kh, kw = self.patch_embeddings.kernel_size
        if (kh, kw) == (1, 1):
            cols = feat.flatten(2).transpose(1, 2)                              # [B, n, C]
        else:
            cols = torch.nn.functional.unfold(feat, (kh, kw), stride=(kh, kw)).transpose(1, 2)   # [B, n, C*kh*kw]
        n = cols.shape[1]
        w = self.patch_embeddings.weight.reshape(self.patch_embeddings.out_channels, -1)
        tok = ops.linear(_to_compute(cols.reshape(B * n, -1)), (w,), (self.patch_embeddings.bias,))
        tok = tok.float().view(B, n, -1)
        emb = torch.cat((self.cls_token.expand(B, -1, -1), tok), dim=1) + self.position_embeddings
        return torch.nn.functional.dropout(emb, p, self.training) if p > 0 else emb


class Block(nn.Module):
    """Pre-LN transformer block.  The residual stream (block input/output) is fp32; each
    LayerNorm reads it, emits the compute-dtype GEMM operand and also hands its input back for the
    residual branch, so backward adds the two gradient streams inside the LayerNorm kernel; both
    residual adds happen in GEMM epilogues."""

    def __init__(self, config, vis):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis)

    def forward(self, x):
        cdt = vitk.get_compute_dtype()
        x = _to_stream(x)          # the residual stream stays fp32; LayerNorm hands bf16 to the GEMMs
        h, y = ops.add_norm(x, self.attention_norm.weight, self.attention_norm.bias, self.attention_norm.eps, cdt)
        x, weights = self.attn(y, residual=h)
        h, y = ops.add_norm(x, self.ffn_norm.weight, self.ffn_norm.bias, self.ffn_norm.eps, cdt)
        x = self.ffn(y, residual=h)
        return x, weights

    def load_from(self, weights, n_block):
        """Copy one encoder block out of a JAX ``.npz`` checkpoint ([in,out] kernels are transposed)."""
        root = f"Transformer/encoderblock_{n_block}"
        hs = self.hidden_size

        def arr(sub, leaf):
            return np2th(weights[pjoin(root, _JAX_BLOCK_KEYS[sub], leaf)])

        with torch.no_grad():
            for sub, lin in (("q", self.attn.query), ("k", self.attn.key), ("v", self.attn.value),
                             ("o", self.attn.out)):
                lin.weight.copy_(arr(sub, "kernel").view(hs, hs).t())
                lin.bias.copy_(arr(sub, "bias").view(-1))
            for sub, lin in (("fc1", self.ffn.fc1), ("fc2", self.ffn.fc2)):
                lin.weight.copy_(arr(sub, "kernel").t())
                lin.bias.copy_(arr(sub, "bias").t())
            for sub, ln in (("ln1", self.attention_norm), ("ln2", self.ffn_norm)):
                ln.weight.copy_(arr(sub, "scale"))
                ln.bias.copy_(arr(sub, "bias"))


class Encoder(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=LN_EPS)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(Block(config, vis))

    def forward(self, hidden_states):
        attn_weights = []
        for layer_block in self.layer:
            hidden_states, weights = layer_block(hidden_states)
            if self.vis:
                attn_weights.append(weights)
        # final norm writes fp32 so that a stock fp32 classifier head can consume it
        encoded = ops.layer_norm(_to_stream(hidden_states), self.encoder_norm.weight, self.encoder_norm.bias,
                                 self.encoder_norm.eps, out_dtype=torch.float32)
        return encoded, attn_weights


class Transformer(nn.Module):
    def __init__(self, config, img_size, vis):
        super().__init__()
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)

    def forward(self, input_ids):
        return self.encoder(self.embeddings(input_ids))


class VisionTransformer(nn.Module):
    """Caller of the hot path (reference modeling.py:260-334): classifier head and loss stay on
    stock PyTorch ops."""

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        super().__init__()
        self.num_classes = num_classes
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.transformer = Transformer(config, img_size, vis)
        self.head = nn.Linear(config.hidden_size, num_classes)

    def forward(self, x, labels=None, task=None):
        x, attn_weights = self.transformer(x)
        logits = self.head(x[:, 0])
        if labels is None:
            return logits, attn_weights
        return nn.functional.cross_entropy(logits.view(-1, self.num_classes), labels.view(-1))

    def load_from(self, weights):
        """Load a JAX ``.npz`` checkpoint; position embeddings are bilinearly resized
        (scipy ``zoom(order=1)``) when the token grid differs."""
        emb, enc = self.transformer.embeddings, self.transformer.encoder
        with torch.no_grad():
            if self.zero_head:
                nn.init.zeros_(self.head.weight)
                nn.init.zeros_(self.head.bias)
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
            if emb.hybrid:                          # reference :325-334
                root = emb.hybrid_model.root
                root.conv.weight.copy_(np2th(weights["conv_root/kernel"], conv=True))
                root.gn.weight.copy_(np2th(weights["gn_root/scale"]).view(-1))
                root.gn.bias.copy_(np2th(weights["gn_root/bias"]).view(-1))
                for bname, block in emb.hybrid_model.body.named_children():
                    for uname, unit in block.named_children():
                        unit.load_from(weights, n_block=bname, n_unit=uname)

    def _fit_posemb(self, posemb, ntok_new):
        if posemb.shape[1] == ntok_new:
            return torch.from_numpy(posemb)
        from scipy import ndimage
        logger.info("load_pretrained: resized variant: %s to %d tokens", posemb.shape, ntok_new)
        n_special = 1 if self.classifier == "token" else 0
        special, grid = posemb[:, :n_special], posemb[0, n_special:]
        gs_old = int(np.sqrt(len(grid)))
        gs_new = int(np.sqrt(ntok_new - n_special))
        print("load_pretrained: grid-size from %s to %s" % (gs_old, gs_new))
        grid = grid.reshape(gs_old, gs_old, -1)
        factor = gs_new / gs_old
        grid = ndimage.zoom(grid, (factor, factor, 1), order=1).reshape(1, gs_new * gs_new, -1)
        return torch.from_numpy(np.concatenate([special, grid], axis=1))
