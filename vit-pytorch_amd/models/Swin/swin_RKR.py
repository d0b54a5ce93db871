"""Swin Transformer with RKR task-conditioned linears on the vitk HIP kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/Swin/swin_RKR.py``: ``RG_FC`` (:13-43), ``SFG_FC`` (:45-56), ``Mlp``
(:58-87), ``window_partition`` / ``window_reverse`` (:90-119), ``WindowAttention`` (:122-215),
``SwinTransformerBlock`` (:234-343), ``PatchMerging`` (:364-401), ``BasicLayer`` (:413-470),
``PatchEmbed`` (:484-522) and ``SwinTransformer`` (:532-683) keep their constructor signatures,
``forward(x, task)`` / ``forward(x, labels, task)``, attribute names, buffers and ``state_dict`` keys.
``DropPath``, ``to_2tuple`` and ``trunc_normal_`` (taken from ``timm`` by the reference) are defined
here.

How a block runs.  The qkv projection acts per token, so the cyclic shift, ``window_partition``,
``window_reverse`` and the shift back commute with it: LayerNorm and ONE qkv GEMM run on the tokens
in their natural order, the window-attention kernel picks every window's tokens (and the -100 shift
mask, and the relative-position bias) by index arithmetic, and writes its output back in natural
order for the output-projection GEMM, which adds the residual in its epilogue.  No roll, partition
or mask tensor is ever materialised (the ``attn_mask`` / ``relative_position_index`` buffers exist for
``state_dict`` compatibility only).  RKR corrections are folded into the GEMM weight operand as in
``models/ViT/modeling_RKR.py``.  Stochastic depth scales a branch per sample, so blocks that have it
active in training add the residual with one fused pass (``ops.scale_add``) instead of the GEMM epilogue.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import vitk
from vitk import ops

from models.ViT import modeling as base
from models.ViT.modeling_RKR import RG_FC, SFG_FC  # noqa: F401  (same layers, same state_dict keys, reference :13-56)


def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth: drops the whole residual branch of a sample with probability ``drop_prob``."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def sample_scale(self, x):
        """Per-sample factor (kept: 1 / keep-probability, dropped: 0), fp32 [batch]."""
        keep = 1.0 - self.drop_prob
        return torch.empty(x.shape[0], dtype=torch.float32, device=x.device).bernoulli_(keep) / keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        scale = self.sample_scale(x)
        return x * scale.view((x.shape[0],) + (1,) * (x.ndim - 1)).to(x.dtype)


class _RkrLayers:
    """Which task-conditioned layers a model file builds its linears from.  ``swin_PB.py`` and
    ``swin_RKRPB.py`` differ from this file in nothing else (as in the reference)."""

    @staticmethod
    def fc(config, h_in, h_out):
        return RG_FC(config, h_in, h_out)

    @staticmethod
    def sfg(config, c_out):
        """The output-feature scale that follows a linear, or None when the model has none."""
        return SFG_FC(c_out, config.task_num) if config.SFG else None


class Mlp(nn.Module):
    layers = _RkrLayers

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., config=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = self.layers.fc(config, in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = self.layers.fc(config, hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        sfg1, sfg2 = self.layers.sfg(config, hidden_features), self.layers.sfg(config, out_features)
        self.SFG = sfg1 is not None
        if self.SFG:
            self.sfg1, self.sfg2 = sfg1, sfg2

    def forward(self, x, task, residual=None):
        """``residual`` (not in the reference signature) is added in the second GEMM's epilogue."""
        if not isinstance(self.act, nn.GELU):
            raise NotImplementedError("the fused MLP implements GELU only")
        w1, b1 = self.fc1.effective(task, self.sfg1 if self.SFG else None)
        w2, b2 = self.fc2.effective(task, self.sfg2 if self.SFG else None)
        p = self.drop.p if self.training else 0.0
        return ops.mlp(base._to_compute(x), w1, b1, w2, b2, residual=residual, dropout_p=p)


def window_partition(x, window_size):
    """(B, H, W, C) -> (num_windows*B, window_size, window_size, C).  Kept for callers of the reference
    API; the blocks below never call it."""
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    """(num_windows*B, window_size, window_size, C) -> (B, H, W, C)."""
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class WindowAttention(nn.Module):
    """Window based multi-head self attention with relative position bias (reference :122-215)."""
    layers = _RkrLayers

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.,
                 config=None):
        super().__init__()
        self.dim = dim
        self.window_size = window_size  # Wh, Ww
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        coords = torch.stack(torch.meshgrid([torch.arange(window_size[0]), torch.arange(window_size[1])], indexing="ij"))
        flat = torch.flatten(coords, 1)
        rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size[0] - 1
        rel[:, :, 1] += window_size[1] - 1
        rel[:, :, 0] *= 2 * window_size[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = self.layers.fc(config, dim, dim * 3)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = self.layers.fc(config, dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)
        sfg = self.layers.sfg(config, dim)
        self.SFG = sfg is not None
        if self.SFG:
            self.sfg_proj = sfg

    def forward_tokens(self, x, task, B, H, W, shift, residual=None):
        """x [B, H*W, C] (LayerNorm output, natural token order) -> [B, H*W, C]: qkv GEMM, shifted
        window attention, output projection (+ ``residual`` in its epilogue when given)."""
        if self.training and self.attn_drop.p > 0:
            raise NotImplementedError("dropout on attention probabilities is not implemented by the fused kernel")
        if self.window_size[0] != self.window_size[1]:
            raise NotImplementedError("square windows only")
        C = self.dim
        x = base._to_compute(x).reshape(B * H * W, C)
        wq, bq = self.qkv.effective(task)
        wo, bo = self.proj.effective(task, self.sfg_proj if self.SFG else None)
        qkv = ops.linear(x, (wq,), (bq,))
        ctx = ops.window_attention(qkv, self.relative_position_bias_table, B, H, W, self.num_heads,
                                   self.window_size[0], shift, self.scale)
        p = self.proj_drop.p if self.training else 0.0
        out = ops.linear(ctx, (wo,), (bo,), residual=residual, dropout_p=p)
        return out.view(B, H * W, C)

    def forward(self, x, task, mask=None):
        """Reference signature: x (num_windows*B, N, C) already partitioned.  Every window is handled as
        an image of its own; an explicit mask tensor is not supported (the blocks pass the shift to
        ``forward_tokens`` instead and the kernel derives the mask)."""
        if mask is not None:
            raise NotImplementedError("pass the shift to forward_tokens(); explicit masks are not supported")
        B_, N, C = x.shape
        return self.forward_tokens(x, task, B_, self.window_size[0], self.window_size[1], 0)

    def extra_repr(self) -> str:
        return f'dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}'


class SwinTransformerBlock(nn.Module):
    attention_cls, mlp_cls = WindowAttention, Mlp

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, config=None):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            # a window larger than the feature map: one window, no shift
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = self.attention_cls(
            dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
            qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop, config=config)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = self.mlp_cls(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop,
                       config=config)
        attn_mask = None
        if self.shift_size > 0:
            # same tensor as the reference registers (reference :281-300); the kernel does not read it
            H, W = self.input_resolution
            img_mask = torch.zeros((1, H, W, 1))
            cnt = 0
            for h in (slice(0, -self.window_size), slice(-self.window_size, -self.shift_size),
                      slice(-self.shift_size, None)):
                for w in (slice(0, -self.window_size), slice(-self.window_size, -self.shift_size),
                          slice(-self.shift_size, None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = window_partition(img_mask, self.window_size).view(-1, self.window_size * self.window_size)
            attn_mask = mw.unsqueeze(1) - mw.unsqueeze(2)
            attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, x, task):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        cdt = vitk.get_compute_dtype()
        dropping = self.training and isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0
        h, y = ops.add_norm(base._to_stream(x), self.norm1.weight, self.norm1.bias, self.norm1.eps, cdt)
        if dropping:       # shortcut + drop_path(branch): one fused pass over the stream
            x = ops.scale_add(h, self.attn.forward_tokens(y, task, B, H, W, self.shift_size), self.drop_path.sample_scale(h))
        else:
            x = self.attn.forward_tokens(y, task, B, H, W, self.shift_size, residual=h)
        h, y = ops.add_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, cdt)
        if dropping:
            return ops.scale_add(h, self.mlp(y, task), self.drop_path.sample_scale(h))
        return self.mlp(y, task, residual=h)

    def extra_repr(self) -> str:
        return f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, " \
               f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}"


class PatchMerging(nn.Module):
    """2x2 neighbours concatenated -> LayerNorm(4C) -> Linear(4C, 2C) without bias (reference :364-401)."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."
        x = base._to_stream(x).view(B, H, W, C)
        x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
        x = x.view(B, -1, 4 * C)
        y = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, vitk.get_compute_dtype())
        out = ops.linear(y.reshape(-1, 4 * C), (self.reduction.weight,), (torch.zeros(2 * C, device=y.device),))
        return out.float().view(B, -1, 2 * C)

    def extra_repr(self) -> str:
        return f"input_resolution={self.input_resolution}, dim={self.dim}"


class BasicLayer(nn.Module):
    block_cls = SwinTransformerBlock

    def __init__(self, dim, input_resolution, depth, num_heads, window_size,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False, config=None):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            self.block_cls(dim=dim, input_resolution=input_resolution,
                                 num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else window_size // 2,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                 drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer, config=config)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x, task):
        if self.use_checkpoint:
            raise NotImplementedError("activation checkpointing is not wired to the fused operators")
        for blk in self.blocks:
            x = blk(x, task)
        if self.downsample is not None:
            x = self.downsample(x)
        return x

    def extra_repr(self) -> str:
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"


class PatchEmbed(nn.Module):
    """Image to patch embedding (reference :484-522): the strided convolution as one GEMM over the
    flattened patches."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        ph, pw = self.patch_size
        gh, gw = H // ph, W // pw
        cols = x.view(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * ph * pw)
        tok = ops.linear(base._to_compute(cols), (self.proj.weight.reshape(self.embed_dim, -1),), (self.proj.bias,)).float().view(B, gh * gw, self.embed_dim)
        if self.norm is not None:
            tok = ops.layer_norm(tok, self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)
        return tok


class SwinTransformer(nn.Module):
    layer_cls = BasicLayer

    def __init__(self, config, img_size, num_classes, **kwargs):
        super().__init__()
        depths, num_heads = config.depths, config.num_heads
        embed_dim = config.embed_dim
        norm_layer = nn.LayerNorm
        self.multi_head = config.multi_head
        self.num_classes = num_classes
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = config.ape
        self.patch_norm = config.patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = config.mlp_ratio
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=config.patch_size, in_chans=config.in_chans,
                                      embed_dim=embed_dim, norm_layer=norm_layer if self.patch_norm else None)
        num_patches = self.patch_embed.num_patches
        patches_resolution = self.patch_embed.patches_resolution
        self.patches_resolution = patches_resolution
        if self.ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
            trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=config.drop_rate)
        dpr = [x.item() for x in torch.linspace(0, config.drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i_layer in range(self.num_layers):
            self.layers.append(self.layer_cls(
                dim=int(embed_dim * 2 ** i_layer),
                input_resolution=(patches_resolution[0] // (2 ** i_layer), patches_resolution[1] // (2 ** i_layer)),
                depth=depths[i_layer], num_heads=num_heads[i_layer], window_size=config.window_size,
                mlp_ratio=self.mlp_ratio, qkv_bias=config.qkv_bias, qk_scale=config.qk_scale,
                drop=config.drop_rate, attn_drop=0.,
                drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])],
                norm_layer=norm_layer,
                downsample=PatchMerging if (i_layer < self.num_layers - 1) else None,
                use_checkpoint=config.use_checkpoint, config=config))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        if self.multi_head:
            self.head = nn.ModuleList([nn.Linear(self.num_features, n) for n in num_classes])
        else:
            self.head = nn.Linear(self.num_features, num_classes)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def forward_features(self, x, task):
        x = self.patch_embed(x)
        if self.ape:
            x = x + self.absolute_pos_embed
        x = self.pos_drop(x)
        for layer in self.layers:
            x = layer(x, task)
        x = ops.layer_norm(base._to_stream(x), self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)
        return x.mean(dim=1)               # AdaptiveAvgPool1d(1) over the tokens + flatten

    def forward(self, x, labels=None, task=None):
        x = self.forward_features(x, task)
        x = self.head[task](x) if self.multi_head else self.head(x)
        if labels is not None:
            return F.cross_entropy(x, labels)
        return x, 0
