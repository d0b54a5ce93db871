"""Model presets for the ViT hot path.

Mirrors the preset registry of the reference (`/root/reference/models/config.py:323-356`,
ViT part `:99-208`) so that `get_config(model_type, dataset)` keeps returning an object
with the accessors the modules use (`config.hidden_size`, `config.transformer["num_heads"]`,
`config.patches.get("grid")`, ...).  The reference builds these with `ml_collections`;
that package is not a dependency here: `ConfigDict` below is a small attribute+item
mapping with the same surface the model code touches.

Two presets the reference keeps only in its unwired `models/ViT/_configs.py`
(`:224-232` R50 hybrid, `:242-255` L/16) are registered as well because the benchmark
configurations name them.
"""
from copy import deepcopy


class ConfigDict(dict):
    """dict with attribute access; nested dicts are wrapped on assignment."""

    def __init__(self, initial=None, **kw):
        super().__init__()
        for k, v in dict(initial or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __deepcopy__(self, memo):
        return ConfigDict({k: deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}


# ---------------------------------------------------------------------------
# ViT backbones: (patch, hidden, mlp, heads, layers)
# ---------------------------------------------------------------------------
_VIT_BACKBONES = {
    "B_16": dict(patch=16, hidden=768, mlp=3072, heads=12, layers=12),
    "L_16": dict(patch=16, hidden=1024, mlp=4096, heads=16, layers=24),
}


def _vit_backbone(name):
    b = _VIT_BACKBONES[name]
    c = ConfigDict()
    c.patches = ConfigDict(size=(b["patch"], b["patch"]))
    c.hidden_size = b["hidden"]
    c.transformer = ConfigDict(
        mlp_dim=b["mlp"],
        num_heads=b["heads"],
        num_layers=b["layers"],
        attention_dropout_rate=0.0,
        dropout_rate=0.1,
    )
    c.classifier = "token"
    c.representation_size = None
    return c


# Per-variant flag sets layered on the B/16 backbone (reference config.py:115-208).
_PB_FLAGS = dict(threshold_fn="binarizer", mask_scale=1e-2, mask_init="1s")
_VIT_B16_VARIANTS = {
    "ViT-B_16": dict(RG=False, SFG=False, multi_head=False, K=2),
    "ViT-B_16_MultiHead": dict(RG=False, SFG=False, multi_head=True, K=2, lamb=1.0, rkr_scale=1e-1),
    "ViT-B_16_RKR": dict(RG=True, SFG=True, multi_head=True, K=2, rkr_scale=1e-1),
    "ViT-B_16_RKRwoRG": dict(RG=False, SFG=True, multi_head=True, K=2, rkr_scale=1e-1),
    "ViT-B_16_RKRwoSFG": dict(RG=True, SFG=False, multi_head=True, K=2, rkr_scale=1e-1),
    "ViT-B_16_RKRTSN": dict(RG=True, SFG=False, multi_head=True, K=2, lamb=1.0, task_emb_d=50),
    "ViT-B_16_PB": dict(multi_head=True, **_PB_FLAGS),
    "ViT-B_16_RKRPB": dict(RG=True, SFG=True, multi_head=True, K=2, rkr_scale=1e-1,
                           RGnoPB=False, LMnoPB=False, RMnoPB=False, SFGnoPB=False, PBwR=False,
                           **_PB_FLAGS),
    "ViT-B_16_RKRPBG": dict(RG=True, SFG=True, multi_head=True, lamb=0.5, rkr_scale=1e-1),
    "ViT-B_16_PBG": dict(multi_head=True, lamb=0.25),
}

# ResNet-18 flag sets (reference config.py:9-91); no backbone fields.
_RESNET18_VARIANTS = {
    "ResNet18": dict(RG=False, SFG=False, multi_head=False),
    "ResNet18_MultiHead": dict(RG=False, SFG=False, multi_head=True),
    "ResNet18_RKR": dict(RG=True, SFG=True, multi_head=True, K=2, rkr_scale=1e-1),
    "ResNet18_RKRwoRG": dict(RG=False, SFG=True, multi_head=True, K=2, rkr_scale=1e-1),
    "ResNet18_RKRwoSFG": dict(RG=True, SFG=False, multi_head=True, K=2, rkr_scale=1e-1),
    "ResNet18_PB": dict(**_PB_FLAGS),
    "ResNet18_RKRPB": dict(RG=True, SFG=True, multi_head=True, K=2, rkr_scale=1e-1,
                           RGnoPB=False, SFGnoPB=False, LMnoPB=False, RMnoPB=False, PBwR=False,
                           **_PB_FLAGS),
    "ResNet18_PBG": dict(lamb=0.25),
}

_SWIN_VARIANTS = {
    "Swin": dict(RG=False, SFG=False, multi_head=False, K=2, rkr_scale=1e-1),
    "Swin_MultiHead": dict(RG=False, SFG=False, multi_head=True, K=2, rkr_scale=1e-1),
    "Swin_RKR": dict(RG=True, SFG=True, multi_head=True, K=2, rkr_scale=1e-1),
    "Swin_PB": dict(multi_head=True, **_PB_FLAGS),
    "Swin_RKRPB": dict(RG=True, SFG=True, multi_head=True, K=2, rkr_scale=1e-1,
                       RGnoPB=False, LMnoPB=False, RMnoPB=False, SFGnoPB=False, PBwR=False,
                       **_PB_FLAGS),
    "Swin_PBG": dict(multi_head=False, lamb=0.25),
}


def _swin_backbone(dataset):
    """Truncated Swin presets per dataset (reference config.py:218-263)."""
    c = ConfigDict(use_checkpoint=False, drop_rate=0.0, label_smoothing=0.1,
                   patch_size=4, in_chans=3, window_size=7, mlp_ratio=4.0,
                   qkv_bias=True, qk_scale=None, ape=False, patch_norm=True)
    if dataset == "cifar100":
        c.update(drop_path_rate=0.2, embed_dim=96, depths=[2, 2], num_heads=[3, 6])
    elif dataset == "imagenet":
        c.update(drop_path_rate=0.5, embed_dim=128, depths=[2, 2, 18], num_heads=[4, 8, 16, 32])
    elif dataset == "VD":
        c.update(drop_path_rate=0.2, embed_dim=128, depths=[2, 2, 18], num_heads=[4, 8, 16, 32])
    return c


def _with_flags(base, flags):
    out = deepcopy(base)
    for k, v in flags.items():
        out[k] = v
    return out


def get_l16_config():
    """ViT-L/16 (reference models/ViT/_configs.py:242-255)."""
    return _with_flags(_vit_backbone("L_16"), _VIT_B16_VARIANTS["ViT-B_16"])


def get_r50_b16_config():
    """R50 + ViT-B/16 hybrid (reference models/ViT/_configs.py:224-232)."""
    c = _with_flags(_vit_backbone("B_16"), _VIT_B16_VARIANTS["ViT-B_16"])
    del c.patches["size"]
    c.patches.grid = (14, 14)
    c.resnet = ConfigDict(num_layers=(3, 4, 9), width_factor=1)
    return c


def get_tiny_config(hidden=128, heads=2, mlp=256, layers=2, patch=16, dropout=0.0):
    """Small geometry used by the parity tests (head size 64 like every real preset)."""
    c = ConfigDict()
    c.patches = ConfigDict(size=(patch, patch))
    c.hidden_size = hidden
    c.transformer = ConfigDict(mlp_dim=mlp, num_heads=heads, num_layers=layers,
                               attention_dropout_rate=0.0, dropout_rate=dropout)
    c.classifier = "token"
    c.representation_size = None
    return _with_flags(c, _VIT_B16_VARIANTS["ViT-B_16"])


def get_config(model_type, dataset=None):
    """Same lookup contract as the reference's `get_config` (config.py:323-356)."""
    if model_type in _RESNET18_VARIANTS:
        return ConfigDict(_RESNET18_VARIANTS[model_type])
    if model_type in _VIT_B16_VARIANTS:
        return _with_flags(_vit_backbone("B_16"), _VIT_B16_VARIANTS[model_type])
    if model_type in _SWIN_VARIANTS:
        return _with_flags(_swin_backbone(dataset), _SWIN_VARIANTS[model_type])
    if model_type == "ViT-L_16":
        return get_l16_config()
    if model_type == "R50-ViT-B_16":
        return get_r50_b16_config()
    if model_type == "ViT-tiny":
        return get_tiny_config()
    raise KeyError(model_type)


CONFIG_NAMES = (tuple(_RESNET18_VARIANTS) + tuple(_VIT_B16_VARIANTS) + tuple(_SWIN_VARIANTS)
                + ("ViT-L_16", "R50-ViT-B_16", "ViT-tiny"))
