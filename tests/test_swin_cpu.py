"""Swin RKR model on the CPU: state_dict layout of the drop-in module against the parameter names of
reference models/Swin/swin_RKR.py (as oracle/swin_oracle.synth_swin_params lists them), the buffers the
reference registers, and two properties of the oracle's window attention that do not depend on the
reference being importable (it needs `timm`: parity of the Swin path is unpinned)."""
import pytest
import torch

from oracle import swin_oracle as SO

NCLS = [10, 7, 5]
TINY = dict(img=56, patch=4, embed=64, depths=[2, 2], heads=[2, 4], window=7, mlp_ratio=4.0)
TINY96 = dict(img=56, patch=4, embed=96, depths=[2, 1], heads=[3, 6], window=7, mlp_ratio=2.0)


def swin_config(geo, RG=True, SFG=True, drop_path=0.0, variant="rkr", fn="binarizer"):
    from models import config as cfgmod
    cfg = cfgmod.get_config({"rkr": "Swin_RKR", "pb": "Swin_PB", "rkrpb": "Swin_RKRPB"}[variant], "cifar100")
    if variant != "rkr":
        cfg.threshold_fn = fn
    cfg.embed_dim, cfg.depths, cfg.num_heads = geo["embed"], list(geo["depths"]), list(geo["heads"])
    cfg.window_size, cfg.mlp_ratio, cfg.patch_size = geo["window"], geo["mlp_ratio"], geo["patch"]
    cfg.drop_path_rate, cfg.drop_rate = drop_path, 0.0
    cfg.RG, cfg.SFG, cfg.K, cfg.rkr_scale, cfg.multi_head, cfg.task_num = RG, SFG, 2, 1e-1, True, len(NCLS)
    return cfg


@pytest.mark.parametrize("geo", [TINY, TINY96], ids=["embed64", "embed96"])
def test_swin_module_state_dict_layout(geo):
    from models.Swin.swin_RKR import SwinTransformer
    model = SwinTransformer(swin_config(geo), geo["img"], num_classes=NCLS)
    want = SO.synth_swin_params(geo, NCLS, seed=3)
    have = model.state_dict()
    buffers = {k for k in have if k.endswith("relative_position_index") or k.endswith("attn_mask")}
    assert set(have) - buffers == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    # block 1 of stage 0 is shifted and carries the reference's mask buffer; stage 1 is a single 7x7 window
    assert "layers.0.blocks.1.attn_mask" in buffers and "layers.0.blocks.0.attn_mask" not in buffers
    assert not any(k.startswith("layers.1.") and k.endswith("attn_mask") for k in buffers)
    blk = model.layers[0].blocks[1]
    assert torch.equal(blk.attn_mask, SO.shift_mask(14, 14, 7, 3))
    assert torch.equal(blk.attn.relative_position_index, SO.relative_position_index(7))
    last = model.layers[1].blocks[-1]
    assert last.shift_size == 0 and last.window_size == 7
    missing, unexpected = model.load_state_dict(want, strict=False)
    assert set(missing) == buffers and not unexpected


@pytest.mark.parametrize("variant", ["pb", "rkrpb"])
def test_swin_pb_variants_state_dict_layout(variant):
    """swin_PB.py / swin_RKRPB.py: the RKR file with the piggyback layer types (reference files differ in
    nothing else), same parameter names as those layers have in the ViT models."""
    import importlib
    mod = importlib.import_module({"pb": "models.Swin.swin_PB", "rkrpb": "models.Swin.swin_RKRPB"}[variant])
    from models.Swin import layers_PB
    model = mod.SwinTransformer(swin_config(TINY, variant=variant), TINY["img"], num_classes=NCLS)
    want = SO.synth_swin_params(TINY, NCLS, seed=3, variant=variant)
    have = model.state_dict()
    buffers = {k for k in have if k.endswith("relative_position_index") or k.endswith("attn_mask")}
    assert set(have) - buffers == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    fc = model.layers[0].blocks[1].attn.qkv
    assert isinstance(fc, layers_PB.ElementWiseLinear if variant == "pb" else layers_PB.ElementWiseRG)
    assert isinstance(model.layers[0].blocks[1], mod.SwinTransformerBlock) and isinstance(model.layers[1], mod.BasicLayer)
    assert hasattr(model.layers[0].blocks[0].mlp, "sfg1") == (variant == "rkrpb")


def test_oracle_window_attention_without_shift_is_blockwise():
    """Without a shift each window is independent: perturbing one window's tokens changes only that window."""
    B, H, W, heads, ws = 1, 14, 14, 2, 7
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * H * W, 3 * heads * 32, generator=g)
    table = torch.randn(169, heads, generator=g)
    a = SO.shifted_window_attention(qkv, table, B, H, W, heads, ws, 0)
    q2 = qkv.clone().view(H, W, -1)
    q2[:7, :7] += 1.0
    b = SO.shifted_window_attention(q2.view(B * H * W, -1), table, B, H, W, heads, ws, 0)
    d = (a - b).abs().view(H, W, -1).amax(-1)
    assert float(d[:7, :7].min()) > 0 and float(d[7:, :].max()) == 0 and float(d[:, 7:].max()) == 0


def test_oracle_shift_equals_roll_of_unshifted_when_unmasked():
    """The shifted attention is the un-shifted attention of the rolled image, except across the wrap-around
    border where the -100 mask applies: windows that contain no wrapped tokens agree exactly."""
    B, H, W, heads, ws, shift = 1, 14, 14, 2, 7, 3
    g = torch.Generator().manual_seed(2)
    C3 = 3 * heads * 32
    qkv = torch.randn(B * H * W, C3, generator=g)
    table = torch.randn(169, heads, generator=g)
    a = SO.shifted_window_attention(qkv, table, B, H, W, heads, ws, shift).view(H, W, -1)
    rolled = torch.roll(qkv.view(H, W, C3), shifts=(-shift, -shift), dims=(0, 1)).reshape(B * H * W, C3)
    b = SO.shifted_window_attention(rolled, table, B, H, W, heads, ws, 0).view(H, W, -1)
    b = torch.roll(b, shifts=(shift, shift), dims=(0, 1))
    # first window of the shifted frame = original rows/cols shift .. shift+6: no wrapped token inside
    assert torch.allclose(a[shift:shift + 7, shift:shift + 7], b[shift:shift + 7, shift:shift + 7], atol=1e-6)
    assert not torch.allclose(a[:shift], b[:shift], atol=1e-3)          # wrapped rows are masked apart
