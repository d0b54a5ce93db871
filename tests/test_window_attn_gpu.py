"""Swin window attention kernel (vitk_window_attn_fwd / _bwd through ops.window_attention) against the
CPU restatement of reference models/Swin/swin_RKR.py in oracle/swin_oracle.py: explicit roll, window
partition, relative-position-bias gather, -100 shift mask, softmax, window reverse, roll back.
fp32 at 2e-5, bf16 at 2e-2 of each tensor's largest entry.  Parity unpinned against the reference file
itself (it needs `timm`, absent here); the oracle restates it line by line."""
import numpy as np
import pytest
import torch

from oracle import swin_oracle as SO

pytestmark = pytest.mark.gpu
DEV = "cuda"
# (B, H, W, heads, window, shift)
CASES = [(2, 14, 14, 3, 7, 0), (2, 14, 14, 3, 7, 3), (1, 28, 21, 4, 7, 3), (3, 8, 8, 2, 4, 2), (2, 8, 16, 2, 8, 4),
         (2, 7, 7, 6, 7, 0), (1, 6, 4, 1, 2, 1)]


def _inputs(B, H, W, heads, ws, seed):
    g = torch.Generator().manual_seed(seed)
    C = heads * 32
    qkv = torch.randn(B * H * W, 3 * C, generator=g)
    table = 0.5 * torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    dout = torch.randn(B * H * W, C, generator=g)
    return qkv, table, dout


@pytest.fixture(autouse=True)
def _default_kernels():
    import vitk
    yield
    vitk.lib.vitk_debug_force_window_attn_valu(0)


@pytest.mark.parametrize("dtype,valu", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)],
                         ids=["fp32", "bf16-mfma", "bf16-valu"])
@pytest.mark.parametrize("B,H,W,heads,ws,shift", CASES)
def test_window_attention_matches_oracle(dtype, valu, B, H, W, heads, ws, shift):
    import vitk
    from vitk import ops
    vitk.lib.vitk_debug_force_window_attn_valu(valu)
    qkv, table, dout = _inputs(B, H, W, heads, ws, seed=B * 1000 + H * 10 + shift)
    if dtype == torch.bfloat16:                        # both sides start from the same rounded operands
        qkv, dout = qkv.bfloat16().float(), dout.bfloat16().float()
    q_ref, t_ref = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    o_ref = SO.shifted_window_attention(q_ref, t_ref, B, H, W, heads, ws, shift)
    o_ref.backward(dout)
    q_dev = qkv.to(DEV, dtype).requires_grad_(True)
    t_dev = table.to(DEV).requires_grad_(True)
    o_dev = ops.window_attention(q_dev, t_dev, B, H, W, heads, ws, shift)
    o_dev.backward(dout.to(DEV, dtype))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for name, got, want in (("out", o_dev, o_ref), ("dqkv", q_dev.grad, q_ref.grad), ("dtable", t_dev.grad, t_ref.grad)):
        got, want = got.detach().float().cpu().numpy(), want.detach().numpy()
        assert got.shape == want.shape, name
        err, scale = np.abs(got - want).max(), np.abs(want).max()
        assert err <= tol * scale + 1e-7, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


def test_window_attention_shift_mask_blocks_cross_region_attention():
    """With a shift, tokens that wrap around the image border must not attend across regions: moving
    the values of one region leaves the outputs of the others unchanged to exp(-100)."""
    from vitk import ops
    B, H, W, heads, ws, shift = 1, 14, 14, 2, 7, 3
    qkv, table, _ = _inputs(B, H, W, heads, ws, seed=5)
    C = heads * 32
    a = ops.window_attention(qkv.to(DEV), table.to(DEV), B, H, W, heads, ws, shift).cpu()
    q2 = qkv.clone().view(H, W, 3 * C)
    q2[:shift, :, 2 * C:] += 7.0                       # value vectors of the rows that wrap to the bottom
    b = ops.window_attention(q2.view(-1, 3 * C).to(DEV), table.to(DEV), B, H, W, heads, ws, shift).cpu()
    diff = (a - b).abs().view(H, W, C).amax(-1)
    assert float(diff[:shift].min()) > 1.0             # the moved rows see their own values
    assert float(diff[shift:].max()) < 1e-5            # nobody else does


def test_window_attention_rejects_bad_geometry():
    import vitk
    from vitk import ops
    qkv = torch.zeros(2 * 14 * 14, 3 * 64, device=DEV)
    table = torch.zeros(169, 2, device=DEV)
    with pytest.raises(vitk.VitkError):
        ops.window_attention(qkv, table, 2, 14, 14, 2, 5, 0)          # 14 % 5 != 0
    with pytest.raises(vitk.VitkError):
        ops.window_attention(qkv, table, 2, 14, 14, 2, 7, 7)          # shift must be below the window size
    with pytest.raises(vitk.VitkError):
        ops.window_attention(torch.zeros(2 * 14 * 14, 3 * 128, device=DEV), table, 2, 14, 14, 2, 7, 0)   # head dim 64
