"""Dropout on the attention probabilities (reference models/ViT/modeling.py:63,86: nn.Dropout between the softmax and
the product with v), through the C-ABI entry points vitk_attn_fwd_dropout / vitk_attn_bwd_dropout.

torch's Philox stream cannot be reproduced, so the check is made with the kernel's OWN mask: with one-hot value
rows the context IS the dropped, rescaled probability matrix, which yields the mask; a plain PyTorch fp32 computation
with that mask (softmax, mask / keep, product with v, autograd) is then the reference for the output and for dq, dk,
dv on arbitrary values.  Tolerances as for the other kernels: fp32 2e-5, bf16 2e-2 of the reference's maximum.
"""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def _rnd(shape, seed, dtype, scale=1.0):
    a = np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * scale
    return torch.from_numpy(a).to(DEV).to(dtype)


def _fwd(vitk, qkv, B, N, H, p, seed):
    T, D3 = qkv.shape
    D, dh = D3 // 3, D3 // 3 // H
    out = torch.empty((T, D), dtype=qkv.dtype, device=DEV)
    o32 = torch.empty((T, D), dtype=torch.float32, device=DEV) if qkv.dtype == torch.bfloat16 else None
    lse = torch.empty((B, H, N), dtype=torch.float32, device=DEV)
    esz, base, vp = qkv.element_size(), qkv.data_ptr(), ctypes.c_void_p
    rc = vitk.lib.vitk_attn_fwd_dropout(vp(base), vp(base + D * esz), vp(base + 2 * D * esz), vp(out.data_ptr()),
                                        vp(o32.data_ptr()) if o32 is not None else None, vp(lse.data_ptr()), B, H, N, dh,
                                        D3, D, 1.0 / math.sqrt(dh), p, seed, vitk.dtype_code(qkv.dtype),
                                        vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return out, o32, lse


def _bwd(vitk, qkv, out, o32, lse, dout, B, N, H, p, seed):
    T, D3 = qkv.shape
    D, dh = D3 // 3, D3 // 3 // H
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=DEV)
    esz, b0, d0, vp = qkv.element_size(), qkv.data_ptr(), dqkv.data_ptr(), ctypes.c_void_p
    rc = vitk.lib.vitk_attn_bwd_dropout(vp(b0), vp(b0 + D * esz), vp(b0 + 2 * D * esz), vp(out.data_ptr()),
                                        vp(o32.data_ptr()) if o32 is not None else None, vp(dout.data_ptr()),
                                        vp(lse.data_ptr()), vp(delta.data_ptr()), vp(d0), vp(d0 + D * esz),
                                        vp(d0 + 2 * D * esz), B, H, N, dh, D3, D, D3, 1.0 / math.sqrt(dh), p, seed,
                                        vitk.dtype_code(qkv.dtype), vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return dqkv


def _heads(x, B, N, H):           # [T, D] -> [B, H, N, dh]
    return x.reshape(B, N, H, -1).permute(0, 2, 1, 3)


def _recover_mask(vitk, qkv, B, N, H, p, seed):
    """[B,H,N,N] keep mask of (p, seed): value rows one-hot over 64 keys at a time make the context equal to the dropped
    probabilities of those keys."""
    T, D3 = qkv.shape
    D = D3 // 3
    dropped = torch.zeros(B, H, N, N, device=DEV)
    for k0 in range(0, N, 64):
        probe = qkv.clone()
        v = torch.zeros(B, N, H, 64, device=DEV)
        for key in range(k0, min(N, k0 + 64)):
            v[:, key, :, key - k0] = 1.0
        probe[:, 2 * D:] = v.reshape(T, D).to(qkv.dtype)
        out, o32, _ = _fwd(vitk, probe, B, N, H, p, seed)
        o = _heads((o32 if o32 is not None else out).float(), B, N, H)        # [B,H,N,64]
        w = min(64, N - k0)
        dropped[:, :, :, k0:k0 + w] = o[..., :w]
    return dropped > 0, dropped


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,N,p", [(2, 2, 40, 0.25), (1, 3, 100, 0.1), (2, 1, 197, 0.5)])
def test_attention_dropout_against_torch_with_the_kernels_mask(dtype, B, H, N, p):
    import vitk
    D = H * 64
    seed = 0x1234567 + N
    qkv = _rnd((B * N, 3 * D), 11 + N, dtype, 0.7)
    keep, dropped = _recover_mask(vitk, qkv, B, N, H, p, seed)
    frac = float(keep.float().mean())
    n = keep.numel()
    assert abs(frac - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n) + 1e-3, f"keep fraction {frac} vs {1 - p}"
    # the same mask again for the same seed, another one for another seed
    keep2, _ = _recover_mask(vitk, qkv, B, N, H, p, seed)
    assert torch.equal(keep, keep2)
    keep3, _ = _recover_mask(vitk, qkv, B, N, H, p, seed + 1)
    assert not torch.equal(keep, keep3)

    # reference with that mask (fp32 on the dtype-rounded inputs)
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = (_heads(x[:, i * D:(i + 1) * D], B, N, H) for i in range(3))
    probs = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    # dropped probabilities seen through one-hot values are the reference's masked, rescaled probabilities
    ref_dropped = (probs * keep / (1 - p)).detach()
    e = float((dropped - ref_dropped).abs().max() / ref_dropped.abs().max())
    assert e <= TOL[dtype], f"dropped probabilities: {e:.3e}"
    ctx = (probs * keep / (1 - p)) @ v                                   # [B,H,N,64]
    ref_out = ctx.permute(0, 2, 1, 3).reshape(B * N, D)
    dout = _rnd((B * N, D), 5, dtype)
    ref_out.backward(dout.float())

    out, o32, lse = _fwd(vitk, qkv, B, N, H, p, seed)
    e = float((out.float() - ref_out.detach()).abs().max() / ref_out.detach().abs().max())
    print(f"{dtype} N={N} p={p}: out {e:.3e}")
    assert e <= TOL[dtype]
    ref_lse = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) / 8.0, dim=-1)
    assert float((lse - ref_lse).abs().max()) <= (1e-2 if dtype == torch.bfloat16 else 1e-4)
    dqkv = _bwd(vitk, qkv, out, o32, lse, dout, B, N, H, p, seed)
    for i, name in enumerate(("dq", "dk", "dv")):
        got, ref = dqkv[:, i * D:(i + 1) * D].float(), x.grad[:, i * D:(i + 1) * D]
        e = float((got - ref).abs().max() / ref.abs().max())
        print(f"{dtype} N={N} p={p}: {name} {e:.3e}")
        assert e <= TOL[dtype], f"{name}: {e:.3e}"


def test_zero_rate_is_the_plain_operator():
    import vitk
    B, H, N, D = 2, 2, 100, 128
    qkv = _rnd((B * N, 3 * D), 3, torch.bfloat16, 0.7)
    a, _, la = _fwd(vitk, qkv, B, N, H, 0.0, 99)
    out = torch.empty_like(a)
    lse = torch.empty_like(la)
    esz, base, vp = 2, qkv.data_ptr(), ctypes.c_void_p
    rc = vitk.lib.vitk_attn_fwd(vp(base), vp(base + D * esz), vp(base + 2 * D * esz), vp(out.data_ptr()), None,
                                vp(lse.data_ptr()), B, H, N, 64, 3 * D, D, 0.125, vitk.BF16,
                                vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert torch.equal(a, out) and torch.equal(la, lse)
    # rates outside [0, 1) are rejected
    rc = vitk.lib.vitk_attn_fwd_dropout(vp(base), vp(base + D * esz), vp(base + 2 * D * esz), vp(out.data_ptr()), None,
                                        vp(lse.data_ptr()), B, H, N, 64, 3 * D, D, 0.125, 1.0, 0, vitk.BF16,
                                        vp(torch.cuda.current_stream().cuda_stream))
    assert rc != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_with_attention_dropout_trains_and_eval_ignores_it(dtype):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    vitk.set_compute_dtype(dtype)
    try:
        cfg = cfgmod.get_tiny_config()
        cfg.transformer["attention_dropout_rate"] = 0.2
        cfg.transformer["dropout_rate"] = 0.0
        torch.manual_seed(0)
        model = VisionTransformer(cfg, 32, num_classes=10).to(DEV)
        plain_cfg = cfgmod.get_tiny_config()
        plain_cfg.transformer["dropout_rate"] = 0.0
        plain = VisionTransformer(plain_cfg, 32, num_classes=10).to(DEV)
        plain.load_state_dict(model.state_dict())
        x = torch.randn(4, 3, 32, 32, device=DEV)
        y = torch.randint(0, 10, (4,), device=DEV)
        model.eval(); plain.eval()
        with torch.no_grad():
            assert torch.equal(model(x)[0], plain(x)[0])             # eval: dropout is the identity
        model.train(); plain.train()
        l1 = model(x, y)
        l0 = plain(x, y)
        assert float(l1.detach()) != float(l0.detach())               # training: the mask acts
        l1.backward()
        for n_, p_ in model.named_parameters():
            assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n_
    finally:
        vitk.set_compute_dtype(torch.bfloat16)
