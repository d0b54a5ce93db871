"""CPU restatement of the Swin window attention of /root/reference/models/Swin/swin_RKR.py, for tests only
(the product path never imports this file).

PARITY UNPINNED against the reference itself: swin_RKR.py imports `timm`, which this environment does
not have, so the reference file cannot be imported to generate golden vectors, and the reference
ships no fixtures for it.  The functions below restate the reference line by line with plain torch
ops (roll, view/permute partitioning, index gather, masked_fill) and are what the HIP kernel - which
does none of those steps explicitly - is compared with.
"""
import torch


def window_partition(x, window_size):
    """swin_RKR.py:90-102: (B, H, W, C) -> (num_windows*B, ws, ws, C)."""
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    """swin_RKR.py:105-119."""
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def relative_position_index(ws):
    """swin_RKR.py:152-161."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(H, W, ws, shift):
    """swin_RKR.py:281-302: (num_windows, ws*ws, ws*ws) of 0 / -100, or None without shift."""
    if shift == 0:
        return None
    img_mask = torch.zeros((1, H, W, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img_mask, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


def window_attention_core(qkv_windows, table, heads, ws, scale, mask):
    """swin_RKR.py:184-209 between the qkv projection and the output projection:
    qkv_windows (B_, N, 3*C) -> (B_, N, C)."""
    B_, N, C3 = qkv_windows.shape
    C = C3 // 3
    qkv = qkv_windows.reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * scale) @ k.transpose(-2, -1)
    bias = table[relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B_, N, C)


def shifted_window_attention(qkv, table, B, H, W, heads, ws, shift, scale=None):
    """What SwinTransformerBlock.forward does around the attention core (swin_RKR.py:313-337), applied
    to the per-token qkv projection (the projection commutes with roll and partition):
    qkv (B*H*W, 3*C) in token order -> (B*H*W, C) in token order."""
    C3 = qkv.shape[1]
    C = C3 // 3
    scale = (C // heads) ** -0.5 if scale is None else scale
    x = qkv.view(B, H, W, C3)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, C3)
    ow = window_attention_core(xw, table, heads, ws, scale, shift_mask(H, W, ws, shift))
    o = window_reverse(ow.view(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * H * W, C)
