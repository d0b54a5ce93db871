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


# ----------------------------------------------------------------------------- whole model
def swin_geometry(cfg):
    """Per stage: (dim, (H, W), depth, heads, window, shifts) as SwinTransformer / BasicLayer /
    SwinTransformerBlock derive them (swin_RKR.py:611-626, 444-454, 263-266)."""
    res0 = cfg["img"] // cfg["patch"]
    out = []
    for i, (depth, heads) in enumerate(zip(cfg["depths"], cfg["heads"])):
        res = res0 // (2 ** i)
        ws = cfg["window"]
        shifts = [0 if j % 2 == 0 else ws // 2 for j in range(depth)]
        if res <= ws:
            ws, shifts = res, [0] * depth
        out.append((cfg["embed"] * 2 ** i, (res, res), depth, heads, ws, shifts))
    return out


def swin_forward(params, x, cfg, task):
    """swin_RKR.py:656-676 (forward_features + head), dropout / stochastic depth off.  RG_FC / SFG_FC are
    the layers of the ViT RKR model (vit_oracle.rg_fc / sfg_fc follow the same lines)."""
    import torch.nn.functional as F
    from . import vit_oracle as O
    eps = 1e-5                                            # nn.LayerNorm default, swin_RKR.py:578
    ln = lambda t, p: F.layer_norm(t, (t.shape[-1],), params[p + "weight"], params[p + "bias"], eps)  # noqa: E731
    # PatchEmbed :514-522
    t = F.conv2d(x, params["patch_embed.proj.weight"], params["patch_embed.proj.bias"], stride=cfg["patch"])
    t = t.flatten(2).transpose(1, 2)
    if "patch_embed.norm.weight" in params:
        t = ln(t, "patch_embed.norm.")
    if "absolute_pos_embed" in params:
        t = t + params["absolute_pos_embed"]
    B = t.shape[0]
    geo = swin_geometry(cfg)
    for i, (dim, (H, W), depth, heads, ws, shifts) in enumerate(geo):
        for j in range(depth):
            p = f"layers.{i}.blocks.{j}."
            shortcut = t
            y = ln(t, p + "norm1.")                                                      # :312
            qkv = O.rg_fc(params, y.reshape(B * H * W, dim), p + "attn.qkv.", task)      # :185
            a = shifted_window_attention(qkv, params[p + "attn.relative_position_bias_table"], B, H, W, heads, ws,
                                         shifts[j], (dim // heads) ** -0.5)             # :313-337 around :186-209
            a = O.sfg_fc(params, O.rg_fc(params, a, p + "attn.proj.", task), p + "attn.sfg_proj.", task)   # :211-213
            t = shortcut + a.view(B, H * W, dim)                                         # :340
            y = ln(t, p + "norm2.")
            hdn = F.gelu(O.sfg_fc(params, O.rg_fc(params, y, p + "mlp.fc1.", task), p + "mlp.sfg1.", task))   # :78-81
            t = t + O.sfg_fc(params, O.rg_fc(params, hdn, p + "mlp.fc2.", task), p + "mlp.sfg2.", task)      # :83-85, 341
        if i < len(geo) - 1:                                                             # PatchMerging :389-399
            v = t.view(B, H, W, dim)
            v = torch.cat([v[:, 0::2, 0::2, :], v[:, 1::2, 0::2, :], v[:, 0::2, 1::2, :], v[:, 1::2, 1::2, :]], -1)
            v = ln(v.view(B, -1, 4 * dim), f"layers.{i}.downsample.norm.")
            t = F.linear(v, params[f"layers.{i}.downsample.reduction.weight"])
    t = ln(t, "norm.").mean(dim=1)                                                       # :665-667
    hw = params.get(f"head.{task}.weight")
    if hw is not None:
        return F.linear(t, hw, params[f"head.{task}.bias"])
    return F.linear(t, params["head.weight"], params["head.bias"])


def swin_loss_and_grads(params, x, labels, cfg, task):
    import torch.nn.functional as F
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits = swin_forward(leaf, x, cfg, task)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), logits.detach(), grads


def synth_swin_params(cfg, num_classes, RG=True, SFG=True, K=2, seed=0, variant="rkr", ternary=False):
    """Parameters of a Swin model by state_dict name; `num_classes` is the per-task list (one head per
    task).  variant "rkr": swin_RKR.py:23-32, 48, 148-176, 271-279, 377-378, 508-510, 628-634;
    "pb": swin_PB.py with layers_PB.py:56-102 (`mask_reals.<t>` per linear, no scales);
    "rkrpb": swin_RKRPB.py with layers_PB.py:138-249, 306-347 in its default configuration (shared
    `LM_base` / `RM_base` / `F_base`, masks `mask_reals_LM/RM/F.<t-1>` for tasks t > 0)."""
    import numpy as np
    from .synth import _pb_mask
    rng = np.random.default_rng(seed + 86028121)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    T = len(num_classes)
    out = {}

    def norm(p, d):
        out[p + "weight"] = f32(1.0 + 0.1 * rng.standard_normal(d))
        out[p + "bias"] = f32(0.05 * rng.standard_normal(d))

    def fc(p, h_in, h_out, sfg):
        out[p + "weight"] = f32(rng.standard_normal((h_out, h_in)) / np.sqrt(h_in))
        out[p + "bias"] = f32(0.05 * rng.standard_normal(h_out))
        lm = lambda: f32(0.3 * rng.standard_normal((h_in, K)))  # noqa: E731
        rm = lambda: f32(0.47 * rng.standard_normal((K, h_out)) / np.sqrt(h_in))  # noqa: E731
        if variant == "pb":
            for t in range(T):
                out[f"{p}mask_reals.{t}"] = f32(_pb_mask(rng, (h_out, h_in), ternary))
            return
        if variant == "rkrpb":
            out[p + "LM_base"], out[p + "RM_base"] = lm(), rm()
            for t in range(T - 1):
                out[f"{p}mask_reals_LM.{t}"] = f32(_pb_mask(rng, (h_in, K), ternary))
                out[f"{p}mask_reals_RM.{t}"] = f32(_pb_mask(rng, (K, h_out), ternary))
            if SFG and sfg:
                out[sfg + "F_base"] = f32(1.0 + 0.2 * rng.standard_normal(h_out))
                for t in range(T - 1):
                    out[f"{sfg}mask_reals_F.{t}"] = f32(_pb_mask(rng, (h_out,), ternary))
            return
        for t in range(T):
            if RG:
                out[f"{p}LM_list.{t}"] = lm()
                out[f"{p}RM_list.{t}"] = rm()
            if SFG and sfg:
                out[f"{sfg}F_list.{t}"] = f32(1.0 + 0.2 * rng.standard_normal(h_out))

    E, P = cfg["embed"], cfg["patch"]
    out["patch_embed.proj.weight"] = f32(rng.standard_normal((E, 3, P, P)) / np.sqrt(3 * P * P))
    out["patch_embed.proj.bias"] = f32(0.05 * rng.standard_normal(E))
    norm("patch_embed.norm.", E)
    geo = swin_geometry(cfg)
    for i, (dim, _, depth, heads, ws, _) in enumerate(geo):
        for j in range(depth):
            p = f"layers.{i}.blocks.{j}."
            norm(p + "norm1.", dim)
            out[p + "attn.relative_position_bias_table"] = f32(0.5 * rng.standard_normal(((2 * ws - 1) ** 2, heads)))
            fc(p + "attn.qkv.", dim, 3 * dim, None)
            fc(p + "attn.proj.", dim, dim, p + "attn.sfg_proj.")
            norm(p + "norm2.", dim)
            hidden = int(dim * cfg.get("mlp_ratio", 4.0))
            fc(p + "mlp.fc1.", dim, hidden, p + "mlp.sfg1.")
            fc(p + "mlp.fc2.", hidden, dim, p + "mlp.sfg2.")
        if i < len(geo) - 1:
            out[f"layers.{i}.downsample.reduction.weight"] = f32(rng.standard_normal((2 * dim, 4 * dim)) / np.sqrt(4 * dim))
            norm(f"layers.{i}.downsample.norm.", 4 * dim)
    D = geo[-1][0]
    norm("norm.", D)
    for t, n in enumerate(num_classes):
        out[f"head.{t}.weight"] = f32(rng.standard_normal((n, D)) / np.sqrt(D))
        out[f"head.{t}.bias"] = f32(0.05 * rng.standard_normal(n))
    return out
