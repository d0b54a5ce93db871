"""CPU oracle for the ViT forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

A from-scratch restatement, in plain fp32 PyTorch CPU ops over a flat ``{name: tensor}``
parameter dictionary, of the arithmetic the reference performs in
``/root/reference/models/ViT/modeling.py``.  The reference's arithmetic lives in the
third-party ``torch`` package (``requirements.txt:1`` pins no version; 2.10.0 here), so
the restatement uses the same primitives (``F.linear``, ``F.layer_norm``, ``F.gelu``,
``softmax``, ``matmul``) and ``autograd`` for the backward pass.

Pinning: the reference ships no tests or golden vectors.  The oracle is pinned against
the reference module itself, imported in the build container by ``oracle/gen_golden.py``;
the resulting vectors are committed under ``tests/golden`` and re-checked by
``tests/test_oracle_golden.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module; the product (``vit-pytorch_amd``) never does.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # modeling.py:173-174,233


def _p(params, prefix, name):
    return params[prefix + name]


# ----------------------------------------------------------------------------- modules
def embeddings_forward(params, x, patch, prefix="transformer.embeddings."):
    """modeling.py:153-166 (non-hybrid branch): strided conv == per-patch linear map,
    flatten(2).transpose, prepend cls, add position embeddings (dropout handled by caller)."""
    w = _p(params, prefix, "patch_embeddings.weight")          # [D, C, P, P]
    b = _p(params, prefix, "patch_embeddings.bias")
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    # im2col: [B, gh, gw, C, P, P] -> [B*gh*gw, C*P*P]
    cols = (x[:, :, :gh * patch, :gw * patch]
            .reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
            .reshape(B * gh * gw, C * patch * patch))
    tok = F.linear(cols, w.reshape(w.shape[0], -1), b).reshape(B, gh * gw, -1)
    cls = _p(params, prefix, "cls_token").expand(B, -1, -1)
    return torch.cat((cls, tok), dim=1) + _p(params, prefix, "position_embeddings")


def _std_conv(params, x, name, stride, padding):
    """modeling_resnet.py:35-42 StdConv2d: weight standardised per output channel (biased variance,
    eps 1e-5), no bias in the stem."""
    w = params[name + ".weight"]
    v, m = torch.var_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
    return F.conv2d(x, (w - m) / torch.sqrt(v + 1e-5), None, stride, padding)


def _gn(params, x, name, groups, eps):
    return F.group_norm(x, groups, params[name + ".weight"], params[name + ".bias"], eps)


def bottleneck_forward(params, x, prefix, stride):
    """modeling_resnet.py:76-91 PreActBottleneck.forward.  The projection shortcut exists when the
    unit's parameters hold one; its GroupNorm has one group per channel and the default eps 1e-5."""
    residual = x
    if prefix + "downsample.weight" in params:
        residual = _std_conv(params, x, prefix + "downsample", stride, 0)
        residual = _gn(params, residual, prefix + "gn_proj", residual.shape[1], 1e-5)
    y = F.relu(_gn(params, _std_conv(params, x, prefix + "conv1", 1, 0), prefix + "gn1", 32, 1e-6))
    y = F.relu(_gn(params, _std_conv(params, y, prefix + "conv2", stride, 1), prefix + "gn2", 32, 1e-6))
    y = _gn(params, _std_conv(params, y, prefix + "conv3", 1, 0), prefix + "gn3", 32, 1e-6)
    return F.relu(residual + y)


def resnetv2_forward(params, x, prefix):
    """modeling_resnet.py:136-164 ResNetV2: root conv 7x7/2 pad 3, GroupNorm(32, eps 1e-6), ReLU,
    max-pool 3x3/2 without padding; three stages, the first unit of stages 2 and 3 has stride 2."""
    x = _std_conv(params, x, prefix + "root.conv", 2, 3)
    x = F.max_pool2d(F.relu(_gn(params, x, prefix + "root.gn", 32, 1e-6)), 3, 2, 0)
    for b in (1, 2, 3):
        u = 1
        while f"{prefix}body.block{b}.unit{u}.conv1.weight" in params:
            x = bottleneck_forward(params, x, f"{prefix}body.block{b}.unit{u}.", 2 if (b > 1 and u == 1) else 1)
            u += 1
    return x


def hybrid_embeddings_forward(params, x, prefix="transformer.embeddings."):
    """modeling.py:153-166, hybrid branch: ResNetV2 feature map, patch convolution with kernel =
    stride = its own kernel size, flatten(2).transpose, prepend cls, add position embeddings."""
    feat = resnetv2_forward(params, x, prefix + "hybrid_model.")
    w = _p(params, prefix, "patch_embeddings.weight")
    tok = F.conv2d(feat, w, _p(params, prefix, "patch_embeddings.bias"), stride=w.shape[2:])
    tok = tok.flatten(2).transpose(-1, -2)
    cls = _p(params, prefix, "cls_token").expand(x.shape[0], -1, -1)
    return torch.cat((cls, tok), dim=1) + _p(params, prefix, "position_embeddings")


def attention_forward(params, x, num_heads, prefix, vis=False):
    """modeling.py:73-94.  attention_dropout_rate is 0.0 in every preset."""
    B, N, D = x.shape
    dh = D // num_heads

    def heads(t):  # transpose_for_scores, modeling.py:68-71
        return t.view(B, N, num_heads, dh).permute(0, 2, 1, 3)

    q = heads(F.linear(x, _p(params, prefix, "query.weight"), _p(params, prefix, "query.bias")))
    k = heads(F.linear(x, _p(params, prefix, "key.weight"), _p(params, prefix, "key.bias")))
    v = heads(F.linear(x, _p(params, prefix, "value.weight"), _p(params, prefix, "value.bias")))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, N, D)
    out = F.linear(ctx, _p(params, prefix, "out.weight"), _p(params, prefix, "out.bias"))
    return out, (probs if vis else None)


def mlp_forward(params, x, prefix, drop=0.0):
    """modeling.py:113-119: fc1, exact (erf) GELU (modeling.py:47), dropout, fc2, dropout.  `drop` is the
    training-mode dropout rate (0 = disabled, what every parity comparison uses; torch's own mask stream)."""
    h = F.gelu(F.linear(x, _p(params, prefix, "fc1.weight"), _p(params, prefix, "fc1.bias")))
    if drop > 0:
        h = F.dropout(h, drop, True)
    y = F.linear(h, _p(params, prefix, "fc2.weight"), _p(params, prefix, "fc2.bias"))
    return F.dropout(y, drop, True) if drop > 0 else y


def block_forward(params, x, num_heads, prefix, vis=False, drop=0.0):
    """modeling.py:178-188: pre-LN residual block."""
    D = x.shape[-1]
    h = x
    y = F.layer_norm(x, (D,), _p(params, prefix, "attention_norm.weight"),
                     _p(params, prefix, "attention_norm.bias"), LN_EPS)
    y, w = attention_forward(params, y, num_heads, prefix + "attn.", vis)
    x = y + h
    h = x
    y = F.layer_norm(x, (D,), _p(params, prefix, "ffn_norm.weight"),
                     _p(params, prefix, "ffn_norm.bias"), LN_EPS)
    y = mlp_forward(params, y, prefix + "ffn.", drop)
    return y + h, w


def encoder_forward(params, x, num_heads, num_layers, prefix="transformer.encoder.", vis=False, drop=0.0):
    """modeling.py:238-245."""
    ws = []
    for i in range(num_layers):
        x, w = block_forward(params, x, num_heads, f"{prefix}layer.{i}.", vis, drop)
        if vis:
            ws.append(w)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), _p(params, prefix, "encoder_norm.weight"),
                     _p(params, prefix, "encoder_norm.bias"), LN_EPS)
    return x, ws


def vit_forward(params, x, cfg, labels=None, vis=False):
    """modeling.py:254-257, 270-279: embeddings -> encoder -> head(x[:,0]) -> mean CE."""
    patch = cfg["patch"]
    if "transformer.embeddings.hybrid_model.root.conv.weight" in params:
        e = hybrid_embeddings_forward(params, x)
    else:
        e = embeddings_forward(params, x, patch)
    drop = float(cfg.get("dropout", 0.0))           # training-mode dropout (modeling.py:165,116,118); 0 = off
    if drop > 0:
        e = F.dropout(e, drop, True)
    enc, ws = encoder_forward(params, e, cfg["heads"], cfg["layers"], vis=vis, drop=drop)
    logits = F.linear(enc[:, 0], params["head.weight"], params["head.bias"])
    if labels is not None:
        return F.cross_entropy(logits, labels)
    return logits, ws


def vit_loss_and_grads(params, x, labels, cfg):
    """One training-mode (dropout disabled) forward + backward; returns loss, logits, grads."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits, _ = vit_forward(leaf, x, cfg)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), logits.detach(), grads


# ----------------------------------------------------------------------------- RKR variant
PB_THRESHOLD = 5e-3              # layers_PB.py:11 DEFAULT_THRESHOLD (the only value the reference ever applies)
PB_THRESHOLD_FN = "binarizer"    # config.threshold_fn of the piggyback model under test: "binarizer" | "ternarizer"


def pb_threshold(mask):
    """layers_PB.py:14-54 Binarizer / Ternarizer: hard {0,1} or {-1,0,1} mask in the forward pass, the
    gradient passed through unchanged in the backward pass."""
    hard = (mask > PB_THRESHOLD).to(mask.dtype)
    if PB_THRESHOLD_FN == "ternarizer":
        hard = hard - (mask < 0).to(mask.dtype)
    return mask + (hard - mask).detach()


TASK_EMB = "@task_emb_vec"      # where tsn_forward puts the current task's embedding for rg_fc


def rg_fc(params, x, prefix, task):
    """The task-conditioned linear layer, whichever per-task parameters the dictionary holds for it:
    modeling_RKR.py:56-65 RG_FC.forward: weight = (LM[task] @ RM[task]).permute(1, 0) + W;
    layers_PB.py:104-111 ElementWiseLinear.forward: weight = threshold(mask_reals[task]) * W;
    layers_PB.py:251-302 ElementWiseRG.forward: weight = R.permute(1, 0) + W with R from shared
    factors LM_base / RM_base that task 0 uses as they are and task t > 0 masks with mask t - 1
    (both factors; only the one that has no per-task list under LMnoPB / RMnoPB; the product under PBwR)."""
    if prefix + "unc_filt" in params:               # modeling_PBG.py:60-74 PiggybackFC.forward
        w = params[prefix + "unc_filt"]
        if prefix + "weights_mat" in params:
            bank = params[prefix + "concat_unc_filter"]
            mixed = torch.matmul(torch.reshape(bank, (bank.shape[1], bank.shape[0])), params[prefix + "weights_mat"])
            w = torch.cat((w, torch.reshape(mixed, (mixed.shape[1], bank.shape[1]))), dim=0)
        return F.linear(x, w, params[prefix + "bias"])
    w = params[prefix + "weight"]
    if prefix + "unc_filt_LM" in params:            # modeling_RKRPBG.py:102-138 RG_FC.forward
        R = torch.matmul(params[prefix + "unc_filt_LM"], params[prefix + "unc_filt_RM"])
        if prefix + "weights_mat_LM" in params:
            bank = params[prefix + "concat_unc_filter"]
            mat = torch.matmul(params[prefix + "weights_mat_LM"], params[prefix + "weights_mat_RM"])
            mixed = torch.matmul(torch.reshape(bank, (w.shape[1], bank.shape[0])), mat)
            R = torch.cat((R, torch.reshape(mixed, (mixed.shape[1], w.shape[1]))), dim=0)
        return F.linear(x, R + w, params[prefix + "bias"])
    if prefix + "LM_fc.weight" in params:          # modeling_RKRTSN.py:76-85: factors generated from the task embedding
        emb, K = params[TASK_EMB], params[prefix + "LM_fc.bias"].numel() // w.shape[1]
        lm = F.linear(emb, params[prefix + "LM_fc.weight"], params[prefix + "LM_fc.bias"]).reshape(w.shape[1], K)
        rm = F.linear(emb, params[prefix + "RM_fc.weight"], params[prefix + "RM_fc.bias"]).reshape(K, w.shape[0])
        return F.linear(x, torch.matmul(lm, rm).permute(1, 0) + w, params[prefix + "bias"])
    lm, rm = params.get(f"{prefix}LM_list.{task}"), params.get(f"{prefix}RM_list.{task}")
    lm_base, rm_base = params.get(prefix + "LM_base"), params.get(prefix + "RM_base")
    if lm_base is not None or rm_base is not None:                     # ElementWiseRG
        if lm is None:
            lm = lm_base
            if task > 0 and f"{prefix}mask_reals_LM.{task - 1}" in params:
                lm = pb_threshold(params[f"{prefix}mask_reals_LM.{task - 1}"]) * lm_base
        if rm is None:
            rm = rm_base
            if task > 0 and f"{prefix}mask_reals_RM.{task - 1}" in params:
                rm = pb_threshold(params[f"{prefix}mask_reals_RM.{task - 1}"]) * rm_base
        R = torch.matmul(lm, rm)
        if task > 0 and f"{prefix}mask_reals.{task - 1}" in params:     # PBwR: the product is masked
            R = pb_threshold(params[f"{prefix}mask_reals.{task - 1}"]) * R
        return F.linear(x, R.permute(1, 0) + w, params[prefix + "bias"])
    if lm is not None:
        w = torch.matmul(lm, rm).permute(1, 0) + w
    mask = params.get(f"{prefix}mask_reals.{task}")
    if mask is not None:
        w = pb_threshold(mask) * w
    return F.linear(x, w, params[prefix + "bias"])


def sfg_fc(params, x, prefix, task):
    """modeling_RKR.py:73-78 SFG_FC.forward: x * F[task] broadcast over batch and tokens (no-op when
    the model was built without SFG); layers_PB.py:349-368 ElementWiseSFG.forward: F[task] when the
    per-task list exists (SFGnoPB), else F_base for task 0 and threshold(mask_F[task-1]) * F_base."""
    f = params.get(f"{prefix}F_list.{task}")
    if f is None and prefix + "SFG_F" in params:     # modeling_RKRPBG.py:141-152: one scale, whatever the task
        f = params[prefix + "SFG_F"]
    if f is None and prefix + "F_base" in params:
        f = params[prefix + "F_base"]
        if task > 0:
            f = pb_threshold(params[f"{prefix}mask_reals_F.{task - 1}"]) * f
    return x if f is None else x * f


def rkr_attention_forward(params, x, num_heads, prefix, task, vis=False):
    """modeling_RKR.py:130-160; modeling_RKRPB.py:84-113, which scales query / key / value only after
    the heads were split off the unscaled projections (:90-97), so there only SFG_out has an effect
    (recognised here by the F_base parameter of its ElementWiseSFG)."""
    B, N, D = x.shape
    dh = D // num_heads

    def heads(t):
        return t.view(B, N, num_heads, dh).permute(0, 2, 1, 3)

    def qkv_scale(t, name):
        if prefix + name + "F_base" in params:
            return t
        return sfg_fc(params, t, prefix + name, task)

    q = heads(qkv_scale(rg_fc(params, x, prefix + "query.", task), "SFG_query."))
    k = heads(qkv_scale(rg_fc(params, x, prefix + "key.", task), "SFG_key."))
    v = heads(qkv_scale(rg_fc(params, x, prefix + "value.", task), "SFG_value."))
    probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, N, D)
    out = sfg_fc(params, rg_fc(params, ctx, prefix + "out.", task), prefix + "SFG_out.", task)
    return out, (probs if vis else None)


def rkr_mlp_forward(params, x, prefix, task):
    """modeling_RKR.py:189-200 with dropout disabled."""
    h = F.gelu(sfg_fc(params, rg_fc(params, x, prefix + "fc1.", task), prefix + "SFG1.", task))
    return sfg_fc(params, rg_fc(params, h, prefix + "fc2.", task), prefix + "SFG2.", task)


def rkr_vit_forward(params, x, cfg, task, vis=False):
    """modeling_RKR.py:255-266, 316-323, 333-335, 353-370: blocks with task-conditioned linears,
    per-task head when the parameter dictionary holds ``head.<task>.weight``."""
    e = embeddings_forward(params, x, cfg["patch"])
    ws = []
    for i in range(cfg["layers"]):
        prefix = f"transformer.encoder.layer.{i}."
        D = e.shape[-1]
        h = e
        y = F.layer_norm(e, (D,), params[prefix + "attention_norm.weight"], params[prefix + "attention_norm.bias"], LN_EPS)
        y, w = rkr_attention_forward(params, y, cfg["heads"], prefix + "attn.", task, vis)
        e = y + h
        h = e
        y = F.layer_norm(e, (D,), params[prefix + "ffn_norm.weight"], params[prefix + "ffn_norm.bias"], LN_EPS)
        e = rkr_mlp_forward(params, y, prefix + "ffn.", task) + h
        if vis:
            ws.append(w)
    D = e.shape[-1]
    e = F.layer_norm(e, (D,), params["transformer.encoder.encoder_norm.weight"],
                     params["transformer.encoder.encoder_norm.bias"], LN_EPS)
    hw = params.get(f"head.{task}.weight")
    if hw is not None:
        logits = F.linear(e[:, 0], hw, params[f"head.{task}.bias"])
    else:
        logits = F.linear(e[:, 0], params["head.weight"], params["head.bias"])
    return logits, ws


def rkr_loss_and_grads(params, x, labels, cfg, task):
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits, _ = rkr_vit_forward(leaf, x, cfg, task)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), logits.detach(), grads


# ----------------------------------------------------------------------------- RKR-TSN variant
def task_embedding(params, task_vecs, prefix="task_embedding.fc."):
    """modeling_RKRTSN.py:39-57: eight linear layers, no activation between them."""
    e = task_vecs
    for i in range(8):
        e = F.linear(e, params[f"{prefix}{i}.weight"], params[f"{prefix}{i}.bias"])
    return e


def tsn_forward(params, x, cfg, task, task_vecs, vis=False):
    """modeling_RKRTSN.py:376-386: embeddings of all tasks, the current one conditions every RG_FC."""
    embs = task_embedding(params, task_vecs)
    p = dict(params)
    p[TASK_EMB] = embs[task]
    logits, ws = rkr_vit_forward(p, x, cfg, task, vis)
    return logits, ws, embs


def tsn_loss_and_grads(params, x, labels, cfg, task, task_vecs, emb_labels, lamb=1.0):
    """modeling_RKRTSN.py:393-402 and the gradients of ``loss + lamb * loss_distill``.  ``emb_labels``:
    recorded embeddings of tasks 0..task-1; the current task's target is its own detached embedding."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits, _, embs = tsn_forward(leaf, x, cfg, task, task_vecs)
    loss = F.cross_entropy(logits, labels)
    distill = torch.zeros(())
    if task != 0:
        target = torch.stack(list(emb_labels[:task]) + [embs[task].detach()], dim=0)
        distill = F.l1_loss(embs[:task + 1], target)
    (loss + lamb * distill).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), distill.detach(), logits.detach(), grads


# ----------------------------------------------------------------------------- host logic
def resize_posemb(posemb, ntok_new, classifier="token"):
    """Position-embedding grid resize of modeling.py:296-319 (scipy zoom, order=1)."""
    from scipy import ndimage
    posemb = np.asarray(posemb)
    if classifier == "token":
        tok, grid = posemb[:, :1], posemb[0, 1:]
        ntok_new -= 1
    else:
        tok, grid = posemb[:, :0], posemb[0]
    gs_old = int(np.sqrt(len(grid)))
    gs_new = int(np.sqrt(ntok_new))
    grid = grid.reshape(gs_old, gs_old, -1)
    grid = ndimage.zoom(grid, (gs_new / gs_old, gs_new / gs_old, 1), order=1)
    return np.concatenate([tok, grid.reshape(1, gs_new * gs_new, -1)], axis=1)


def warmup_cosine(step, warmup_steps, t_total, cycles=0.5):
    """utils/scheduler.py:58-63."""
    if step < warmup_steps:
        return float(step) / float(max(1.0, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, t_total - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * progress)))


def warmup_linear(step, warmup_steps, t_total):
    """utils/scheduler.py:40-43."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


def sgd_finetune(params, batches, cfg, lr=3e-2, momentum=0.9, clip=1.0, warmup=2, t_total=5):
    """train.py:475-484,520-538: SGD(momentum) + clip_grad_norm_ + warmup-cosine; returns losses."""
    params = {k: v.detach().clone() for k, v in params.items()}
    bufs = {k: None for k in params}
    losses = []
    for step, (x, y) in enumerate(batches):
        loss, _, grads = vit_loss_and_grads(params, x, y, cfg)
        losses.append(float(loss))
        total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
        coef = min(1.0, clip / (total + 1e-6))
        cur_lr = lr * warmup_cosine(step, warmup, t_total)
        for k in params:
            g = grads[k] * coef
            bufs[k] = g.clone() if bufs[k] is None else bufs[k] * momentum + g
            params[k] = params[k] - cur_lr * bufs[k]
    return losses
