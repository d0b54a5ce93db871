"""Deterministic synthetic weights and batches (numpy Generator: stable across versions).

TEST INFRASTRUCTURE: shared by ``oracle/gen_golden.py`` (which feeds the same tensors to the
imported reference) and by the tests, so fixtures only need to hold expected outputs.
"""
import math

import numpy as np
import torch

GEOMS = {
    # name: (hidden, heads, mlp, layers, patch)
    "tiny": dict(hidden=128, heads=2, mlp=256, layers=2, patch=16),
    "B_16": dict(hidden=768, heads=12, mlp=3072, layers=12, patch=16),
    "L_16": dict(hidden=1024, heads=16, mlp=4096, layers=24, patch=16),
}


def param_shapes(geom, img, num_classes, in_ch=3):
    """state_dict names and shapes of VisionTransformer (reference modeling.py:51-66,98-105,
    125-151,170-176,229-236,262-268), in registration order."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    D, F_, P = g["hidden"], g["mlp"], g["patch"]
    n = (img // P) ** 2
    s = [
        ("transformer.embeddings.position_embeddings", (1, n + 1, D)),
        ("transformer.embeddings.cls_token", (1, 1, D)),
        ("transformer.embeddings.patch_embeddings.weight", (D, in_ch, P, P)),
        ("transformer.embeddings.patch_embeddings.bias", (D,)),
    ]
    for i in range(g["layers"]):
        p = f"transformer.encoder.layer.{i}."
        s += [(p + "attention_norm.weight", (D,)), (p + "attention_norm.bias", (D,)),
              (p + "ffn_norm.weight", (D,)), (p + "ffn_norm.bias", (D,)),
              (p + "ffn.fc1.weight", (F_, D)), (p + "ffn.fc1.bias", (F_,)),
              (p + "ffn.fc2.weight", (D, F_)), (p + "ffn.fc2.bias", (D,))]
        for nm in ("query", "key", "value", "out"):
            s += [(p + f"attn.{nm}.weight", (D, D)), (p + f"attn.{nm}.bias", (D,))]
    s += [("transformer.encoder.encoder_norm.weight", (D,)),
          ("transformer.encoder.encoder_norm.bias", (D,)),
          ("head.weight", (num_classes, D)), ("head.bias", (num_classes,))]
    return s


def synth_params(geom, img, num_classes, seed=0, in_ch=3):
    """Random but well-conditioned parameters: matrices ~ N(0, 1/fan_in), LN scales ~ 1 +- 0.1,
    biases / cls / position embeddings ~ N(0, 0.02..0.1) so every term is exercised."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes(geom, img, num_classes, in_ch):
        if name.endswith("norm.weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.05 * rng.standard_normal(shape)
        elif name.endswith("position_embeddings") or name.endswith("cls_token"):
            a = 0.1 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) / np.sqrt(fan_in)
        out[name] = torch.from_numpy(a.astype(np.float32))
    return out


def resnet_param_shapes(block_units, width_factor, prefix="transformer.embeddings.hybrid_model."):
    """state_dict names and shapes of ResNetV2 (reference modeling_resnet.py:55-74,129-160)."""
    w = int(64 * width_factor)
    s = [(prefix + "root.conv.weight", (w, 3, 7, 7)), (prefix + "root.gn.weight", (w,)), (prefix + "root.gn.bias", (w,))]
    stages = [(w, 4 * w, w, 1), (4 * w, 8 * w, 2 * w, 2), (8 * w, 16 * w, 4 * w, 2)]
    for b, ((cin, cout, cmid, stride), units) in enumerate(zip(stages, block_units), start=1):
        for u in range(1, units + 1):
            p = f"{prefix}body.block{b}.unit{u}."
            ci = cin if u == 1 else cout
            s += [(p + "gn1.weight", (cmid,)), (p + "gn1.bias", (cmid,)), (p + "conv1.weight", (cmid, ci, 1, 1)),
                  (p + "gn2.weight", (cmid,)), (p + "gn2.bias", (cmid,)), (p + "conv2.weight", (cmid, cmid, 3, 3)),
                  (p + "gn3.weight", (cout,)), (p + "gn3.bias", (cout,)), (p + "conv3.weight", (cout, cmid, 1, 1))]
            if u == 1 and (stride != 1 or ci != cout):
                s += [(p + "downsample.weight", (cout, ci, 1, 1)),
                      (p + "gn_proj.weight", (cout,)), (p + "gn_proj.bias", (cout,))]
    return s


def synth_hybrid_params(geom, img, num_classes, block_units, width_factor, seed=0):
    """Parameters of the R50-style hybrid (reference modeling.py:128-147 with modeling_resnet.py): the
    transformer of `geom` on the 1/16-resolution feature map of a ResNetV2 stem, patch projection with
    kernel 1 (grid = img/16)."""
    g = dict(GEOMS[geom] if isinstance(geom, str) else geom)
    n = (img // 16) ** 2
    out = synth_params(dict(g, patch=16), img, num_classes, seed=seed)
    rng = np.random.default_rng(seed + 15485863)
    cin = int(64 * width_factor) * 16
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    out["transformer.embeddings.patch_embeddings.weight"] = f32(rng.standard_normal((g["hidden"], cin, 1, 1)) / np.sqrt(cin))
    assert out["transformer.embeddings.position_embeddings"].shape[1] == n + 1
    for name, shape in resnet_param_shapes(block_units, width_factor):
        if name.endswith("gn.weight") or ".gn" in name and name.endswith(".weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.05 * rng.standard_normal(shape)
        else:
            a = rng.standard_normal(shape) / np.sqrt(int(np.prod(shape[1:])))
        out[name] = f32(a)
    return out


def synth_rkr_params(geom, img, num_classes, K=2, RG=True, SFG=True, seed=0, in_ch=3):
    """Parameters of the RKR model (reference models/ViT/modeling_RKR.py:35-78,99-128,163-182,338-351)
    by state_dict name; `num_classes` is the per-task list.  The rank-K factors are scaled so that
    (LM.RM)^T is about a fifth of W in magnitude (the reference initialises it near 2 %), and the SFG
    scales are 1 +- 0.2, so both matter in every check."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    D, F_ = g["hidden"], g["mlp"]
    T = len(num_classes)
    base = synth_params(geom, img, num_classes[0], seed=seed, in_ch=in_ch)
    rng = np.random.default_rng(seed + 7919)
    out = {k: v for k, v in base.items() if not k.startswith("head.")}
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    for i in range(g["layers"]):
        p = f"transformer.encoder.layer.{i}."
        fcs = [(p + f"attn.{nm}.", D, D, p + f"attn.SFG_{nm}.") for nm in ("query", "key", "value", "out")]
        fcs += [(p + "ffn.fc1.", D, F_, p + "ffn.SFG1."), (p + "ffn.fc2.", F_, D, p + "ffn.SFG2.")]
        for fc, h_in, h_out, sfg in fcs:
            for t in range(T):
                if RG:
                    out[f"{fc}LM_list.{t}"] = f32(0.3 * rng.standard_normal((h_in, K)))
                    out[f"{fc}RM_list.{t}"] = f32(0.47 * rng.standard_normal((K, h_out)) / np.sqrt(h_in))
                if SFG:
                    out[f"{sfg}F_list.{t}"] = f32(1.0 + 0.2 * rng.standard_normal(h_out))
    for t, n in enumerate(num_classes):
        out[f"head.{t}.weight"] = f32(rng.standard_normal((n, D)) / np.sqrt(D))
        out[f"head.{t}.bias"] = f32(0.05 * rng.standard_normal(n))
    return out


def synth_pb_params(geom, img, num_classes, seed=0, in_ch=3, ternary=False):
    """Parameters of the piggyback model (reference models/ViT/modeling_PB.py:68-71,111-112 with
    layers_PB.py:56-102) by state_dict name.  Real-valued masks are spread around the 5e-3 threshold
    (about 60 % of the entries above it; for the ternarizer a quarter below zero), with no entry
    closer to a decision boundary than 2e-4, so the hard mask does not depend on rounding."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    D, F_ = g["hidden"], g["mlp"]
    base = synth_params(geom, img, num_classes[0], seed=seed, in_ch=in_ch)
    rng = np.random.default_rng(seed + 104729)
    out = {k: v for k, v in base.items() if not k.startswith("head.")}
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    for i in range(g["layers"]):
        p = f"transformer.encoder.layer.{i}."
        fcs = [(p + f"attn.{nm}.", D, D) for nm in ("query", "key", "value", "out")]
        fcs += [(p + "ffn.fc1.", D, F_), (p + "ffn.fc2.", F_, D)]
        for fc, h_in, h_out in fcs:
            for t in range(len(num_classes)):
                m = rng.uniform(-0.006 if ternary else -0.002, 0.015, size=(h_out, h_in))
                for edge in ((5e-3, 0.0) if ternary else (5e-3,)):
                    near = np.abs(m - edge) < 2e-4
                    m[near] = edge + np.where(m[near] >= edge, 2e-4, -2e-4)
                out[f"{fc}mask_reals.{t}"] = f32(m)
    for t, n in enumerate(num_classes):
        out[f"head.{t}.weight"] = f32(rng.standard_normal((n, D)) / np.sqrt(D))
        out[f"head.{t}.bias"] = f32(0.05 * rng.standard_normal(n))
    return out


def _pb_mask(rng, shape, ternary=False):
    m = rng.uniform(-0.006 if ternary else -0.002, 0.015, size=shape)
    for edge in ((5e-3, 0.0) if ternary else (5e-3,)):
        near = np.abs(m - edge) < 2e-4
        m[near] = edge + np.where(m[near] >= edge, 2e-4, -2e-4)
    return m


def synth_rkrpb_params(geom, img, num_classes, K=2, seed=0, in_ch=3, ternary=False, SFG=True, RGnoPB=False,
                       LMnoPB=False, RMnoPB=False, PBwR=False, SFGnoPB=False):
    """Parameters of the RKR+piggyback model (reference models/ViT/modeling_RKRPB.py:68-77,125-130 with
    layers_PB.py:138-249,306-347) by state_dict name, for the given configuration flags.  Rank-K
    factors are sized as in synth_rkr_params, masks as in synth_pb_params (task t > 0 owns mask
    t - 1), SFG vectors are 1 +- 0.2."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    D, F_ = g["hidden"], g["mlp"]
    T = len(num_classes)
    base = synth_params(geom, img, num_classes[0], seed=seed, in_ch=in_ch)
    rng = np.random.default_rng(seed + 15485863)
    out = {k: v for k, v in base.items() if not k.startswith("head.")}
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    for i in range(g["layers"]):
        p = f"transformer.encoder.layer.{i}."
        fcs = [(p + f"attn.{nm}.", D, D, p + f"attn.SFG_{nm}.") for nm in ("query", "key", "value", "out")]
        fcs += [(p + "ffn.fc1.", D, F_, p + "ffn.SFG1."), (p + "ffn.fc2.", F_, D, p + "ffn.SFG2.")]
        for fc, h_in, h_out, sfg in fcs:
            lm = lambda: f32(0.3 * rng.standard_normal((h_in, K)))  # noqa: E731
            rm = lambda: f32(0.47 * rng.standard_normal((K, h_out)) / np.sqrt(h_in))  # noqa: E731
            masks = lambda name, shape: out.update(  # noqa: E731
                {f"{fc}{name}.{t}": f32(_pb_mask(rng, shape, ternary)) for t in range(T - 1)})
            if RGnoPB:                                       # same branch order as layers_PB.py:181-249
                out.update({f"{fc}LM_list.{t}": lm() for t in range(T)})
                out.update({f"{fc}RM_list.{t}": rm() for t in range(T)})
            elif LMnoPB:
                out.update({f"{fc}LM_list.{t}": lm() for t in range(T)})
                out[fc + "RM_base"] = rm()
                masks("mask_reals_RM", (K, h_out))
            elif RMnoPB:
                out.update({f"{fc}RM_list.{t}": rm() for t in range(T)})
                out[fc + "LM_base"] = lm()
                masks("mask_reals_LM", (h_in, K))
            else:
                out[fc + "LM_base"], out[fc + "RM_base"] = lm(), rm()
                if PBwR:
                    masks("mask_reals", (h_in, h_out))
                else:
                    masks("mask_reals_LM", (h_in, K))
                    masks("mask_reals_RM", (K, h_out))
            if SFG:
                out[sfg + "F_base"] = f32(1.0 + 0.2 * rng.standard_normal(h_out))
                for t in range(T if SFGnoPB else T - 1):
                    if SFGnoPB:
                        out[f"{sfg}F_list.{t}"] = f32(1.0 + 0.2 * rng.standard_normal(h_out))
                    else:
                        out[f"{sfg}mask_reals_F.{t}"] = f32(_pb_mask(rng, (h_out,), ternary))
    for t, n in enumerate(num_classes):
        out[f"head.{t}.weight"] = f32(rng.standard_normal((n, D)) / np.sqrt(D))
        out[f"head.{t}.bias"] = f32(0.05 * rng.standard_normal(n))
    return out


def synth_rkrtsn_params(geom, img, num_classes, K=2, SFG=True, task_emb_d=6, seed=0, in_ch=3):
    """Parameters of the RKR-TSN model (reference models/ViT/modeling_RKRTSN.py:39-98,134-143,192-199,
    367-374) by state_dict name.  The generated factors come out at the magnitudes of
    synth_rkr_params (their biases carry most of it, the embedding-dependent part about a third)."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    D, F_ = g["hidden"], g["mlp"]
    T = len(num_classes)
    base = synth_params(geom, img, num_classes[0], seed=seed, in_ch=in_ch)
    rng = np.random.default_rng(seed + 32452843)
    out = {k: v for k, v in base.items() if not k.startswith("head.")}
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    for i in range(8):
        n_out = T if i == 7 else task_emb_d
        out[f"task_embedding.fc.{i}.weight"] = f32(rng.standard_normal((n_out, task_emb_d)) / np.sqrt(task_emb_d))
        out[f"task_embedding.fc.{i}.bias"] = f32(0.1 * rng.standard_normal(n_out))
    for i in range(g["layers"]):
        p = f"transformer.encoder.layer.{i}."
        fcs = [(p + f"attn.{nm}.", D, D, p + f"attn.SFG_{nm}.") for nm in ("query", "key", "value", "out")]
        fcs += [(p + "ffn.fc1.", D, F_, p + "ffn.SFG1."), (p + "ffn.fc2.", F_, D, p + "ffn.SFG2.")]
        for fc, h_in, h_out, sfg in fcs:
            out[fc + "LM_fc.weight"] = f32(0.1 * rng.standard_normal((h_in * K, T)))
            out[fc + "LM_fc.bias"] = f32(0.3 * rng.standard_normal(h_in * K))
            out[fc + "RM_fc.weight"] = f32(0.15 * rng.standard_normal((h_out * K, T)) / np.sqrt(h_in))
            out[fc + "RM_fc.bias"] = f32(0.47 * rng.standard_normal(h_out * K) / np.sqrt(h_in))
            if SFG:
                for t in range(T):
                    out[f"{sfg}F_list.{t}"] = f32(1.0 + 0.2 * rng.standard_normal(h_out))
    for t, n in enumerate(num_classes):
        out[f"head.{t}.weight"] = f32(rng.standard_normal((n, D)) / np.sqrt(D))
        out[f"head.{t}.bias"] = f32(0.05 * rng.standard_normal(n))
    return out


def synth_task_vecs(num_tasks, task_emb_d=6, seed=0):
    return torch.from_numpy(np.random.default_rng(3000 + seed).standard_normal((num_tasks, task_emb_d)).astype(np.float32))


def tsn_drift(num_tasks):
    """What the golden run adds to ``task_embedding.fc.7.bias`` between task 0 and task 1, so that the
    recorded embedding of task 0 differs from its current value and the distillation term is live."""
    return 0.05 * torch.arange(1, num_tasks + 1, dtype=torch.float32) * torch.tensor([1.0, -1.0] * num_tasks)[:num_tasks]


# Golden cases of the RKR+piggyback model, tests/golden/rkrpb_tiny.npz
RKRPB_CASES = [   # (name, configuration flags, threshold_fn, tasks, seed)
    ("default", dict(), "binarizer", (0, 2), 41),
    ("ternary", dict(), "ternarizer", (1,), 43),
    ("pbwr", dict(PBwR=True), "binarizer", (1,), 47),
    ("lmnopb", dict(LMnoPB=True), "binarizer", (0, 1), 53),
    ("rmnopb", dict(RMnoPB=True), "binarizer", (2,), 59),
    ("nopb", dict(RGnoPB=True, SFGnoPB=True), "binarizer", (1,), 61),
]


def synth_batch(batch, img, num_classes, seed=0, in_ch=3):
    x = np.random.default_rng(1000 + seed).standard_normal((batch, in_ch, img, img)).astype(np.float32)
    y = np.random.default_rng(2000 + seed).integers(0, num_classes, batch).astype(np.int64)
    return torch.from_numpy(x), torch.from_numpy(y)


def oracle_cfg(geom):
    g = GEOMS[geom] if isinstance(geom, str) else geom
    return dict(patch=g["patch"], heads=g["heads"], layers=g["layers"])


def subsample(t, n=4096, seed=7):
    """Fixed pseudo-random subsample of a tensor's flat view (indices depend only on numel)."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= n:
        return np.arange(flat.numel()), flat.cpu().numpy().copy()
    idx = np.sort(np.random.default_rng(seed).choice(flat.numel(), n, replace=False))
    return idx, flat[torch.from_numpy(idx)].cpu().numpy().copy()


# ----------------------------------------------------------------------------- Piggyback-GAN style variants
def _pbg_sizes(h_out, lamb, task):
    """(unconstrained filters, piggyback filters, filter-bank rows) of one layer (modeling_PBG.py:47-58)."""
    lamb_num = math.ceil(lamb * h_out)
    return lamb_num, h_out - lamb_num, h_out + (task - 1) * lamb_num


def _fc_list(g):
    D, F_ = g["hidden"], g["mlp"]
    out = []
    for i in range(g["layers"]):
        p = f"transformer.encoder.layer.{i}."
        out += [(p + f"attn.{nm}.", D, D) for nm in ("query", "key", "value", "out")]
        out += [(p + "ffn.fc1.", D, F_), (p + "ffn.fc2.", F_, D)]
    return out


def synth_pbg_params(geom, img, num_classes, task, lamb, seed=0, in_ch=3):
    """Parameters and buffers of the filter-bank model (reference models/ViT/modeling_PBG.py:36-74) built for
    ``task`` by state_dict name: task 0 holds full ``unc_filt`` matrices; task t > 0 holds ceil(lamb * out)
    unconstrained filters, the mixing matrix ``weights_mat`` and the bank ``concat_unc_filter`` of earlier tasks'
    filters (a buffer).  One head."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    base = synth_params(geom, img, num_classes, seed=seed, in_ch=in_ch)
    rng = np.random.default_rng(seed + 15485863)
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    out = {k: v for k, v in base.items() if ".attn." not in k and ".ffn." not in k}
    for fc, h_in, h_out in _fc_list(g):
        w = base[fc + "weight"]
        out[fc + "bias"] = base[fc + "bias"]
        if task == 0:
            out[fc + "unc_filt"] = w
            continue
        lamb_num, rem, bank = _pbg_sizes(h_out, lamb, task)
        out[fc + "unc_filt"] = w[:lamb_num].clone()
        out[fc + "weights_mat"] = f32(rng.standard_normal((bank, rem)) / np.sqrt(bank))
        out[fc + "concat_unc_filter"] = f32(rng.standard_normal((bank, h_in)) * float(w.std()))
    return out


def synth_rkrpbg_params(geom, img, num_classes, task, lamb, K=2, seed=0, in_ch=3, SFG=True):
    """Parameters and buffers of the rank-K filter-bank model (reference models/ViT/modeling_RKRPBG.py:40-138,
    141-152) built for ``task``: a full weight per layer plus rank-K factors of the unconstrained filters and,
    for task t > 0 with lamb < 1, of the mixing matrix; the bank of earlier filters is a buffer."""
    g = GEOMS[geom] if isinstance(geom, str) else geom
    base = synth_params(geom, img, num_classes, seed=seed, in_ch=in_ch)
    rng = np.random.default_rng(seed + 32452843)
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    out = dict(base)
    for fc, h_in, h_out in _fc_list(g):
        ws = float(base[fc + "weight"].std())
        lamb_num, rem, bank = _pbg_sizes(h_out, lamb, task)
        full = task == 0 or lamb == 1.0
        rows = h_out if full else lamb_num
        out[fc + "unc_filt_LM"] = f32(rng.standard_normal((rows, K)) * np.sqrt(ws / np.sqrt(K)))
        out[fc + "unc_filt_RM"] = f32(rng.standard_normal((K, h_in)) * np.sqrt(ws / np.sqrt(K)))
        if not full:
            out[fc + "weights_mat_LM"] = f32(rng.standard_normal((bank, K)) / np.sqrt(np.sqrt(K * bank)))
            out[fc + "weights_mat_RM"] = f32(rng.standard_normal((K, rem)) / np.sqrt(np.sqrt(K * bank)))
            out[fc + "concat_unc_filter"] = f32(rng.standard_normal((bank, h_in)) * ws)
    if SFG:
        D, F_ = g["hidden"], g["mlp"]
        for i in range(g["layers"]):
            p = f"transformer.encoder.layer.{i}."
            for nm, n in (("attn.SFG_query", D), ("attn.SFG_key", D), ("attn.SFG_value", D), ("attn.SFG_out", D),
                          ("ffn.SFG1", F_), ("ffn.SFG2", D)):
                out[f"{p}{nm}.SFG_F"] = f32(1.0 + 0.2 * rng.standard_normal(n))
    return out


# (name, task the model is built for, lamb)
PBG_CASES = [("t0", 0, 0.5), ("t2", 2, 0.3), ("t1half", 1, 0.5)]
RKRPBG_CASES = [("t0", 0, 0.5), ("t1", 1, 0.3), ("t2full", 2, 1.0)]
