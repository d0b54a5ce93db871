#!/usr/bin/env python3
"""How far plain bf16 arithmetic moves the gradients of the reference computation.

Runs the CPU oracle (fp32) and two bf16 variants of the same restatement on the ViT-B/16 batch-8
fixture inputs: (a) torch.autocast(bfloat16), (b) an emulation that rounds to bf16 exactly the
tensors the HIP path stores in bf16 (optionally keeping the residual stream in fp32).
Prints per-tensor max-norm and L2 relative errors: the noise floor any bf16 path has on this input.
"""
import math
import sys
import os

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth, vit_oracle as O  # noqa: E402


class R(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


r = R.apply


class RF(torch.autograd.Function):
    """Rounds the value only (a bf16 forward stream whose gradient stays fp32)."""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class RB(torch.autograd.Function):
    """Rounds the gradient only (an fp32 forward stream with a bf16 gradient stream)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def emulated(params, x, cfg, stream_bf16, probs_bf16=True):
    # stream_bf16: True / False, or "fwd" (values only) / "bwd" (gradients only)
    s = {True: r, False: (lambda t: t), "fwd": RF.apply, "bwd": RB.apply}[stream_bf16]
    P = lambda k: params[k]       # noqa: E731
    W = lambda k: r(params[k])    # noqa: E731
    patch, H = cfg["patch"], cfg["heads"]
    B, C, Hh, _ = x.shape
    gh = Hh // patch
    cols = r(x.reshape(B, C, gh, patch, gh, patch).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gh, -1))
    pre = "transformer.embeddings."
    w = W(pre + "patch_embeddings.weight")
    tok = F.linear(cols, w.reshape(w.shape[0], -1), P(pre + "patch_embeddings.bias")).reshape(B, gh * gh, -1)
    xx = s(torch.cat((P(pre + "cls_token").expand(B, -1, -1), tok), 1) + P(pre + "position_embeddings"))
    for i in range(cfg["layers"]):
        p = f"transformer.encoder.layer.{i}."
        D = xx.shape[-1]
        dh, N = D // H, xx.shape[1]
        a = r(F.layer_norm(xx, (D,), P(p + "attention_norm.weight"), P(p + "attention_norm.bias"), 1e-6))
        hd = lambda t: t.view(B, N, H, dh).permute(0, 2, 1, 3)  # noqa: E731
        q = hd(r(F.linear(a, W(p + "attn.query.weight"), P(p + "attn.query.bias"))))
        k = hd(r(F.linear(a, W(p + "attn.key.weight"), P(p + "attn.key.bias"))))
        v = hd(r(F.linear(a, W(p + "attn.value.weight"), P(p + "attn.value.bias"))))
        pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1)
        if probs_bf16:
            pr = r(pr)
        ctx = r((pr @ v).permute(0, 2, 1, 3).reshape(B, N, D))
        xx = s(F.linear(ctx, W(p + "attn.out.weight"), P(p + "attn.out.bias")) + xx)
        c = r(F.layer_norm(xx, (D,), P(p + "ffn_norm.weight"), P(p + "ffn_norm.bias"), 1e-6))
        g = r(F.gelu(r(F.linear(c, W(p + "ffn.fc1.weight"), P(p + "ffn.fc1.bias")))))
        xx = s(F.linear(g, W(p + "ffn.fc2.weight"), P(p + "ffn.fc2.bias")) + xx)
    D = xx.shape[-1]
    enc = F.layer_norm(xx, (D,), P("transformer.encoder.encoder_norm.weight"),
                       P("transformer.encoder.encoder_norm.bias"), 1e-6)
    return F.linear(enc[:, 0], P("head.weight"), P("head.bias"))


def report(tag, leaf, grads, logits, lg):
    mx, l2 = [], []
    for k, v in leaf.items():
        if k.endswith("key.bias"):
            continue
        d = v.grad - grads[k]
        mx.append((float(d.abs().max() / grads[k].abs().max()), k))
        l2.append((float(d.norm() / grads[k].norm()), k))
    mx.sort(reverse=True)
    l2.sort(reverse=True)
    print(f"{tag}: logits {float((lg - logits).abs().max() / logits.abs().max()):.2e}")
    print("   max-norm:", ", ".join(f"{e:.2e} {k.split('layer.')[-1]}" for e, k in mx[:4]))
    print("   L2      :", ", ".join(f"{e:.2e} {k.split('layer.')[-1]}" for e, k in l2[:4]))


def main():
    geom, img, batch, ncls, seed = "B_16", 224, 8, 10, 3
    if len(sys.argv) > 1 and sys.argv[1] == "tiny":
        geom, img, batch, ncls, seed = "tiny", 48, 2, 10, 48
    params = synth.synth_params(geom, img, ncls, seed=seed)
    x, y = synth.synth_batch(batch, img, ncls, seed=seed)
    cfg = synth.oracle_cfg(geom)
    _, logits, grads = O.vit_loss_and_grads(params, x, y, cfg)
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lg, _ = O.vit_forward(leaf, x, cfg)
    F.cross_entropy(lg.float(), y).backward()
    report("torch.autocast(bf16)", leaf, grads, logits, lg.float())
    for stream, pb in ((True, True), ("fwd", True), ("bwd", True), (False, True), (False, False)):
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        lg = emulated(leaf, x, cfg, stream, pb)
        F.cross_entropy(lg, y).backward()
        report(f"emulated stream={ {True: 'bf16', False: 'fp32'}.get(stream, stream) } probs={'bf16' if pb else 'fp32'}", leaf, grads, logits, lg)


def rkr_autocast():
    """`python tools/bf16_calibration.py rkr`: torch.autocast(bfloat16) on the RKR restatement, ViT-B/16
    geometry, the inputs of tests/golden/rkr_b16_bs8.npz; worst max-norm error per parameter family."""
    ncls = [10, 6]
    params = synth.synth_rkr_params("B_16", 224, ncls, seed=23)
    x, y = synth.synth_batch(8, 224, ncls[1], seed=61)
    cfg = synth.oracle_cfg("B_16")
    _, logits, grads = O.rkr_loss_and_grads(params, x, y, cfg, 1)
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lg, _ = O.rkr_vit_forward(leaf, x, cfg, 1)
        loss = F.cross_entropy(lg.float(), y)
    loss.backward()
    fam = {}
    for k, v in leaf.items():
        if v.grad is None or k.endswith("key.bias") or float(grads[k].abs().max()) == 0.0:
            continue
        e = float((v.grad - grads[k]).abs().max() / grads[k].abs().max())
        name = "LM_list" if "LM_list" in k else "RM_list" if "RM_list" in k else "F_list" if "F_list" in k else "other"
        if e > fam.get(name, (0, ""))[0]:
            fam[name] = (e, k)
    print(f"autocast logits {float((lg.float() - logits).abs().max() / logits.abs().max()):.2e}")
    for name, (e, k) in sorted(fam.items()):
        print(f"   {name:8s} worst {e:.2e}  ({k})")


def tsn_autocast():
    """`python tools/bf16_calibration.py tsn`: torch.autocast(bfloat16) on the RKR-TSN restatement, the
    task-0 inputs of tests/golden/rkrtsn_tiny.npz; worst max-norm error per parameter family."""
    ncls = [10, 7, 5]
    for name, sfg, seed in (("sfg", True, 67), ("nosfg", False, 71)):
        params = synth.synth_rkrtsn_params("tiny", 32, ncls, SFG=sfg, seed=seed)
        cfg, tv = synth.oracle_cfg("tiny"), synth.synth_task_vecs(3, 6, seed=seed)
        x, y = synth.synth_batch(12, 32, ncls[0], seed=110)
        _, _, _, g32 = O.tsn_loss_and_grads(params, x, y, cfg, 0, tv, [])
        with torch.autocast("cpu", dtype=torch.bfloat16):
            _, _, _, g16 = O.tsn_loss_and_grads(params, x, y, cfg, 0, tv, [])
        fam = {}
        for k, v in g32.items():
            s = float(v.abs().max())
            if s == 0.0 or k.endswith("key.bias"):
                continue
            e = float((g16[k].float() - v).abs().max()) / s
            fname = next((t for t in ("LM_fc", "RM_fc", "task_embedding", "F_list") if t in k), "other")
            if e > fam.get(fname, (0, ""))[0]:
                fam[fname] = (e, k)
        print(name)
        for fname, (e, k) in sorted(fam.items()):
            print(f"   {fname:14s} worst {e:.2e}  ({k})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rkr":
        rkr_autocast()
    elif len(sys.argv) > 1 and sys.argv[1] == "tsn":
        tsn_autocast()
    else:
        main()
