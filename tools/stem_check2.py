#!/usr/bin/env python3
"""Where does the kernel-library stem's backward leave the CPU oracle's?  Runs both on the hybrid_tiny
geometry with the same parameters, input and upstream gradient, records the gradient of every operator
output in call order and prints the relative difference from the output end backwards.
usage: stem_check2.py [fp32|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops, stem_ops  # noqa: E402
from models.ViT.modeling_resnet import ResNetV2  # noqa: E402
from oracle import synth, vit_oracle as O  # noqa: E402
from oracle.gen_golden import HYBRID_TINY as H  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
vitk.set_compute_dtype(dtype)
params = synth.synth_hybrid_params("tiny", H["img"], H["ncls"], H["block_units"], H["width_factor"], seed=H["seed"])
pre = "transformer.embeddings.hybrid_model."
x, _ = synth.synth_batch(6, 64, 10, seed=int(os.environ.get("BATCH_SEED", "2473")))
torch.manual_seed(1)
go = torch.randn(6, 512, 4, 4) * 1e-2

# ---- oracle on the CPU, recording conv / gn / relu outputs in call order
rec_ref = []


def tap(name, t, nchw):
    t.retain_grad()
    rec_ref.append((name, t, nchw))
    return t


_conv0, _gn0, _relu0 = O._std_conv, O._gn, O.F.relu
O._std_conv = lambda p, xx, name, s, pad: tap("conv " + name[len(pre):], _conv0(p, xx, name, s, pad), True)
O._gn = lambda p, xx, name, g, eps: tap("gn   " + name[len(pre):], _gn0(p, xx, name, g, eps), True)


class _F:
    def __getattr__(self, k):
        if k == "relu":
            return lambda t: tap("relu", _relu0(t), True)
        return getattr(torch.nn.functional, k)


O.F = _F()
leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
out_ref = O.resnetv2_forward(leaf, x, pre)
out_ref.backward(go)

# ---- the kernel library's stem on the GPU, recording its operator outputs in call order
rec = []


def tapg(name, t, nchw):
    if t is not None and t.requires_grad:
        t.retain_grad()
        rec.append((name, t, nchw))
    return t


gn0, conv0, lin0, ar0 = stem_ops.group_norm, stem_ops.conv2d_cl, ops.linear, stem_ops.add_relu
stem_ops.group_norm = lambda *a: tapg("gn" + ("+relu" if a[5] else ""), gn0(*a), False)
def _conv(*a, **k):
    _in_conv[0] = 1
    try:
        return tapg("convKxK", conv0(*a, **k), False)
    finally:
        _in_conv[0] = 0


stem_ops.conv2d_cl = _conv
_in_conv = [0]


def _lin(*a, **k):
    y = lin0(*a, **k)
    return y if _in_conv[0] else tapg("conv1x1", y, False)


ops.linear = _lin


def ar(a, b, lp):
    o, l = ar0(a, b, lp)
    tapg("add+relu", o, False)
    return o, l


stem_ops.add_relu = ar
stem = ResNetV2(H["block_units"], H["width_factor"])
stem.load_state_dict({k[len(pre):]: v for k, v in params.items() if k.startswith(pre)})
stem = stem.cuda()
out = stem(x.cuda())
out.backward(go.cuda())

rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()  # noqa: E731
print(f"compute dtype {dtype}; feature map {rel(out.detach().cpu().float(), out_ref.detach()):.1e}")
# the oracle's sequence has separate relu entries; ours fuses them: walk both from the end
ref_ops = [(n, t) for n, t, _ in rec_ref]
ours = [(n, t, nchw) for n, t, nchw in rec]
# build comparable list from the oracle: merge "gn" followed by "relu" into one entry (the relu output)
merged = []
i = 0
while i < len(ref_ops):
    n, t = ref_ops[i]
    if n.startswith("gn") and n.endswith(("gn1", "gn2", "root.gn")) and i + 1 < len(ref_ops) and ref_ops[i + 1][0] == "relu":
        merged.append((n + " +relu", ref_ops[i + 1][1]))
        i += 2
    elif n == "relu":
        merged.append(("add +relu", t))
        i += 1
    else:
        merged.append((n, t))
        i += 1
print(len(merged), "oracle operator outputs,", len(ours), "library operator outputs")
for (rn, rt), (on, ot, nchw) in reversed(list(zip(merged, ours))):
    g = ot.grad
    if g is None:
        print(f"{rn:45s} | {on:10s} no gradient kept")
        continue
    g = g.detach().float().cpu()
    rg = rt.grad
    if g.dim() == 2:           # [B*H*W, C] token rows
        B, C, Hh, Ww = rg.shape
        g = g.view(B, Hh, Ww, C).permute(0, 3, 1, 2)
    elif not nchw:
        g = g.permute(0, 3, 1, 2)
    print(f"{rn:45s} | {on:10s} d(out) {rel(g, rg):.1e}")
worst = []
for k, p in stem.named_parameters():
    worst.append((rel(p.grad.detach().float().cpu(), leaf[pre + k].grad), k))
for e, k in worst:
    print(f"grad {k:40s} {e:.1e}")
