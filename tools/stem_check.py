#!/usr/bin/env python3
"""Where does the hybrid stem's backward on the GPU (stock PyTorch-ROCm operators) leave the CPU's?
Runs the stem on both with identical parameters, input and upstream gradient, records the gradient
of every sub-module output, and prints them from the output end backwards."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
from models.ViT.modeling_resnet import ResNetV2  # noqa: E402
from oracle import synth  # noqa: E402
from oracle.gen_golden import HYBRID_TINY as H  # noqa: E402

inplace = "--no-inplace" not in sys.argv
if "--deterministic" in sys.argv:            # MIOpen: no algorithm search by timing, deterministic algorithms only
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
if "--no-miopen" in sys.argv:
    torch.backends.cudnn.enabled = False
params = synth.synth_hybrid_params("tiny", H["img"], H["ncls"], H["block_units"], H["width_factor"], seed=H["seed"])
pre = "transformer.embeddings.hybrid_model."
stem = ResNetV2(H["block_units"], H["width_factor"])
stem.load_state_dict({k[len(pre):]: v for k, v in params.items() if k.startswith(pre)})
if not inplace:
    for m in stem.modules():
        if isinstance(m, torch.nn.ReLU):
            m.inplace = False
x, _ = synth.synth_batch(6, 64, 10, seed=80)
torch.manual_seed(1)
go = torch.randn(6, 512, 4, 4) * 1e-2


def run(model, x, go):
    outs = {}

    def hook(name):
        def f(mod, inp, out):
            if out.requires_grad:
                out.retain_grad()
                outs[name] = out
        return f
    hs = [m.register_forward_hook(hook(n)) for n, m in model.named_modules()
          if n and not isinstance(m, (torch.nn.Sequential, torch.nn.ReLU))]
    model.zero_grad(set_to_none=True)
    o = model(x)
    o.backward(go)
    for h in hs:
        h.remove()
    act = {k: v.grad.detach().cpu() for k, v in outs.items() if v.grad is not None}
    par = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    return act, par


ca, cp = run(stem, x, go)
ga, gp = run(copy.deepcopy(stem).cuda(), x.cuda(), go.cuda())
rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()  # noqa: E731
print(f"inplace ReLU: {inplace}  flags: {sys.argv[1:]}")
for k in reversed(list(ca)):
    if k in ga:
        print(f"d(out of {k:28s}) {rel(ga[k], ca[k]):.1e}")
print("worst parameter gradient:", max((rel(gp[k], cp[k]), k) for k in cp))
