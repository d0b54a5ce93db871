#!/usr/bin/env python3
"""Same-process A/B of the optimizer step inside the full training step: torch (clip_grad_norm_ +
SGD) against utils.optim.FusedSGD, blocks of steps alternating.  usage: optim_ab.py [batch] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from models import config as cfgmod  # noqa: E402
from models.ViT.modeling import VisionTransformer  # noqa: E402
from utils.optim import FusedSGD  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
vitk.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
model = VisionTransformer(cfgmod.get_config("ViT-B_16"), 224, zero_head=True, num_classes=10).to(dev).train()
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 10, (B,), device=dev)
opts = {"torch": torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9),
        "fused": FusedSGD(model.parameters(), lr=1e-3, momentum=0.9)}


def block(kind, steps):
    opt = opts[kind]
    host_opt = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = model(x, y)
        loss.backward()
        h0 = time.perf_counter()
        if kind == "fused":
            opt.step(max_grad_norm=1.0)
        else:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        host_opt += time.perf_counter() - h0
        opt.zero_grad()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{kind:6s}: {1e3 * dt / steps:7.2f} ms/step   host loop {1e3 * t_host / steps:6.2f} ms/step   "
          f"host in optimizer {1e3 * host_opt / steps:5.2f} ms/step", flush=True)


for kind in ("torch", "fused"):
    block(kind, 3)
for _ in range(3):
    for kind in ("torch", "fused"):
        block(kind, K)
