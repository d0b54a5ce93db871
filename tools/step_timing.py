#!/usr/bin/env python3
"""How much of a training step is host-bound: time to enqueue a step vs time to execute it."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk
from models import config as cfgmod
from models.ViT.modeling import VisionTransformer
vitk.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfg = cfgmod.get_config("ViT-B_16")
model = VisionTransformer(cfg, 224, zero_head=True, num_classes=10).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=3e-2, momentum=0.9)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 10, (B,), device=dev)
def step(parts):
    t0 = time.perf_counter(); loss = model(x, y); t1 = time.perf_counter()
    loss.backward(); t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step(); opt.zero_grad(); t3 = time.perf_counter()
    parts[0] += t1 - t0; parts[1] += t2 - t1; parts[2] += t3 - t2
for _ in range(3): step([0, 0, 0])
torch.cuda.synchronize()
parts = [0, 0, 0]
t0 = time.perf_counter()
for _ in range(2): step(parts)          # short enough not to fill the launch queue
tq = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f"enqueue {1e3*tq/2:.1f} ms/step (fwd {1e3*parts[0]/2:.1f} bwd {1e3*parts[1]/2:.1f} opt {1e3*parts[2]/2:.1f}); executed {1e3*tw/2:.1f} ms/step")
