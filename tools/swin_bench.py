#!/usr/bin/env python3
"""Training-step throughput of the Swin RKR drop-in (models/Swin/swin_RKR.py) on one GPU, bf16, synthetic
data, and the share of the window-attention kernels in it.
Usage: python tools/swin_bench.py [dataset-preset: imagenet|cifar100] [batch] [img] [--steps N --warmup W]
With --steps only the N training steps run (after W warm-up steps): the form the profiling tools use."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402
from models import config as cfgmod  # noqa: E402
from models.Swin.swin_RKR import SwinTransformer  # noqa: E402
from utils.optim import FusedSGD  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    argv = list(sys.argv[1:])
    steps = warmup = None
    for flag in ("--steps", "--warmup"):
        if flag in argv:
            k = argv.index(flag)
            val = int(argv[k + 1])
            del argv[k:k + 2]
            if flag == "--steps":
                steps = val
            else:
                warmup = val
    preset = argv[0] if len(argv) > 0 else "imagenet"
    batch = int(argv[1]) if len(argv) > 1 else 128
    img = int(argv[2]) if len(argv) > 2 else 224
    vitk.set_compute_dtype(torch.bfloat16)
    cfg = cfgmod.get_config("Swin_RKR", preset)
    cfg.task_num = 3
    torch.manual_seed(0)
    model = SwinTransformer(cfg, img, num_classes=[10, 10, 10]).cuda().train()
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=0)
    x = torch.randn(batch, 3, img, img, device="cuda")
    y = torch.randint(0, 10, (batch,), device="cuda")

    def step():
        loss = model(x, y, task=1)
        loss.backward()
        opt.step(max_grad_norm=1.0)
        opt.zero_grad()
        return loss

    ms = timeit(step) if steps is None else timeit(step, steps, warmup if warmup is not None else 2)
    print(f"Swin_RKR preset={preset} embed={cfg.embed_dim} depths={list(cfg.depths)} heads={list(cfg.num_heads)[:len(cfg.depths)]} "
          f"drop_path={cfg.drop_path_rate} img={img} batch={batch}: {ms:.1f} ms/step, {batch / ms * 1e3:.0f} img/s", flush=True)
    if steps is not None:
        return
    # window attention alone, per stage geometry
    res = img // cfg.patch_size
    total = 0.0
    for i, (depth, heads) in enumerate(zip(cfg.depths, cfg.num_heads)):
        H = res // 2 ** i
        ws = min(cfg.window_size, H)
        C = cfg.embed_dim * 2 ** i
        qkv = torch.randn(batch * H * H, 3 * C, device="cuda").bfloat16().requires_grad_(True)
        tab = torch.randn((2 * ws - 1) ** 2, heads, device="cuda", requires_grad=True)
        shift = ws // 2 if H > ws else 0
        fwd = timeit(lambda: ops.window_attention(qkv.detach(), tab.detach(), batch, H, H, heads, ws, shift))
        o = ops.window_attention(qkv, tab, batch, H, H, heads, ws, shift)
        do = torch.randn_like(o)

        def fb():
            qkv.grad = tab.grad = None
            ops.window_attention(qkv, tab, batch, H, H, heads, ws, shift).backward(do)
        both = timeit(fb)
        byts = qkv.numel() * 2 + o.numel() * 2
        print(f"  stage {i}: {H}x{H} tokens, C={C}, heads={heads}, window={ws}: fwd {fwd:.3f} ms ({byts / fwd / 1e6:.0f} GB/s of q,k,v,o), "
              f"fwd+bwd {both:.3f} ms, x{depth} blocks", flush=True)
        total += both * depth
    print(f"  window attention fwd+bwd over all blocks: {total:.1f} ms of the {ms:.1f} ms step")


if __name__ == "__main__":
    main()
