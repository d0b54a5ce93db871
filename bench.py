#!/usr/bin/env python3
"""Training-throughput benchmark of the ViT hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (started plainly: launches the N ranks itself, as child processes)

One step = forward + backward + bucketed gradient all-reduce + clip_grad_norm_(1.0) + SGD(momentum
0.9) step + warmup-cosine scheduler step (the loop of reference train.py:513-538) on a synthetic
224 px batch that already sits in HBM, ViT-B/16, bf16 activations, dropout 0.1 active (train mode).
Weak scaling: every rank processes `--batch` images per step (default 512, the reference's own
default train.py:599; `--batch 256` is the second documented data point).  Rank 0 prints ONE JSON line; with
N > 1 it also carries `rccl_ranks` (a sum of ones over the communicator: how many ranks really joined),
`images_per_sec_per_rank` {min, max} (every rank's own clock) and `replicas_in_sync`.

Extra objects on that line:
  roofline     - the dominant kernel (the bf16 MFMA GEMM `gemm_nt_bf16_kernel`, every instance of
                 it inside the sampled timed steps - every 5th and the last): algorithmic FLOPs of
                 those launches / their summed duration, measured with HIP events on the launch
                 stream during the timed region.
                 `traffic` = HBM bytes per launch from the committed rocprofv3 counter passes
                 (profiles/r03_counters.json), reported only while the kernel sources still hash to
                 what those passes were collected on; null otherwise; `traffic_source` names the file
                 and the commit it was collected at, or says why there is no figure.
  cpu_baseline - the CPU oracle (fp32 PyTorch restatement of the reference) timed on this host's
                 cores on a bounded sample (ViT-B/16, batch 8, train mode with dropout 0.1, 3 warm-up
                 + 5 timed iterations), rank 0, N=1 only.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak
GEOM = {"B_16": dict(name="ViT-B_16", hidden=768, mlp=3072, heads=12, layers=12),
        "L_16": dict(name="ViT-L_16", hidden=1024, mlp=4096, heads=16, layers=24),
        # the R50 stem (units 3, 4, 9, 1/16 map) under ViT-B: reference models/ViT/_configs.py:224-232
        "R50_B16": dict(name="R50-ViT-B_16", hidden=768, mlp=3072, heads=12, layers=12, stem=(3, 4, 9))}


def train_flops_per_image(g, img, patch=16):
    """Matmul FLOPs (2*MAC) of one forward, times 3 for forward+backward (the usual convention;
    the patch embedding has no input gradient, so the exact factor is a hair below 3)."""
    n = (img // patch) ** 2 + 1
    D, F_, H, L = g["hidden"], g["mlp"], g["heads"], g["layers"]
    dh = D // H
    layer = 2 * n * D * 3 * D + 2 * 2 * H * n * n * dh + 2 * n * D * D + 2 * 2 * n * D * F_
    if "stem" in g:                     # hybrid: the patch embedding is a 1x1 convolution over the 1024-channel map
        return 3.0 * (L * layer + 2 * (n - 1) * 1024 * D + stem_flops_per_image(g["stem"], img))
    fwd = L * layer + 2 * (n - 1) * (3 * patch * patch) * D
    return 3.0 * fwd


def stem_flops_per_image(units, img, width=64):
    """Convolution FLOPs (2*MAC) of one forward of the ResNetV2 stem (modeling_resnet.py:129-164)."""
    s = img // 2                                                    # root 7x7 / 2
    fl = 2 * s * s * 49 * 3 * width
    s = (s - 3) // 2 + 1                                            # 3x3 / 2 max-pool without padding
    cin = width
    for stage, n_units in enumerate(units):
        cout, cmid = width * 4 * 2 ** stage, width * 2 ** stage
        for u in range(n_units):
            stride = 2 if (stage > 0 and u == 0) else 1
            so = (s - 1) // stride + 1
            fl += 2 * s * s * cin * cmid + 2 * so * so * 9 * cmid * cmid + 2 * so * so * cmid * cout
            if u == 0:
                fl += 2 * so * so * cin * cout                      # projection shortcut
            s, cin = so, cout
    return fl


class GemmTimer:
    """Wraps vitk.ops.gemm_nt to bracket every launch with HIP events on its stream."""

    def __init__(self, ops):
        self.ops, self.orig, self.records, self.on = ops, ops.gemm_nt, [], False

    def install(self):
        def timed(a, b, out, *args, **kw):
            if not self.on:
                return self.orig(a, b, out, *args, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(a, b, out, *args, **kw)
            e1.record()
            self.records.append((e0, e1, 2.0 * a.shape[0] * a.shape[1] * b.shape[0]))
            return r
        self.ops.gemm_nt = timed

    @staticmethod
    def sample_set(steps):
        """Timed steps whose launches carry event pairs: the middle one and the last one."""
        return {steps - 1} | ({steps // 2 - 1} if steps >= 4 else set())

    @classmethod
    def sampled_steps(cls, steps):
        return len(cls.sample_set(steps))

    def summary(self):
        if not self.records:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.records)
        fl = sum(f for _, _, f in self.records)
        return dict(launches=len(self.records), ms=ms, flops=fl)


def cpu_baseline(iters=5, warm=3):
    """fp32 CPU oracle, ViT-B/16 224 px, batch 8, training-mode forward+backward (BASELINE config 0)."""
    from oracle import synth, vit_oracle as O
    params = synth.synth_params("B_16", 224, 10, seed=3)
    x, y = synth.synth_batch(8, 224, 10, seed=3)
    cfg = dict(synth.oracle_cfg("B_16"), dropout=0.1)      # model.train() with the preset's dropout
    for _ in range(warm):
        O.vit_loss_and_grads(params, x, y, cfg)
    t0 = time.perf_counter()
    for _ in range(iters):
        O.vit_loss_and_grads(params, x, y, cfg)
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(8 / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"ViT-B/16 224px fp32 fwd+bwd, batch 8, train mode (dropout 0.1), {iters} timed iterations after "
                      f"{warm} warm-up (oracle/vit_oracle.py), {dt:.2f} s/iter"}


def kernel_source_hash():
    """sha256 over the kernel sources and the C-ABI header: identifies the tree a counter file belongs to
    (same recipe as tools/collect_counters.py)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "vit-pytorch_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files.append(os.path.join(ROOT, "include", "vitk.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


COUNTER_FILE = "profiles/r03_counters.json"


def nt_traffic_per_launch(args):
    """(HBM bytes per NT-GEMM launch, where the figure comes from): read + write, gfx950-corrected, averaged
    over every NT instance of the step, from the committed rocprofv3 counter passes -- or (None, why) when that
    file was collected on other kernel sources, another workload or another batch."""
    try:
        with open(os.path.join(ROOT, COUNTER_FILE)) as fh:
            cj = json.load(fh)
        if cj.get("source_sha256") != kernel_source_hash():
            return None, f"{COUNTER_FILE} was collected on other kernel sources"
        if not (args.variant == "plain" and args.img == 224 and args.model == "B_16" and args.batch == 512
                and cj.get("workload") == "vit"):
            return None, f"{COUNTER_FILE} holds the default workload only"
        tot, n = 0.0, 0
        for k, e in cj["kernels"].items():
            if k.startswith("gemm_nt_"):
                tot += (e["read_bytes_per_dispatch"] + e["write_bytes_per_dispatch"]) * e["dispatches"]
                n += e["dispatches"]
        if not n:
            return None, f"{COUNTER_FILE} holds no NT GEMM"
        return int(tot / n), f"{COUNTER_FILE}@{(cj.get('git_head_when_collected') or '?')[:10]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
    except (OSError, KeyError, ValueError, TypeError) as e:
        return None, f"{COUNTER_FILE}: {type(e).__name__}"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly: run the N ranks as children of this process through
    torch.distributed.run (this process has not touched the GPU and never does), pass their output
    through and exit with their status."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(world, rank):
    """Launch-path rehearsal without a GPU: rendezvous over gloo, one all-reduce, one JSON line."""
    dist.init_process_group("gloo", init_method="env://")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "allreduce_sum": float(t.item()),
                          "expected": world * (world + 1) / 2}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 512, the reference's own default)")
    ap.add_argument("--model", default="B_16", choices=sorted(GEOM))
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--dropout", type=float, default=None, help="override the preset's 0.1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=64)
    ap.add_argument("--optim", default="fused", choices=["fused", "torch"],
                    help="fused: utils.optim.FusedSGD (clip + SGD in two launches); torch: clip_grad_norm_ + torch.optim.SGD")
    ap.add_argument("--variant", default="plain", choices=["plain", "rkr", "pb", "rkrpb"],
                    help="rkr: task-conditioned linears of models/ViT/modeling_RKR.py; pb: piggyback masks of "
                         "modeling_PB.py (weights frozen, masks trained); rkrpb: masked shared factors of modeling_RKRPB.py "
                         "(3 tasks, task 1 trained)")
    ap.add_argument("--no-wrap", action="store_true",
                    help="N=1 only: leave the model out of the data-parallel wrapper (gradients then go through the "
                         "per-pass arena instead of the flat buckets)")
    ap.add_argument("--dry-run", action="store_true", help="rehearse the multi-rank launch on the CPU (gloo), no GPU work")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on a ONE-GPU box: every rank computes on GPU 0 and the gradient exchange "
                         "goes through gloo; exercises the N > 1 path of the data-parallel wrapper (side stream, flags, "
                         "finer persistent launches) end to end.  Its number is not a benchmark result.")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 512

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if world != args.gpus:
        args.gpus = world
    if args.dry_run:
        return dry_run(world, rank)
    if args.rehearse:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("VITK_REHEARSE_WATCHDOG", "60")), exit=True)   # a hang names its line
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo", init_method="env://")
        else:
            dist.init_process_group("nccl", init_method="env://", device_id=dev)

    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    from utils.dist_util import DataParallel
    from utils.scheduler import WarmupCosineSchedule

    vitk.set_compute_dtype(torch.bfloat16)
    g = GEOM[args.model]
    cfg = cfgmod.get_config(g["name"])
    if args.dropout is not None:
        cfg.transformer.dropout_rate = args.dropout
    torch.manual_seed(100)                          # reference default seed (train.py:628)
    if args.variant != "plain":
        cfg.RG, cfg.SFG, cfg.K, cfg.rkr_scale, cfg.multi_head, cfg.task_num = True, True, 2, 1e-1, True, 3
        if args.variant == "rkr":
            from models.ViT.modeling_RKR import VisionTransformer as VisionTransformerMT
        else:
            cfg.threshold_fn, cfg.mask_scale, cfg.mask_init = "binarizer", 1e-2, "1s"
            cfg.RGnoPB = cfg.LMnoPB = cfg.RMnoPB = cfg.SFGnoPB = cfg.PBwR = False
            if args.variant == "pb":
                from models.ViT.modeling_PB import VisionTransformer as VisionTransformerMT
            else:
                from models.ViT.modeling_RKRPB import VisionTransformer as VisionTransformerMT
        inner = VisionTransformerMT(cfg, args.img, zero_head=True, num_classes=[10, 10, 10])
        if args.variant == "pb":
            # the piggyback layers leave weight / bias to a checkpoint: random-init them here, frozen as in
            # the reference's piggyback fine-tuning (only the masks, the norms and the heads train)
            from models.ViT.layers_PB import ElementWiseLinear
            for m in inner.modules():
                if isinstance(m, ElementWiseLinear):
                    torch.nn.init.normal_(m.weight, std=0.02)
                    torch.nn.init.zeros_(m.bias)
                    m.weight.requires_grad_(False)
                    m.bias.requires_grad_(False)
        inner = inner.to(dev).train()

        class _Task1(torch.nn.Module):          # the driver trains one task at a time (train.py:390)
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, x, y):
                return self.m(x, y, task=1)
        model = _Task1(inner)
    else:
        model = VisionTransformer(cfg, args.img, zero_head=True, num_classes=10).to(dev).train()
    # N=1 goes through the same wrapper (gradient hooks, flat buckets) as N>1; only the collectives are absent
    net = model if (world == 1 and args.no_wrap) else DataParallel(model, bucket_mb=args.bucket_mb)
    if args.optim == "fused":
        from utils.optim import FusedSGD
        opt = FusedSGD([p for p in model.parameters() if p.requires_grad], lr=3e-2, momentum=0.9, weight_decay=0)
    else:
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=3e-2, momentum=0.9, weight_decay=0)
    sched = WarmupCosineSchedule(opt, warmup_steps=500, t_total=10000)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((args.batch, 3, args.img, args.img), device=dev, generator=gen)
    y = torch.randint(0, 10, (args.batch,), device=dev, generator=gen)

    timer = GemmTimer(vitk.ops)
    if not args.no_kernel_timer:
        timer.install()

    def step():
        loss = net(x, y)
        loss.backward()
        if args.optim == "fused":
            opt.step(max_grad_norm=1.0)
        else:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        sched.step()
        opt.zero_grad()
        return loss

    def note(msg):
        if args.rehearse:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    note("model built")
    for _ in range(args.warmup):
        step()
        note("warm-up step enqueued")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Per-launch event pairs slow a step down when every GEMM carries them, so they are attached on two of
    # the timed steps only (the middle one and the last one).
    fence()
    sampled = timer.sample_set(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.on = i in sampled
        loss = step()
        note(f"step {i} enqueued")
    timer.on = False
    fence()
    dt = time.perf_counter() - t0
    ranks_joined, rank_ips = None, None
    if world > 1:
        cdev = "cpu" if args.rehearse else dev          # where the process group's collectives run
        own = torch.tensor([dt], device=cdev, dtype=torch.float64)
        t, lo = own.clone(), own.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dt = float(t.item())
        # every rank's own clock over the same K steps: the slowest sets `value`, the spread shows stragglers
        rank_ips = (args.batch * args.steps / dt, args.batch * args.steps / float(lo.item()))
        # how many ranks the communicator really joins: a sum of ones over the transport the gradients use
        one = torch.ones(1, device=cdev, dtype=torch.float32)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_joined = int(round(float(one.item())))
    final_loss = float(loss.detach())
    in_sync = None
    if world > 1:
        # the replicas saw different data and exchanged only gradients: their parameters must still be the same bits
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        if args.rehearse:
            chk = chk.cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool((lo == hi).item())

    if rank == 0:
        ips = world * args.batch * args.steps / dt
        fpi = train_flops_per_image(g, args.img)
        out = {
            "metric": "train images/sec ViT-B/16 224px" if args.model == "B_16" else f"train images/sec {g['name']} {args.img}px",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            **({"rehearsal": "all ranks on GPU 0, gloo exchange: not a benchmark result"} if args.rehearse else {}),
            "config": {"workload": f"{g['name']}{'' if args.variant == 'plain' else ' ' + args.variant.upper()} {args.img}px fine-tune step (fwd+bwd+"
                                   f"{'gradient bucketing' if world == 1 else 'bucketed grad all-reduce'}+clip+SGD), "
                                   f"dropout {cfg.transformer.dropout_rate}, 10 classes, random-init weights",
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                       "tokens_per_image": (args.img // 16) ** 2 + 1,
                       "parallelism": f"dp{world}", "bucket_mb": args.bucket_mb, "optimizer": args.optim},
            "train_gflop_per_image": round(fpi / 1e9, 2),
            "model_tflops": round(ips * fpi / 1e12, 1),
            "mfma_peak_fraction": round(ips * fpi / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            "final_loss": round(final_loss, 4),
            "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
        }
        if in_sync is not None:
            out["replicas_in_sync"] = in_sync
            out["rccl_ranks" if not args.rehearse else "gloo_ranks"] = ranks_joined
            out["images_per_sec_per_rank"] = {"min": round(rank_ips[0], 2), "max": round(rank_ips[1], 2)}
        s = timer.summary()
        traffic = nt_traffic_per_launch(args)
        if s:
            ach = s["flops"] / (s["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic[0], "traffic_source": traffic[1],
                               "kernel": "gemm_nt_pps_kernel / gemm_nt_bf16_kernel (bf16 MFMA NT GEMM, all epilogue instances)",
                               "launches_per_step": s["launches"] // timer.sampled_steps(args.steps),
                               "sampled_steps": timer.sampled_steps(args.steps),
                               "avg_launch_us": round(1e3 * s["ms"] / s["launches"], 1),
                               "gflop_per_launch": round(s["flops"] / s["launches"] / 1e9, 1)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
