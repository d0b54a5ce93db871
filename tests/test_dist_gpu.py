"""The data-parallel wrapper on the GPU: a one-rank RCCL group exercises the real path (flat fp32
buckets, post-accumulate hooks, all-reduce on the side stream, hand-back before backward returns)
and must leave gradients identical to the unwrapped model."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_data_parallel_one_rank_rccl_matches_plain_backward():
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    from utils.dist_util import DataParallel

    dev = torch.device("cuda", 0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1, device_id=dev)
    try:
        vitk.set_compute_dtype(torch.bfloat16)
        cfg = cfgmod.get_tiny_config()
        torch.manual_seed(3)
        model = VisionTransformer(cfg, 32, zero_head=False, num_classes=10).to(dev).train()
        x = torch.randn(6, 3, 32, 32, device=dev)
        y = torch.randint(0, 10, (6,), device=dev)
        model(x, y).backward()
        want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        model.zero_grad(set_to_none=True)

        dp = DataParallel(model, bucket_mb=0.05)
        dp.bucketer.always_reduce = True                    # run all_reduce(AVG) on the side stream for real
        assert len(dp.bucketer.buckets) > 1
        for _ in range(2):                                  # second pass reuses the bucket views
            model.zero_grad()
            dp(x, y).backward()
            torch.cuda.synchronize()
            for n, p in model.named_parameters():
                if n in want:
                    # atomics in the weight-gradient kernels reorder fp32 sums between runs
                    scale = want[n].abs().max().clamp_min(1e-12)
                    assert ((p.grad - want[n]).abs().max() / scale) < 1e-4, n
                    b, i = dp.bucketer._where[p]
                    assert p.grad.data_ptr() == b.view(i).data_ptr(), n      # lives in its bucket slice
            # the kernels accumulated every gradient of the five modules straight into the buckets:
            # only the stock torch classifier head (weight, bias) had to be copied in
            assert dp.bucketer.copies <= 2, dp.bucketer.copies
            assert dp.bucketer.launch_log == list(range(len(dp.bucketer.buckets)))
        # accumulation without zero_grad: the second pass adds on top of the reduced values
        dp(x, y).backward()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            if n in want:
                scale = want[n].abs().max().clamp_min(1e-12)
                assert ((p.grad - 2 * want[n]).abs().max() / scale) < 2e-4, n
        model.zero_grad()
        dp(x, y).backward()

        # the fused optimizer step on gradients that are views into the flat buckets
        from utils.optim import FusedSGD
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        norm = FusedSGD(model.parameters(), lr=0.1, momentum=0.9).step(max_grad_norm=1.0)
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        assert abs(float(norm) - float(total)) <= 1e-5 * float(total)
        coef = torch.clamp(1.0 / (total + 1e-6), max=1.0)
        for n, p in model.named_parameters():
            if n in grads:
                assert torch.allclose(p.detach(), before[n] - 0.1 * coef * grads[n], rtol=2e-6, atol=1e-7), n
    finally:
        dist.destroy_process_group()
        vitk.set_compute_dtype(torch.bfloat16)


def test_rccl_transport_through_the_c_abi():
    """vitk_comm_* on a one-rank communicator: all-reduce (mean) and broadcast are identities, and the
    bucketer driven through them leaves the same gradients as plain backward."""
    import vitk
    from vitk.comm import Communicator
    from utils.dist_util import GradBucketer

    dev = torch.device("cuda", 0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1, device_id=dev)
    try:
        comm = Communicator(device=dev)
        for dtype in (torch.float32, torch.bfloat16):
            t = torch.randn(100003, device=dev).to(dtype)
            want = t.clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            comm.all_reduce(t, average=True, stream=side)
            comm.broadcast(t, root=0, stream=side)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(t, want)
        comm.close()

        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.GELU(), torch.nn.Linear(256, 16)).to(dev)
        x = torch.randn(32, 64, device=dev)
        net(x).square().mean().backward()
        want = [p.grad.clone() for p in net.parameters()]
        net.zero_grad(set_to_none=True)
        b = GradBucketer(net.parameters(), bucket_mb=0.01, transport="rccl")
        assert b.comm is not None and len(b.buckets) > 1
        net(x).square().mean().backward()
        torch.cuda.synchronize()
        for p, w in zip(net.parameters(), want):
            assert torch.allclose(p.grad, w, rtol=1e-6, atol=1e-7)
        b.remove()
        b.comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ranks,batch", [(2, 16), (4, 8), (8, 4)])
def test_ranks_share_the_gpu_through_bench_rehearsal(ranks, batch):
    """`python bench.py --gpus N --rehearse`: bench.py starts the ranks itself; all compute on GPU 0 and
    exchange gradients through gloo, so the N > 1 path of the wrapper (collectives on the side stream, used
    flags, 1024 persistent GEMM workgroups, attention oversubscription) runs end to end on a one-GPU box.
    The ranks see different data; their models must stay identical, which the shared loss line shows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--rehearse", "--batch",
                        str(batch), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=140)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["config"]["global_batch"] == ranks * batch
    assert out["config"]["parallelism"] == f"dp{ranks}"
    assert out["value"] > 0 and math.isfinite(out["final_loss"]) and out["replicas_in_sync"] is True
    assert out["gloo_ranks"] == ranks                      # every rank joined the exchange (a sum of ones)
    per = out["images_per_sec_per_rank"]
    assert 0 < per["min"] <= per["max"] and abs(out["value"] - ranks * per["min"]) <= 0.02 * out["value"]
