"""The gradient bucketer on CPU: bucket planning (pure host logic) and a world_size-2 gloo run
compared with the single-process mean of per-shard gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_plan_buckets_reverse_order_and_caps():
    from utils.dist_util import plan_buckets
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 1000, 300000, 5, 700000, 20)]
    buckets = plan_buckets(list(reversed(ps)), bucket_bytes=1 << 20)
    flat = [p for b in buckets for p in b]
    assert all(a is b for a, b in zip(flat, reversed(ps))) and len(flat) == len(ps)   # order preserved
    for b in buckets:
        nbytes = sum(p.numel() * 4 for p in b)
        assert nbytes <= (1 << 20) or len(b) == 1          # only a lone oversized tensor may exceed the cap
    assert [len(b) for b in buckets] == [1, 1, 1, 1, 2]


def test_single_process_bucketer_is_transparent():
    from utils.dist_util import GradBucketer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    ref = [p.detach().clone() for p in net.parameters()]
    x = torch.randn(5, 8)
    net(x).square().sum().backward()
    want = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    b = GradBucketer(net.parameters(), bucket_mb=0.0005)
    assert len(b.buckets) > 1
    for _ in range(2):                                     # twice: .grad views are reused / re-staged
        net.zero_grad()
        net(x).square().sum().backward()
        for p, w, r in zip(net.parameters(), want, ref):
            assert torch.allclose(p.grad, w) and torch.equal(p.detach(), r)
    # accumulation without zero_grad adds on top of the reduced values
    net(x).square().sum().backward()
    for p, w in zip(net.parameters(), want):
        assert torch.allclose(p.grad, 2 * w)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vit-pytorch_amd"))
    from utils.dist_util import DataParallel, get_rank, get_world_size
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    assert get_rank() == rank and get_world_size() == world
    torch.manual_seed(100 + rank)                          # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.GELU(), torch.nn.Linear(32, 32),
                              torch.nn.GELU(), torch.nn.Linear(32, 4))
    unused = torch.nn.Linear(3, 3)                         # never used in forward (find_unused_parameters)
    net.add_module("unused", unused)
    dp = DataParallel(net, bucket_mb=0.002)
    assert len(dp.bucketer.buckets) > 1
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 8, generator=g)
    y = torch.randn(8, 4, generator=g)
    xs, ys = x[rank::world], y[rank::world]
    res = {}
    for step in range(2):
        net.zero_grad()
        h = xs
        for m in list(net.children())[:-1]:
            h = m(h)
        torch.nn.functional.mse_loss(h, ys).backward()
        res[step] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}
    res["params"] = {n: p.detach().clone() for n, p in net.named_parameters()}
    torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_mean_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n      # broadcast from rank 0
    # single-process reference on rank 0's (broadcast) weights: mean over ranks of the shard gradients
    net = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.GELU(), torch.nn.Linear(32, 32),
                              torch.nn.GELU(), torch.nn.Linear(32, 4))
    net.add_module("unused", torch.nn.Linear(3, 3))
    net.load_state_dict(r0["params"])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 8, generator=g)
    y = torch.randn(8, 4, generator=g)
    want = {n: torch.zeros_like(p) for n, p in net.named_parameters()}
    for rank in range(world):
        net.zero_grad()
        h = x[rank::world]
        for m in list(net.children())[:-1]:
            h = m(h)
        torch.nn.functional.mse_loss(h, y[rank::world]).backward()
        for n, p in net.named_parameters():
            if p.grad is not None:
                want[n] += p.grad / world
    for step in (0, 1):
        for n in want:
            for r in (r0, r1):
                if n.startswith("unused"):
                    assert r[step][n] is None, n
                else:
                    assert torch.allclose(r[step][n], want[n], rtol=1e-6, atol=1e-7), (step, n)
