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


# ----------------------------------------------------------------------------- buckets as gradient storage
def test_plan_buckets_keeps_colocated_groups_adjacent():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vit-pytorch_amd"))
    from utils.dist_util import GradBucketer, plan_buckets
    from vitk.ops import colocate_grads
    q, qb, k, kb, v, vb, o = [torch.nn.Parameter(torch.zeros(n)) for n in (128, 64, 128, 64, 128, 64, 128)]
    colocate_grads(q, k, v)
    colocate_grads(qb, kb, vb)
    order = [o, vb, v, kb, k, qb, q]                       # reverse registration order, as the bucketer sees it
    flat = [p for b in plan_buckets(order, 1 << 20) for p in b]
    assert [id(p) for p in flat] == [id(p) for p in (o, qb, kb, vb, q, k, v)]
    # a group is never split over two buckets, even if that overfills one
    buckets = plan_buckets(order, 128 * 4 * 2)
    where = {id(p): i for i, b in enumerate(buckets) for p in b}
    assert where[id(q)] == where[id(k)] == where[id(v)] and where[id(qb)] == where[id(kb)] == where[id(vb)]
    # frozen members simply drop out
    k.requires_grad_(False)
    b = GradBucketer([q, qb, k, kb, v, vb, o])
    got = [id(p) for bk in b.buckets for p in bk.params]
    assert id(k) not in got and got.index(id(v)) == got.index(id(q)) + 1


class _SlotLinear(torch.autograd.Function):
    """x @ w^T + b whose backward accumulates dW / db into the parameters' bucket slices when the
    bucketer offers them: the protocol the kernel library's backward functions follow on the GPU."""
    took = 0

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.gparams = (w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        from vitk.ops import grad_block
        x, w = ctx.saved_tensors
        pw, pb = ctx.gparams
        blk = grad_block([pw]) if ctx.needs_input_grad[1] else None
        _SlotLinear.took += blk is not None
        dw = blk.view(w.shape) if blk is not None else torch.zeros_like(w)
        dw.add_(dy.t() @ x)
        blk = grad_block([pb]) if ctx.needs_input_grad[2] else None
        _SlotLinear.took += blk is not None
        db = blk if blk is not None else torch.zeros_like(pb)
        db.add_(dy.sum(0))
        return dy @ w, dw, db


class _Branchy(torch.nn.Module):
    """Four branches (slot-protocol linears) summed, then a stock torch trunk.  `skip` leaves one branch
    out of the forward: its parameters get no gradient on that rank."""

    def __init__(self):
        super().__init__()
        self.branch = torch.nn.ModuleList(torch.nn.Linear(8, 8) for _ in range(4))
        self.trunk = torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))

    def forward(self, x, skip):
        h = 0
        for k, lin in enumerate(self.branch):
            if k != skip:
                h = h + _SlotLinear.apply(x, lin.weight, lin.bias)
        return self.trunk(h)


_FROZEN = ("trunk.1.bias", "branch.2.bias")


def _branchy_data(step):
    g = torch.Generator().manual_seed(70 + step)
    return torch.randn(16, 8, generator=g), torch.randn(16, 4, generator=g)


def _worker4(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vit-pytorch_amd"))
    from utils.dist_util import DataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    net = _Branchy()
    for n, p in net.named_parameters():
        p.requires_grad_(n not in _FROZEN)
    dp = DataParallel(net, bucket_mb=0.0004)               # ~100 floats per bucket: many buckets
    assert len(dp.bucketer.buckets) >= 4
    res = {"copies": [], "launch": [], "took": []}
    for step in range(3):
        if step != 1:
            net.zero_grad()                                # step 1 accumulates on top of step 0 (no zero_grad)
        x, y = _branchy_data(step)
        _SlotLinear.took = 0
        torch.nn.functional.mse_loss(dp(x[rank::world], skip=rank), y[rank::world]).backward()
        res[step] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}
        res["copies"].append(dp.bucketer.copies)
        res["launch"].append(list(dp.bucketer.launch_log))
        res["took"].append(_SlotLinear.took)
        for n, p in net.named_parameters():                # every gradient lives in its bucket slice
            if p.grad is not None:
                b, i = dp.bucketer._where[p]
                assert p.grad.data_ptr() == b.view(i).data_ptr(), n
    res["params"] = {n: p.detach().clone() for n, p in net.named_parameters()}
    torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_gloo_frozen_unused_and_accumulation(tmp_path):
    """World 4: a frozen subset, a different unused branch on every rank, three passes of which the
    second accumulates without zero_grad (reference train.py:522-539) -- against the single-process
    mean of the per-shard gradients.  Collectives must have been issued in bucket order on every
    rank, and the slot-protocol gradients must have reached their buckets without a copy."""
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    net = _Branchy()
    net.load_state_dict(rs[0]["params"])
    for r in rs[1:]:
        for n in rs[0]["params"]:
            assert torch.equal(rs[0]["params"][n], r["params"][n]), n
    for n, p in net.named_parameters():
        p.requires_grad_(n not in _FROZEN)
    mean = {}
    for step in range(3):
        x, y = _branchy_data(step)
        acc = {n: torch.zeros_like(p) for n, p in net.named_parameters()}
        for rank in range(world):
            net.zero_grad()
            torch.nn.functional.mse_loss(net(x[rank::world], skip=rank), y[rank::world]).backward()
            for n, p in net.named_parameters():
                if p.grad is not None:
                    acc[n] += p.grad / world
        mean[step] = acc
    want = {0: mean[0], 1: {n: mean[0][n] + mean[1][n] for n in mean[0]}, 2: mean[2]}
    nb = None
    for rank, r in enumerate(rs):
        for step in range(3):
            for n in want[step]:
                got = r[step][n]
                if n in _FROZEN:
                    assert got is None, (rank, step, n)
                else:                  # also the branch this rank left out: other ranks produced its gradient
                    assert got is not None, (rank, step, n)
                    assert torch.allclose(got, want[step][n], rtol=1e-5, atol=1e-7), (rank, step, n)
        nb = len(r["launch"][0])
        for log in r["launch"]:
            assert log == list(range(nb)), (rank, log)           # strictly in bucket order, every bucket once
        # fresh passes: the three used branches (weight and live bias) accumulate in place; the stock
        # trunk gradients are copied in.  The accumulating pass copies nothing new into place either way.
        assert r["took"][0] >= 5 and r["took"][2] >= 5 and r["took"][1] == 0, (rank, r["took"])
        n_trunk_live = 3
        assert r["copies"][0] == n_trunk_live and r["copies"][2] == n_trunk_live, (rank, r["copies"])


def test_buckets_are_issued_in_index_order_whatever_completes_first():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vit-pytorch_amd"))
    from utils.dist_util import GradBucketer
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    bk = GradBucketer(params, bucket_mb=0.00008)           # one Linear per bucket: c, b, a
    assert len(bk.buckets) == 3
    x = torch.randn(3, 4)
    issued = []
    orig = bk._launch
    bk._launch = lambda bucket, index: (issued.append((index, bucket.pending)), orig(bucket, index))
    (a(x) + b(x)).sum().backward()                         # bucket 0 (c) never completes on its own
    assert [i for i, _ in issued] == [0, 1, 2] and issued[0][1] > 0
    assert c.weight.grad is None and torch.allclose(a.weight.grad, b.weight.grad)
