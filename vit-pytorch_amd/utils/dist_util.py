"""Single-node data parallelism for the ViT hot path.

The reference wraps the model in ``torch.nn.parallel.DistributedDataParallel``
(``/root/reference/train.py:270-278``; its own ``utils/dist_util.py:3-18`` only holds the rank /
world-size getters kept below).  This module replaces the wrapper with a gradient bucketer built
for MI355X nodes: gradients are packed into a few large flat fp32 buckets (reverse registration
order, i.e. the order backward produces them), and each bucket is mean-all-reduced with RCCL
(``torch.distributed`` backend "nccl") on a side HIP stream as soon as its last gradient has been
accumulated, overlapping the rest of backward.  xGMI is point-to-point, so few large collectives
beat many small ones: the default bucket is 64 MiB.

One process per GPU; ``MASTER_ADDR``/``RANK``/``WORLD_SIZE`` come from the launcher (env://).
"""
import os

import torch
import torch.distributed as dist


def get_rank():
    if not dist.is_available() or not dist.is_initialized():
        return 0
    return dist.get_rank()


def get_world_size():
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size()


def is_main_process():
    return get_rank() == 0


def format_step(step):
    if isinstance(step, str):
        return step
    names = ("Training Epoch", "Training Iteration", "Validation Iteration")
    return "".join("{}: {} ".format(n, v) for n, v in zip(names, step))


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "work", "ready")

    def __init__(self, params, device):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64          # keep every slice 256-byte aligned
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.pending = len(params)
        self.work = None
        self.ready = set()

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)


def plan_buckets(params, bucket_bytes):
    """Greedy packing of ``params`` (already in reduction order) into buckets of at most
    ``bucket_bytes`` (a single larger tensor gets a bucket of its own).  Pure host logic."""
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.numel() * 4
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    return buckets


class GradBucketer:
    """Bucketed, overlapped mean-all-reduce of ``.grad`` for every trainable parameter.

    After ``loss.backward()`` returns, every ``p.grad`` is a view into a flat bucket holding the
    mean over ranks, and the current stream has been ordered after the collectives -- the same
    contract DistributedDataParallel gives ``clip_grad_norm_`` / ``optimizer.step()``.
    Parameters that received no gradient on this rank keep ``grad=None`` (their bucket slot is
    reduced as zeros so that collectives stay matched across ranks), which is what
    ``find_unused_parameters=True`` permits in the reference.
    """

    def __init__(self, params, bucket_mb=64, process_group=None, average=True, transport=None):
        # transport: "torch" = torch.distributed's RCCL binding (default), "rccl" = the kernel
        # library's own communicator (vitk_allreduce); env VITK_COMM selects when not given.
        self.transport = transport or os.environ.get("VITK_COMM", "torch")
        self.comm = None
        self.always_reduce = False      # tests: issue the collectives even in a one-rank group
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        params = [p for p in params if p.requires_grad]
        self.params = list(reversed(params))
        self.buckets = []
        self._where = {}
        self._armed = False
        self._hooks = []
        if not self.params:
            return
        device = self.params[0].device
        self.on_gpu = device.type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.on_gpu else None
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        if self.transport == "rccl" and self.on_gpu and dist.is_initialized():
            from vitk.comm import Communicator
            self.comm = Communicator(device=device)
        for group in plan_buckets(self.params, int(bucket_mb * (1 << 20))):
            b = _Bucket(group, device)
            for i, p in enumerate(group):
                self._where[p] = (b, i)
            self.buckets.append(b)
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        b, i = self._where[p]
        if i in b.ready:          # gradient accumulated twice in one backward (shared use): re-copy
            self._stage(b, i, p)
            return
        b.ready.add(i)
        self._stage(b, i, p)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    @staticmethod
    def _stage(b, i, p):
        v = b.view(i)
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v

    def _launch(self, b):
        if self.world == 1 and self.comm is None and not self.always_reduce:
            return
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                b.work = self._allreduce(b.flat)
        else:
            b.work = self._allreduce(b.flat)

    def _allreduce(self, flat):
        if self.comm is not None:           # enqueued on the side stream (current here); nothing to wait on
            self.comm.all_reduce(flat, average=self.average)
            return None
        if self.average and self.backend == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work

    # ------------------------------------------------------------------ end of backward
    def finish(self):
        """Flush incomplete buckets (unused parameters), wait for every collective."""
        for b in self.buckets:
            if b.pending > 0 and b.work is None:
                for i, p in enumerate(b.params):
                    if i not in b.ready:
                        b.view(i).zero_()
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if self.average and self.backend != "nccl":
                    b.flat.div_(self.world)
                b.work = None
            b.pending = len(b.params)
            b.ready.clear()
        if self.on_gpu and (self.world > 1 or self.comm is not None or self.always_reduce):
            torch.cuda.current_stream().wait_stream(self.stream)
        self._armed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module, src=0, process_group=None):
    """Rank ``src``'s parameters and buffers to everyone (what DDP's constructor does)."""
    if get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=process_group)


class DataParallel(torch.nn.Module):
    """Minimal stand-in for ``DistributedDataParallel(model, device_ids=[rank],
    find_unused_parameters=True)`` (reference train.py:274-278): broadcasts at construction,
    then reduces gradients through a :class:`GradBucketer` during ``backward``."""

    def __init__(self, module, device_ids=None, find_unused_parameters=True, bucket_mb=64,
                 process_group=None, transport=None):
        super().__init__()
        self.module = module
        broadcast_parameters(module, 0, process_group)
        self.bucketer = GradBucketer(module.parameters(), bucket_mb=bucket_mb, process_group=process_group,
                                     transport=transport)
        if any(p.is_cuda for p in module.parameters()):
            # The bucket all-reduces run on a side stream while backward is still computing and hold
            # compute units for their whole duration.  The persistent attention-backward kernel counts
            # on one workgroup per CU; split it finer so that a workgroup that has to wait for a CU
            # delays the launch by a quarter of it at most (free on an idle GPU).
            import vitk
            vitk.check(vitk.lib.vitk_attn_set_oversubscription(4), "vitk_attn_set_oversubscription")

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
