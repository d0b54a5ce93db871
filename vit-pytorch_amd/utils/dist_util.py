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
    __slots__ = ("flat", "flags", "params", "offsets", "pending", "work", "ready", "clean", "taken", "flags_host",
                 "flags_event")

    def __init__(self, params, device):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64          # keep every slice 256-byte aligned
        # one "this rank produced a gradient" flag per parameter rides at the end of the bucket and is
        # reduced with it: afterwards flag > 0 says that SOME rank did
        self.flat = torch.zeros(n + (len(params) + 63) // 64 * 64, dtype=torch.float32, device=device)
        self.flags = self.flat[n:n + len(params)]
        self.pending = len(params)
        self.work = None
        self.ready = set()
        self.clean = False        # this pass: zero-filled at its start, slices may be accumulated into
        self.taken = set()        # this pass: slices handed to a backward function
        # staging buffer of the flags of an incomplete bucket (parameters unused in a pass): pinned ONCE here
        # (pinning synchronises the device), guarded by an event so that it is not rewritten under a copy
        self.flags_host = torch.empty(len(params), dtype=torch.float32)
        self.flags_event = None
        if self.flat.is_cuda:
            self.flags_host = self.flags_host.pin_memory()
            self.flags_event = torch.cuda.Event()

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)


class _SlotRef:
    """What a parameter carries as ``_vitk_grad_slot`` while a GradBucketer owns its gradient: the
    backward functions of the kernel library ask it for the parameter's slice of the flat bucket and
    let their weight-gradient kernels accumulate straight into it, so the gradient never has to be
    copied into the bucket."""
    __slots__ = ("owner", "bucket", "index", "param")

    def __init__(self, owner, bucket, index, param):
        self.owner, self.bucket, self.index, self.param = owner, bucket, index, param

    @property
    def flat(self):
        return self.bucket.flat

    @property
    def offset(self):
        return self.bucket.offsets[self.index]

    def available(self):
        """True when the slice is zero and nobody else adds into it during this backward pass."""
        return self.owner._slot_available(self)

    def take(self):
        self.bucket.taken.add(self.index)


def _units(params):
    """``params`` (reduction order) as packing units: co-located groups stay together, in group order."""
    index = {id(p): k for k, p in enumerate(params)}
    units, seen = [], set()
    for p in params:
        if id(p) in seen:
            continue
        group = getattr(p, "_vitk_colocate", None)
        members = [q for q in group if id(q) in index] if group else [p]
        if not any(q is p for q in members):
            members = [p]
        for q in members:
            seen.add(id(q))
        units.append(members)
    return units


def plan_buckets(params, bucket_bytes):
    """Greedy packing of ``params`` (already in reduction order) into buckets of at most
    ``bucket_bytes`` (a single larger unit gets a bucket of its own).  Parameters tied by
    :func:`vitk.ops.colocate_grads` (attribute ``_vitk_colocate``) are placed where the first of them falls, adjacent and in group order.
    Pure host logic."""
    buckets, cur, cur_bytes = [], [], 0
    for unit in _units(list(params)):
        nbytes = sum(p.numel() for p in unit) * 4
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.extend(unit)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    return buckets


class GradBucketer:
    """Bucketed, overlapped mean-all-reduce of ``.grad`` for every trainable parameter.

    After ``loss.backward()`` returns, every ``p.grad`` is a view into a flat bucket holding the
    mean over ranks, and the current stream has been ordered after the collectives -- the same
    contract DistributedDataParallel gives ``clip_grad_norm_`` / ``optimizer.step()``.

    The buckets ARE the gradient storage: at the start of a backward pass every bucket whose
    parameters all have ``grad is None`` is zero-filled once, and the kernel library's backward
    functions accumulate their weight / bias / LayerNorm gradients directly into the parameter's
    slice (``_SlotRef``), which autograd then installs as ``.grad`` without a copy.  Gradients that
    arrive any other way (stock PyTorch modules, a second use of a parameter, accumulation on top of
    a previous pass without ``zero_grad``) are copied or added into the slice as before.

    Collectives are issued strictly in bucket order (bucket k only after buckets 0..k-1), whatever
    order the buckets complete in, so ranks that differ in which parameters received a gradient still
    pair the same buffers.  A parameter that received no gradient on this rank contributes zeros; it
    ends up with the reduced gradient if any rank produced one and keeps ``grad=None`` if none did,
    which is what ``find_unused_parameters=True`` gives in the reference.
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
        self._next = 0                  # next bucket to issue
        self._hooks = []
        self.launch_log = []            # bucket indices in the order their collectives were issued (last pass)
        self.copies = 0                 # gradients that had to be copied into their slice (last pass)
        self._staged = {}               # id(bucket) -> [(slice, parameter)] waiting for the bucket's launch
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
                p._vitk_grad_slot = _SlotRef(self, b, i, p)
            self.buckets.append(b)
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ start of a backward pass
    def _begin_pass(self, trigger=None):
        """First event of a pass (a slice request or a gradient hook): decide per bucket whether its
        slices can be accumulated into, and zero those buckets with one fill each.  ``trigger`` is the
        parameter whose hook fired, if that is what opened the pass: its ``.grad`` is already set."""
        self._armed = True
        self._next = 0
        self.launch_log = []
        self.copies = 0
        torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        for b in self.buckets:
            clean = True
            for i, q in enumerate(b.params):
                g = q.grad
                if g is None:
                    continue
                if q is trigger and g.data_ptr() != b.flat.data_ptr() + 4 * b.offsets[i]:
                    continue          # a fresh tensor: it will be copied into the zeroed slice
                clean = False
                break
            b.clean = clean
            b.taken.clear()
            if clean:
                b.flat.zero_()

    def _slot_available(self, ref):
        if not self._armed:
            try:
                self._begin_pass()
            except RuntimeError:        # not inside backward(): nothing would mark the end of the pass
                self._armed = False
                return False
        b = ref.bucket
        return b.clean and ref.param.grad is None and ref.index not in b.taken and ref.index not in b.ready

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        if not self._armed:
            self._begin_pass(trigger=p)
        b, i = self._where[p]
        self._stage(b, i, p)
        if i in b.ready:          # gradient accumulated twice in one backward: the slice was refreshed
            return
        b.ready.add(i)
        b.pending -= 1
        self._launch_ready()

    def _stage(self, b, i, p):
        """A gradient that autograd left outside its slice is noted here and copied in with the rest of its
        bucket's stragglers when the bucket is issued (one multi-tensor launch instead of one per parameter:
        the task-conditioned variants have hundreds of small factors)."""
        v = b.view(i)
        if p.grad.data_ptr() != v.data_ptr():
            self._staged.setdefault(id(b), []).append((v, p))
            self.copies += 1

    def _flush_staged(self, b):
        todo = self._staged.pop(id(b), None)
        if not todo:
            return
        views, grads = [v for v, _ in todo], [p.grad.reshape(v.shape) for v, p in todo]
        if len(todo) > 1 and hasattr(torch, "_foreach_copy_"):
            torch._foreach_copy_(views, grads)
        else:
            for v, g in zip(views, grads):
                v.copy_(g)
        for v, p in todo:
            p.grad = v

    def _launch_ready(self):
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next], self._next)
            self._next += 1

    def _launch(self, b, index):
        self.launch_log.append(index)
        self._flush_staged(b)
        if self.world == 1 and self.comm is None and not self.always_reduce:
            return
        if b.pending == 0:
            b.flags.fill_(1.0)
        else:
            if b.flags_event is not None:
                b.flags_event.synchronize()          # the copy out of the buffer one pass ago (long done)
            b.flags_host.copy_(torch.tensor([1.0 if i in b.ready else 0.0 for i in range(len(b.params))],
                                            dtype=torch.float32))
            b.flags.copy_(b.flags_host, non_blocking=True)
            if b.flags_event is not None:
                b.flags_event.record()
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
        if self.backend == "gloo" and flat.is_cuda:
            # gloo moves host memory: stage through the host, on the stream this was called on (rehearsals of
            # the multi-rank path on a one-GPU box; the real transport is RCCL)
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                host.div_(self.world)
            flat.copy_(host)
            return None
        if self.average and self.backend == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work

    # ------------------------------------------------------------------ end of backward
    def finish(self):
        """Issue the buckets that are still incomplete (unused parameters), in order; wait for every
        collective; re-arm for the next pass."""
        for k in range(self._next, len(self.buckets)):
            b = self.buckets[k]
            if b.pending > 0 and not b.clean:
                for i, p in enumerate(b.params):
                    if i not in b.ready and p.grad is None:
                        b.view(i).zero_()          # (a clean bucket's untouched slices are zero already)
            self._launch(b, k)
        self._next = len(self.buckets)
        reduced = self.world > 1 or self.comm is not None or self.always_reduce
        missing = [(b, i, p) for b in self.buckets for i, p in enumerate(b.params)
                   if i not in b.ready and p.grad is None] if reduced else []
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if self.average and self.backend != "nccl":
                    b.flat.div_(self.world)
                b.work = None
            b.pending = len(b.params)
            b.ready.clear()
            b.taken.clear()
            b.clean = False
        if self.on_gpu and (self.world > 1 or self.comm is not None or self.always_reduce):
            torch.cuda.current_stream().wait_stream(self.stream)
        if missing:
            # parameters that got no gradient HERE but did on another rank take the reduced value, as
            # under DistributedDataParallel (replicas must keep stepping together); only a rank that
            # has such candidates reads the flags back (one small device-to-host copy per bucket)
            seen = {}
            for b, i, p in missing:
                if id(b) not in seen:
                    seen[id(b)] = b.flags.cpu()
                if float(seen[id(b)][i]) > 0.0:
                    p.grad = b.view(i)
        self._armed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if getattr(p, "_vitk_grad_slot", None) is not None and p._vitk_grad_slot.owner is self:
                del p._vitk_grad_slot


def broadcast_parameters(module, src=0, process_group=None):
    """Rank ``src``'s parameters and buffers to everyone (what DDP's constructor does)."""
    if get_world_size() == 1:
        return
    staged = dist.get_backend(process_group) == "gloo"      # gloo moves host memory: GPU tensors go through the host
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if staged and t.is_cuda:
                host = t.data.cpu()
                dist.broadcast(host, src, group=process_group)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src, group=process_group)
    try:        # writes through .data do not move the version counters the kept GEMM operands are checked against
        import vitk
        vitk.ops.invalidate_operands(module.parameters())
    except ImportError:
        pass


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
        if get_world_size() > 1 and any(p.is_cuda for p in module.parameters()):
            # The bucket all-reduces run on a side stream while backward is still computing and hold
            # compute units for their whole duration.  The persistent attention-backward kernel counts
            # on one workgroup per CU; split it finer so that a workgroup that has to wait for a CU
            # delays the launch by a quarter of it at most (free on an idle GPU).
            import vitk
            vitk.check(vitk.lib.vitk_attn_set_oversubscription(4), "vitk_attn_set_oversubscription")
            # likewise the persistent GEMM workgroups: four shares per CU instead of one
            vitk.check(vitk.lib.vitk_gemm_set_persistent(1024), "vitk_gemm_set_persistent")

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
