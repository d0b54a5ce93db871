"""Optimizer step of the fine-tuning loop on the HIP kernel library.

The reference loop (train.py:535-538) runs, per optimisation step,

    torch.nn.utils.clip_grad_norm_(model.parameters(), args.max_grad_norm)
    optimizer.step()            # torch.optim.SGD(lr, momentum=0.9, weight_decay) built at train.py:475-478
    scheduler.step()
    optimizer.zero_grad()

`FusedSGD` is a `torch.optim.Optimizer` (so the LambdaLR schedules of utils/scheduler.py drive it
unchanged) whose `step(max_grad_norm=...)` does the clipping and the update in two kernel launches
over a table of all tensors (`vitk_sgd_norm_sq`, `vitk_sgd_step`).  Same arithmetic as
torch.optim.SGD without Nesterov: the momentum buffer starts as the first (clipped) gradient.
There is no CPU path: the tensors must live on the GPU and the kernel library must load.
"""
import ctypes

import numpy as np
import torch

import vitk

_ENTRY = np.dtype([("param", "<u8"), ("grad", "<u8"), ("mom", "<u8"), ("n", "<i8"), ("block0", "<i4"), ("flags", "<i4")])
_CHUNK = 8192            # elements per block (kernel constant)
assert _ENTRY.itemsize == 40


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("lr, momentum and weight_decay must be non-negative")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay))
        self._slots = None
        self._dev = None          # the table on the device
        self._last = None         # its contents (numpy), to skip uploads of an unchanged table
        self._ring, self._turn = [], 0

    def _table(self, group):
        rows = []
        block0 = 0
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32
                    and p.is_contiguous() and g.is_contiguous()):
                raise ValueError("FusedSGD needs contiguous fp32 parameters and gradients on the GPU")
            st = self.state[p]
            first = 0
            if group["momentum"] != 0.0:
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                    first = 1
                m = st["momentum_buffer"].data_ptr()
            else:
                m = 0
            n = p.numel()
            vec = 2 if (n % 4 == 0 and (p.data_ptr() | g.data_ptr() | m) % 16 == 0) else 0
            rows.append((p.data_ptr(), g.data_ptr(), m, n, block0, first | vec))
            block0 += (n + _CHUNK - 1) // _CHUNK
        return rows, block0

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """One update.  With `max_grad_norm`, gradients of ALL groups are clipped by their joint
        2-norm first (clip_grad_norm_ semantics) and that norm is returned as a 0-d tensor."""
        if closure is not None:
            raise ValueError("FusedSGD does not re-evaluate the model")
        tables = [self._table(g) for g in self.param_groups]
        n_rows = sum(len(r) for r, _ in tables)
        if n_rows == 0:
            return None
        dev = next(p for g in self.param_groups for p in g["params"] if p.grad is not None).device
        multi = len([1 for r, _ in tables if r]) > 1
        need = n_rows * (2 if multi else 1)
        # rows [0, n_rows): every group, blocks numbered globally (norm pass; also the update of a
        # single group).  With several groups, rows [n_rows, 2 n_rows) repeat them numbered per group.
        host = np.zeros(need, dtype=_ENTRY)
        at, blk = 0, 0
        spans = []
        for rows, nblk in tables:
            for r in rows:
                host[at] = r[:4] + (r[4] + blk, r[5])
                if multi:
                    host[n_rows + at] = r
                at += 1
            spans.append(((n_rows if multi else 0) + at - len(rows), len(rows), nblk))
            blk += nblk
        # The caching allocator hands the same gradient addresses back step after step, so the table
        # on the device is usually still right: upload only when it changed.  Uploads go through a
        # ring of pinned buffers, each guarded by an event, so the host never waits for the GPU to
        # catch up with the step it has just enqueued.
        if self._last is None or self._last.shape != host.shape or not np.array_equal(self._last, host):
            nbytes = need * _ENTRY.itemsize
            if self._dev is None or self._dev.numel() < nbytes:
                self._dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                self._ring = []
            if not self._ring or self._ring[0][0].numel() < nbytes:
                # pinning memory synchronises the device: take all four buffers at once, with head room
                self._ring = [(torch.empty(max(2 * nbytes, 1 << 14), dtype=torch.uint8).pin_memory(), torch.cuda.Event())
                              for _ in range(4)]
                self._turn = 0
            else:
                self._ring[self._turn][1].synchronize()      # that upload was enqueued four changes ago
            buf, ev = self._ring[self._turn]
            buf.numpy()[:nbytes] = host.view(np.uint8)
            self._dev[:nbytes].copy_(buf[:nbytes], non_blocking=True)
            ev.record()
            self._turn = (self._turn + 1) % 4
            self._last = host
        if self._slots is None:
            self._slots = torch.zeros(64 + 1024, dtype=torch.float32, device=dev)      # VITK_SGD_SLOTS (include/vitk.h)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        base = self._dev.data_ptr()
        slots = None
        if max_grad_norm is not None:
            vitk.check(vitk.lib.vitk_sgd_norm_sq(base, n_rows, blk, self._slots.data_ptr(), stream), "sgd_norm_sq")
            slots = self._slots.data_ptr()
        for (first_row, cnt, nblk), group in zip(spans, self.param_groups):
            if cnt == 0:
                continue
            vitk.check(vitk.lib.vitk_sgd_step(base + first_row * _ENTRY.itemsize, cnt, nblk, slots,
                                              float(max_grad_norm or 0.0), float(group["lr"]),
                                              float(group["momentum"]), float(group["dampening"]),
                                              float(group["weight_decay"]), stream), "sgd_step")
        # the parameters were written through raw pointers: renew the GEMM operands kept with them (one launch)
        vitk.ops.refresh_operands([p for g in self.param_groups for p in g["params"] if p.grad is not None])
        if max_grad_norm is not None:
            return self._slots[0].sqrt()
        return None
