"""Host side of the hot path: thin tensor-level wrappers over the C-ABI and one
``autograd.Function`` per operator group (patch embedding, projection GEMM, fused attention,
MLP, LayerNorm + residual).  PyTorch supplies device memory, the current HIP stream and the
autograd graph only; every FLOP runs in libvitk.so.
"""
import ctypes
import math

import os

import torch

import weakref

from . import (BF16, F32, EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_PATCH,
               CastEntry, EpilogueArgs, VitkError, check, dtype_code, lib)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise VitkError("vitk kernels run on the GPU only: got a tensor on " + str(t.device))


# ----------------------------------------------------------------------------- dropout seeds
_seed_counter = 0


def next_seed():
    """A fresh 64-bit stream id per dropout site and step, derived from torch's seed."""
    global _seed_counter
    _seed_counter += 1
    s = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    return s


# Hand-over of the dropout identity of a residual-stream producer (a linear / MLP operator with a
# residual epilogue) to the LayerNorm that consumes its output, and of the masked bf16 gradient that
# LayerNorm's backward then emits back to that producer's backward.  Autograd hands tensors between
# Python backward functions as fresh wrappers, so the only handle both sides share is the address; an
# address alone identifies nothing (the caching allocator recycles it), so every producer draws a
# TICKET, unique for the life of the process, that travels with both entries and is compared on use:
#   forward:  producer notes   _DROP_OF[address of y]  = (p, seed, ticket)   and keeps the ticket in its ctx
#             LayerNorm pops it (whatever it finds) and keeps it in its ctx
#   backward: LayerNorm notes  _DZ_OF[address of dx]   = (dz, ticket)
#             producer pops by the address of its incoming gradient and uses dz only if the ticket is its own.
# A stale or recycled entry therefore fails the comparison and the producer falls back to the separate
# mask/convert pass over the gradient it was actually given.  Both tables are emptied when backward()
# ends (no dz tensor outlives its pass), and producers whose inputs need no gradient note nothing.
_DROP_OF = {}     # stream tensor address -> (p, seed, ticket) of the epilogue that wrote it
_DZ_OF = {}       # stream-gradient address -> (dz, ticket)
_ticket_counter = 0
_handover_armed = False


def _next_ticket():
    global _ticket_counter
    _ticket_counter += 1
    return _ticket_counter


def _end_of_backward_handover():
    global _handover_armed
    _DROP_OF.clear()
    _DZ_OF.clear()
    _handover_armed = False


def _note_producer(y, p, seed):
    """Called by a residual-epilogue producer in forward; returns its ticket."""
    if len(_DROP_OF) > 64:          # forwards that are never followed by a backward must not pile up
        _DROP_OF.clear()
    t = _next_ticket()
    _DROP_OF[y.data_ptr()] = (p, seed, t)
    return t


def _note_dz(dx, dz, ticket):
    """Called by LayerNorm's backward; arms the end-of-backward clean-up once per pass."""
    global _handover_armed
    if not _handover_armed:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_handover)
            _handover_armed = True
        except RuntimeError:        # not inside backward(): nothing would ever clear the entry
            return
    _DZ_OF[dx.data_ptr()] = (dz, ticket)


# ----------------------------------------------------------------------------- raw wrappers
def gemm_nt(a, b, out, epi=EPI_NONE, bias=None, residual=None, aux=None, pos=None, npatch=0,
            dropout_p=0.0, seed=0):
    """out[M,N] = epilogue(a[M,K] @ b[N,K]^T); a, b, out 2-D, unit inner stride."""
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(-1) == 1
    if a.dtype == torch.bfloat16 and K % 64:
        # the bf16 kernels consume the reduction dimension 64 deep: zero-extend odd depths (Swin-T's 96
        # channels, a 4x4x3 patch); costs a copy of both operands, only ever on such shapes
        pad = (-K) % 64
        a, b, K = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad)), K + pad
    args = EpilogueArgs(_ptr(bias), _ptr(residual), _ptr(aux), _ptr(pos),
                        residual.stride(0) if residual is not None else 0,
                        aux.stride(0) if aux is not None else 0,
                        npatch, float(dropout_p), seed)
    ldc = out.stride(-2)
    check(lib.vitk_gemm_nt(_ptr(a), _ptr(b), _ptr(out), M, N, K, a.stride(0), b.stride(0), ldc, epi,
                           ctypes.byref(args), dtype_code(a.dtype), dtype_code(out.dtype), _stream()),
          "vitk_gemm_nt")
    return out


_TN_ATOMICS = os.environ.get("VITK_TN_ATOMICS") == "1"      # diagnostic: keep the one-pass atomic write-out
_TN_WORKSPACE = {}          # device index -> the 64 MiB tensor registered with vitk_gemm_tn_set_workspace
TN_WORKSPACE_BYTES = 64 << 20


def _tn_workspace(device):
    """The weight-gradient GEMM sums the partial tiles of its token splits through a workspace (two plain passes
    instead of atomics): one 64 MiB buffer per device, allocated at the first call and kept registered."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _TN_WORKSPACE:
        with torch.cuda.device(idx):
            ws = torch.empty(TN_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
            check(lib.vitk_gemm_tn_set_workspace(_ptr(ws), TN_WORKSPACE_BYTES), "vitk_gemm_tn_set_workspace")
        _TN_WORKSPACE[idx] = ws
    return _TN_WORKSPACE[idx]


def gemm_tn_acc(a, b, c, split_k=0, colsum=None):
    """c[M,N] (fp32) += a[T,M]^T @ b[T,N]; with `colsum` (fp32 [M]) also colsum[m] += sum_t a[t,m]."""
    if a.dtype == torch.bfloat16 and a.is_cuda and not _TN_ATOMICS:
        _tn_workspace(a.device)
    T, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == T and c.shape == (M, N) and c.dtype == torch.float32
    if colsum is not None:
        assert colsum.shape == (M,) and colsum.dtype == torch.float32 and colsum.is_contiguous()
    check(lib.vitk_gemm_tn_acc(_ptr(a), _ptr(b), _ptr(c), _ptr(colsum) if colsum is not None else None, M, N, T,
                               a.stride(0), b.stride(0), c.stride(0), split_k, dtype_code(a.dtype), _stream()),
          "vitk_gemm_tn_acc")
    return c


def colsum_acc(x, out):
    T, N = x.shape
    check(lib.vitk_colsum_acc(_ptr(x), _ptr(out), T, N, x.stride(0), dtype_code(x.dtype), _stream()),
          "vitk_colsum_acc")
    return out


class _GradArena:
    """Zero-filled fp32 memory for the gradients one backward pass accumulates into (weight and bias
    gradients are summed with atomics over token slices, LayerNorm's gamma/beta partials likewise).
    A pass that finds the size of the previous one takes it all with ONE fill and hands out slices;
    the first pass, an overflow, or a call outside backward() falls back to one fill per request.
    A slice stays valid for as long as its tensor lives (it keeps the whole block alive)."""

    def __init__(self):
        self.buf, self.used, self.asked, self.want, self.armed = None, 0, 0, 0, False

    def _end_of_backward(self):
        self.want, self.buf, self.used, self.asked, self.armed = self.asked, None, 0, 0, False

    def take(self, numel, device):
        n = (numel + 63) // 64 * 64                 # slices stay 256-byte aligned
        if not self.armed:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            except RuntimeError:                    # not inside backward(): nothing marks the end of a pass
                return torch.zeros(numel, dtype=torch.float32, device=device)
            self.armed = True
        self.asked += n
        if self.buf is None and self.used == 0 and self.want >= n:
            self.buf = torch.zeros(self.want, dtype=torch.float32, device=device)
        if self.buf is None or self.buf.device != device or self.used + n > self.buf.numel():
            return torch.zeros(numel, dtype=torch.float32, device=device)
        out = self.buf[self.used:self.used + numel]
        self.used += n
        return out


_ARENA = _GradArena()


def zeros_f32(numel, device):
    """Zeroed 1-D fp32 tensor for a gradient that is accumulated into (see _GradArena)."""
    return _ARENA.take(numel, device)


def colocate_grads(*params):
    """Ask for the gradients of ``params`` to sit next to each other, in this order, in whatever flat
    bucket a gradient bucketer (utils.dist_util.GradBucketer) later builds: one GEMM then produces
    all of them in place (query / key / value)."""
    group = tuple(params)
    for p in group:
        p._vitk_colocate = group


def grad_block(params):
    """Zero fp32 storage, contiguous and in order, for the gradients of ``params`` INSIDE the flat
    bucket their gradient bucketer reduces (each parameter then carries a ``_vitk_grad_slot``), or
    None when there is no bucketer, a slice is not free in this pass (the parameter already has a
    gradient, or a second use of it took the slice), or the slices are not adjacent.  The caller lets
    its kernels accumulate into the block and returns views of it as the gradients; autograd installs
    them as ``.grad`` and the bucketer finds nothing to copy."""
    if not params:
        return None
    refs = [getattr(p, "_vitk_grad_slot", None) for p in params]
    if any(r is None for r in refs):
        return None
    flat, off, n = refs[0].flat, refs[0].offset, 0
    for r, p in zip(refs, params):
        if r.flat is not flat or r.offset != off + n or not r.available():
            return None
        n += p.numel()
    for r in refs:
        r.take()
    return flat[off:off + n]


def grad_buf(params, numel, device):
    """Accumulation target for the gradient(s) of ``params`` (``numel`` elements in all): their slice
    of the gradient bucket when there is one and it is free, else zeroed memory of the pass arena."""
    blk = grad_block(params) if params is not None else None
    if blk is not None and blk.numel() == numel:
        return blk
    return zeros_f32(numel, device)


def cast_weight(w, dtype, want=True, want_t=False):
    """fp32 master weight [R,C] -> (copy in `dtype`, transposed copy [C,R])."""
    w2 = w.detach().reshape(w.shape[0], -1)
    if not w2.is_contiguous():
        w2 = w2.contiguous()
    R, C = w2.shape
    out = torch.empty((R, C), dtype=dtype, device=w.device) if want else None
    out_t = torch.empty((C, R), dtype=dtype, device=w.device) if want_t else None
    check(lib.vitk_cast_weight(_ptr(w2), _ptr(out), _ptr(out_t), R, C, dtype_code(dtype), _stream()),
          "vitk_cast_weight")
    return out, out_t


def cast_weights_cat(ws, dtype, want_t=True):
    """Several [Ni,K] fp32 weights -> one [sum Ni, K] matrix in `dtype` (+ its transpose)."""
    _need_gpu(*ws)
    K = ws[0].shape[1]
    ns = [w.shape[0] for w in ws]
    Nt = sum(ns)
    out = torch.empty((Nt, K), dtype=dtype, device=ws[0].device)
    out_t = torch.empty((K, Nt), dtype=dtype, device=ws[0].device) if want_t else None
    if len(ws) == 1:
        w2 = ws[0].detach()
        w2 = w2 if w2.is_contiguous() else w2.contiguous()
        check(lib.vitk_cast_weight(_ptr(w2), _ptr(out), _ptr(out_t), Nt, K, dtype_code(dtype), _stream()),
              "vitk_cast_weight")
        return out, out_t
    cat = torch.cat([w.detach() for w in ws], dim=0)
    check(lib.vitk_cast_weight(_ptr(cat), _ptr(out), _ptr(out_t), Nt, K, dtype_code(dtype), _stream()),
          "vitk_cast_weight")
    return out, out_t


# ----------------------------------------------------------------------------- GEMM operands of parameters
# The GEMMs read a weight as [N, K] and as its transpose [K, N] in the compute dtype.  For weights that are
# PARAMETERS (leaves), both copies are kept with the parameter between passes: a forward pass takes them as
# they are while the parameter is unchanged, and an optimizer that writes parameters through raw pointers
# (utils.optim.FusedSGD) renews all of them with ONE launch after its update (refresh_operands) instead of
# one cast (+ one concatenation for query | key | value) per layer in every forward pass.
# "Unchanged" = same storage address and same autograd version counter: in-place tensor operations,
# load_state_dict(), load_from() and torch's optimizers bump the counter.  Writes that bypass it (through
# `.data`, or a kernel given the raw pointer) must be followed by invalidate_operands() or refresh_operands().
# Weights that are computed tensors (the task-conditioned folds) are cast per pass as before.
_OPERAND_CACHE = os.environ.get("VITK_OPERAND_CACHE", "1") != "0"
_OPERAND_VERIFY = os.environ.get("VITK_OPERAND_VERIFY") == "1"      # tests: re-derive and compare on every hit
operand_stats = {"hit": 0, "cast": 0, "refresh": 0}


class _Operand:
    __slots__ = ("refs", "versions", "ptrs", "dtype", "want", "w", "wt")

    def valid(self):
        ps = [r() for r in self.refs]
        return all(p is not None and p._version == v and p.data_ptr() == a
                   for p, v, a in zip(ps, self.versions, self.ptrs))

    def stamp(self):
        ps = [r() for r in self.refs]
        self.versions = tuple(p._version for p in ps)
        self.ptrs = tuple(p.data_ptr() for p in ps)


def _is_param(p):
    return isinstance(p, torch.nn.Parameter) and p.is_leaf and p.dtype == torch.float32 and p.is_contiguous()


def _cast_group(ws, dtype, want, want_t):
    """[Ni, K] fp32 weights -> ([sum Ni, K] in dtype or None, [K, sum Ni] or None), one launch."""
    K = ws[0].shape[1]
    ns = [w.shape[0] for w in ws]
    Nt = sum(ns)
    dev = ws[0].device
    out = torch.empty((Nt, K), dtype=dtype, device=dev) if want else None
    out_t = torch.empty((K, Nt), dtype=dtype, device=dev) if want_t else None
    _cast_launch([(ws, out, out_t)], dtype)
    return out, out_t


def _cast_launch(jobs, dtype):
    """jobs: (weights, out [sum Ni, K] or None, out_t [K, sum Ni] or None): every weight of every job in one launch
    per 64 weights."""
    n = sum(len(ws) for ws, _, _ in jobs)
    tab = (CastEntry * n)()
    esz = torch.empty((), dtype=dtype).element_size()
    i = 0
    for ws, out, out_t in jobs:
        row = 0
        Nt = sum(w.shape[0] for w in ws)
        for w in ws:
            R = w.shape[0]
            C = w.numel() // R
            e = tab[i]
            e.w = w.data_ptr()
            e.out = out.data_ptr() + row * C * esz if out is not None else None
            e.out_t = out_t.data_ptr() + row * esz if out_t is not None else None
            e.rows, e.cols, e.ld_t, e.tile0 = R, C, Nt, 0
            row += R
            i += 1
    check(lib.vitk_cast_weights_batched(tab, n, dtype_code(dtype), _stream()), "vitk_cast_weights_batched")


def weight_operands(params, dtype, want=True, want_t=True):
    """(w [sum Ni, K], wt [K, sum Ni]) in `dtype` for the GEMMs that read the fp32 weights `params` (several
    weights sharing their input are concatenated along N).  Kept with the parameters between passes (above)."""
    cacheable = _OPERAND_CACHE and all(_is_param(p) for p in params)
    if not cacheable:
        operand_stats["cast"] += 1
        return _cast_group([_f32(p).reshape(p.shape[0], -1) for p in params], dtype, want, want_t)
    head = params[0]
    table = head.__dict__.setdefault("_vitk_operands", {})
    key = (tuple(id(p) for p in params[1:]), dtype)
    ent = table.get(key)
    if ent is not None and ent.valid() and (ent.w is not None or not want) and (ent.wt is not None or not want_t):
        operand_stats["hit"] += 1
        if _OPERAND_VERIFY:
            w, wt = _cast_group([p.detach().reshape(p.shape[0], -1) for p in params], dtype, want, want_t)
            if (want and not torch.equal(w, ent.w)) or (want_t and not torch.equal(wt, ent.wt)):
                raise VitkError("stale GEMM operand of a parameter: it was written without its version counter "
                                "moving; call vitk.ops.invalidate_operands() after such writes")
        return (ent.w if want else None), (ent.wt if want_t else None)
    operand_stats["cast"] += 1
    w, wt = _cast_group([p.detach().reshape(p.shape[0], -1) for p in params], dtype, want, want_t)
    ent = _Operand()
    ent.refs = tuple(weakref.ref(p) for p in params)
    ent.dtype, ent.want, ent.w, ent.wt = dtype, (want, want_t), w, wt
    ent.stamp()
    table[key] = ent
    return w, wt


def invalidate_operands(params=None):
    """Forget the kept GEMM operands of `params` (an iterable of parameters / a module's .parameters()).
    Needed only after writes that do not move the autograd version counter."""
    for p in params:
        p.__dict__.pop("_vitk_operands", None)


def refresh_operands(params):
    """Renew, with one launch per 64 weights, every kept GEMM operand whose first weight is in `params`: called by
    an optimizer that has just updated them through raw pointers.  The renewed copies are new tensors: a graph
    that still holds the old ones keeps reading the values its forward pass used."""
    by_dtype = {}
    for p in params:
        table = p.__dict__.get("_vitk_operands")
        if not table:
            continue
        for key, ent in list(table.items()):
            ps = [r() for r in ent.refs]
            if any(q is None or not q.is_cuda for q in ps):
                del table[key]
                continue
            by_dtype.setdefault(ent.dtype, []).append((ent, ps))
    for dtype, items in by_dtype.items():
        jobs = []
        for ent, ps in items:
            K = ps[0][0].numel() if ps[0].dim() > 1 else 1
            Nt = sum(q.shape[0] for q in ps)
            dev = ps[0].device
            ent.w = torch.empty((Nt, K), dtype=dtype, device=dev) if ent.want[0] else None
            ent.wt = torch.empty((K, Nt), dtype=dtype, device=dev) if ent.want[1] else None
            jobs.append(([q.detach().reshape(q.shape[0], -1) for q in ps], ent.w, ent.wt))
        _cast_launch(jobs, dtype)
        for ent, _ in items:
            ent.stamp()
        operand_stats["refresh"] += 1


def convert(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    n = x.numel()
    if n % 4:
        raise VitkError("vitk_convert needs a multiple of 4 elements")
    check(lib.vitk_convert(_ptr(x), _ptr(out), n, dtype_code(x.dtype), dtype_code(dtype), _stream()),
          "vitk_convert")
    return out


def to_operand(dy2, dtype, p=0.0, seed=0, ticket=0):
    """[rows, cols] gradient -> GEMM operand in `dtype`, applying the forward dropout mask when p > 0
    (one fused pass: mask, 1/keep scale and fp32 -> bf16 hand-over).  When the LayerNorm backward that
    produced `dy2` already emitted this operand for this very producer (same ticket), it is taken
    from there."""
    hit = _DZ_OF.pop(dy2.data_ptr(), None) if ticket else None
    if hit is not None and hit[1] == ticket and hit[0].dtype == dtype and hit[0].shape == dy2.shape:
        return hit[0]
    if p > 0:
        out = torch.empty(dy2.shape, dtype=dtype, device=dy2.device)
        rows, cols = dy2.shape
        check(lib.vitk_dropout_bwd(_ptr(dy2), _ptr(out), rows, cols, float(p), seed, dtype_code(dy2.dtype),
                                   dtype_code(dtype), _stream()), "vitk_dropout_bwd")
        return out
    return convert(dy2, dtype)


def _f32(p):
    t = p.detach()
    if t.dtype != torch.float32:
        raise VitkError("parameters must be fp32 master copies")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- LayerNorm (+ residual pass-through)
class AddNormFn(torch.autograd.Function):
    """y = LayerNorm(x); also returns x itself for the residual branch so that backward can
    fuse  dx = LN'(dy) + d_residual  into one kernel (reference Block.forward,
    models/ViT/modeling.py:179-187)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype):
        _need_gpu(x, gamma, beta)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows, cols = x2.shape
        if cols > 2048 and any(ctx.needs_input_grad[:3]):
            raise VitkError(f"LayerNorm backward handles rows of up to 2048 elements, got {cols}")
        out_dtype = out_dtype or x.dtype
        y = torch.empty((rows, cols), dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g, b = _f32(gamma), _f32(beta)
        check(lib.vitk_layernorm_fwd(_ptr(x2), _ptr(g), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), rows, cols,
                                     float(eps), dtype_code(x2.dtype), dtype_code(out_dtype), _stream()),
              "vitk_layernorm_fwd")
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.xshape = x.shape
        ctx.gparams = (gamma, beta)
        # whoever consumes a noted stream tensor takes the note with it (used or not)
        producer = _DROP_OF.pop(x2.data_ptr(), None)
        ctx.producer = producer if (x2.dtype == torch.float32 and out_dtype == torch.bfloat16) else None
        return x.view_as(x), y.view(x.shape)

    @staticmethod
    def backward(ctx, dh, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        rows, cols = x2.shape
        if dy is None:
            return dh, None, None, None, None
        dy2 = dy.reshape(rows, cols)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = None
        if dh is not None:
            dres = dh.reshape(rows, cols)
            if dres.dtype != x2.dtype:
                dres = dres.to(x2.dtype)
            if not dres.is_contiguous():
                dres = dres.contiguous()
        dx = torch.empty_like(x2)
        pg, pb = ctx.gparams
        if getattr(pg, "_vitk_grad_slot", None) is not None or getattr(pb, "_vitk_grad_slot", None) is not None:
            dgamma, dbeta = grad_buf([pg], cols, x2.device), grad_buf([pb], cols, x2.device)
        else:
            dgb = zeros_f32(2 * cols, x2.device)
            dgamma, dbeta = dgb[:cols], dgb[cols:]
        ws = torch.empty(lib.vitk_layernorm_bwd_workspace(rows, cols), dtype=torch.uint8, device=x2.device)
        if ctx.producer is not None:
            # dx is also the output gradient of the layer that produced x: emit its masked bf16 operand now
            p, seed, ticket = ctx.producer
            dz = torch.empty((rows, cols), dtype=torch.bfloat16, device=x2.device)
            check(lib.vitk_layernorm_bwd_dz(_ptr(dy2), _ptr(x2), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(dres),
                                            _ptr(dx), _ptr(dz), float(p), seed, _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                                            rows, cols, dtype_code(x2.dtype), dtype_code(dy2.dtype), _stream()),
                  "vitk_layernorm_bwd_dz")
            _note_dz(dx, dz, ticket)
        else:
            check(lib.vitk_layernorm_bwd(_ptr(dy2), _ptr(x2), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx),
                                         _ptr(dgamma), _ptr(dbeta), _ptr(ws), rows, cols, dtype_code(x2.dtype),
                                         dtype_code(dy2.dtype), _stream()), "vitk_layernorm_bwd")
        return dx.view(ctx.xshape), dgamma, dbeta, None, None


def add_norm(x, gamma, beta, eps=1e-6, out_dtype=None):
    return AddNormFn.apply(x, gamma, beta, eps, out_dtype)


def layer_norm(x, gamma, beta, eps=1e-6, out_dtype=None):
    return AddNormFn.apply(x, gamma, beta, eps, out_dtype)[1]


# ----------------------------------------------------------------------------- projection GEMM
class LinearFn(torch.autograd.Function):
    """y = x @ cat(W_i)^T + cat(b_i) [+ residual]  (reference nn.Linear q/k/v/out,
    models/ViT/modeling.py:58-62,74-76,92 and the residual add `:182`).  Several weights
    sharing the input are evaluated as ONE GEMM; parameters stay separate tensors."""

    @staticmethod
    def forward(ctx, x, residual, dropout_p, nw, out_dtype, *params):
        ws, bs = params[:nw], params[nw:]
        _need_gpu(x, residual, *params)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T, K = x2.shape
        w, wt = weight_operands(list(ws), x2.dtype)
        bias = _f32(bs[0]) if nw == 1 else torch.cat([_f32(b) for b in bs])
        N = w.shape[0]
        out_dtype = residual.dtype if residual is not None else (out_dtype or x2.dtype)
        y = torch.empty((T, N), dtype=out_dtype, device=x.device)
        seed = next_seed() if dropout_p > 0 else 0
        ticket = 0
        if residual is not None:
            r2 = residual.reshape(T, N)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
            gemm_nt(x2, w, y, EPI_BIAS_RES, bias=bias, residual=r2, dropout_p=dropout_p, seed=seed)
            if any(ctx.needs_input_grad):
                ticket = _note_producer(y, dropout_p, seed)
        else:
            gemm_nt(x2, w, y, EPI_BIAS, bias=bias, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(x2, wt)
        ctx.gparams = (list(ws), list(bs))
        ctx.meta = (x.shape, [p.shape[0] for p in ws], dropout_p, seed,
                    residual.shape if residual is not None else None, nw, ticket)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wt = ctx.saved_tensors
        xshape, ns, p, seed, res_shape, nw, ticket = ctx.meta
        T, K = x2.shape
        N = wt.shape[1]
        dy2 = dy.reshape(T, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = dy.reshape(res_shape) if res_shape is not None else None
        dz = to_operand(dy2, x2.dtype, p, seed, ticket)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((T, K), dtype=x2.dtype, device=x2.device)
            gemm_nt(dz, wt, dx, EPI_NONE)
            dx = dx.view(xshape)
        need_w = any(ctx.needs_input_grad[5:5 + nw])
        need_b = any(ctx.needs_input_grad[5 + nw:5 + 2 * nw])
        dws = [None] * nw
        dbs = [None] * nw
        if need_w or need_b:
            # accumulation targets: the parameters' own slices of the gradient bucket when a bucketer
            # owns them (all of the fused weights live, slices adjacent), else zeroed arena memory
            pws, pbs = ctx.gparams
            dWb = grad_block(pws) if need_w and all(ctx.needs_input_grad[5:5 + nw]) else None
            dbb = grad_block(pbs) if need_b and all(ctx.needs_input_grad[5 + nw:5 + 2 * nw]) else None
            if dWb is not None and dWb.numel() != N * K:
                dWb = None
            if dbb is not None and dbb.numel() != N:
                dbb = None
            if dWb is None and dbb is None:
                flat = zeros_f32(N * K + N, x2.device)      # dW | db
                dWb, dbb = flat[:N * K], flat[N * K:]
            else:
                dWb = dWb if dWb is not None else (zeros_f32(N * K, x2.device) if need_w else None)
                dbb = dbb if dbb is not None else (zeros_f32(N, x2.device) if need_b else None)
            db = dbb if need_b else None
            if need_w:
                dW = dWb.view(N, K)
                gemm_tn_acc(dz, x2, dW, colsum=db)         # db rides along with dW
                dws = list(dW.split(ns, dim=0))
            elif need_b:
                colsum_acc(dz, db)
            if need_b:
                dbs = list(db.split(ns, dim=0))
        return (dx, dres, None, None, None, *dws, *dbs)


def linear(x, weights, biases, residual=None, dropout_p=0.0, out_dtype=None):
    """``out_dtype`` (without a residual): type of the result, e.g. torch.float32 out of bf16 operands."""
    return LinearFn.apply(x, residual, float(dropout_p), len(weights), out_dtype, *weights, *biases)


# ----------------------------------------------------------------------------- small fp32 products of parameters
class MatmulF32Fn(torch.autograd.Function):
    """C = A . B for fp32 matrices whose sides are multiples of 4, on the exact-fp32 MFMA GEMM of the kernel
    library (forward and both gradients): the weight-sized products of the task-conditioned variants
    (rank-K factor products, filter-bank mixing; reference models/ViT/layers_PB.py:253-297,
    models/ViT/modeling_PBG.py:67-69) without a BLAS call."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        a, b = _f32(a), _f32(b)
        M, K = a.shape
        N = b.shape[1]
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        gemm_nt(a, b.t().contiguous(), out, EPI_NONE)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc if dc.is_contiguous() else dc.contiguous()
        M, K = a.shape
        N = b.shape[1]
        da = db = None
        if ctx.needs_input_grad[0]:
            dat = torch.empty((K, M), dtype=torch.float32, device=a.device)     # dA^T = B . dC^T
            gemm_nt(b, dc, dat, EPI_NONE)
            da = dat.t()
        if ctx.needs_input_grad[1]:
            db = torch.zeros((K, N), dtype=torch.float32, device=a.device)      # dB = A^T . dC
            gemm_tn_acc(a, dc, db)
        return da, db


def matmul_f32(a, b):
    """a [M, K] @ b [K, N] in fp32 on the kernel library; sides that are not multiples of 4 are zero-extended
    (torch pads and slices: a few weight-sized elements, differentiable)."""
    M, K = a.shape
    N = b.shape[1]
    pm, pk, pn = (-M) % 4, (-K) % 4, (-N) % 4
    if pm or pk:
        a = torch.nn.functional.pad(a, (0, pk, 0, pm))
    if pk or pn:
        b = torch.nn.functional.pad(b, (0, pn, 0, pk))
    out = MatmulF32Fn.apply(a, b)
    return out[:M, :N] if (pm or pn) else out


# ----------------------------------------------------------------------------- fused attention
class AttentionFn(torch.autograd.Function):
    """ctx = softmax(q.k^T / sqrt(dh)).v on a fused [T, 3*D] projection buffer
    (reference models/ViT/modeling.py:78-91).  Returns (context [B,N,D], lse [B,H,N]).  ``dropout_p`` > 0: the
    probabilities are dropped and rescaled between the softmax and the product with v (`:86`), with a mask
    that backward regenerates from the same seed."""

    @staticmethod
    def forward(ctx, qkv, B, N, H, dropout_p=0.0):
        _need_gpu(qkv)
        T, D3 = qkv.shape
        D = D3 // 3
        dh = D // H
        assert T == B * N and qkv.is_contiguous()
        out = torch.empty((T, D), dtype=qkv.dtype, device=qkv.device)
        # unrounded copy of the context for delta = rowsum(dO * O) in backward (bf16 instance only)
        keep32 = qkv.dtype != torch.float32 and ctx.needs_input_grad[0]
        o32 = torch.empty((T, D), dtype=torch.float32, device=qkv.device) if keep32 else None
        lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        scale = 1.0 / math.sqrt(dh)
        esz = qkv.element_size()
        base = qkv.data_ptr()
        dropout_p = float(dropout_p)
        seed = next_seed() if dropout_p > 0 else 0
        check(lib.vitk_attn_fwd_dropout(ctypes.c_void_p(base), ctypes.c_void_p(base + D * esz),
                                        ctypes.c_void_p(base + 2 * D * esz), _ptr(out), _ptr(o32), _ptr(lse), B, H, N, dh,
                                        D3, D, scale, dropout_p, seed, dtype_code(qkv.dtype), _stream()),
              "vitk_attn_fwd_dropout")
        ctx.save_for_backward(qkv, out, lse, o32)
        ctx.meta = (B, N, H, dh, scale, dropout_p, seed)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        qkv, out, lse, o32 = ctx.saved_tensors
        B, N, H, dh, scale, dropout_p, seed = ctx.meta
        T, D3 = qkv.shape
        D = D3 // 3
        dout = dout.reshape(T, D)
        if dout.dtype != qkv.dtype:
            dout = convert(dout, qkv.dtype)
        if not dout.is_contiguous():
            dout = dout.contiguous()
        if qkv.dtype == torch.float32:
            dqkv = torch.zeros_like(qkv)     # fp32 instance accumulates dk/dv atomically
        else:
            dqkv = torch.empty_like(qkv)
        delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        esz = qkv.element_size()
        b0, d0 = qkv.data_ptr(), dqkv.data_ptr()
        vp = ctypes.c_void_p
        check(lib.vitk_attn_bwd_dropout(vp(b0), vp(b0 + D * esz), vp(b0 + 2 * D * esz), _ptr(out), _ptr(o32), _ptr(dout),
                                        _ptr(lse), _ptr(delta), vp(d0), vp(d0 + D * esz), vp(d0 + 2 * D * esz), B, H, N, dh,
                                        D3, D, D3, scale, dropout_p, seed, dtype_code(qkv.dtype), _stream()),
              "vitk_attn_bwd_dropout")
        return dqkv, None, None, None, None


def attention(qkv, B, N, H, dropout_p=0.0):
    return AttentionFn.apply(qkv, B, N, H, dropout_p)


def attention_probs(qkv, lse, B, N, H):
    """softmax probabilities [B,H,N,N] (fp32) for `vis=True`; not differentiable."""
    T, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    probs = torch.empty((B, H, N, N), dtype=torch.float32, device=qkv.device)
    esz = qkv.element_size()
    base = qkv.data_ptr()
    check(lib.vitk_attn_probs(ctypes.c_void_p(base), ctypes.c_void_p(base + D * esz), _ptr(lse), _ptr(probs),
                              B, H, N, dh, D3, 1.0 / math.sqrt(dh), dtype_code(qkv.dtype), _stream()),
          "vitk_attn_probs")
    return probs


# ----------------------------------------------------------------------------- Swin window attention
class WindowAttentionFn(torch.autograd.Function):
    """softmax(scale * q.k^T + relative-position bias + shift mask) . v per (window, head)
    (reference models/Swin/swin_RKR.py:185-209, with the roll / window_partition / window_reverse /
    roll of SwinTransformerBlock.forward `:316-336` and its mask `:281-304` folded into the kernel's
    addressing).  ``qkv`` [B*H*W, 3*heads*32]: the qkv projection of the un-shifted token sequence;
    returns [B*H*W, heads*32] in the same token order."""

    @staticmethod
    def forward(ctx, qkv, table, B, H, W, heads, window, shift, scale):
        _need_gpu(qkv, table)
        T, C3 = qkv.shape
        C = C3 // 3
        dh = C // heads
        if T != B * H * W or C3 != 3 * heads * dh:
            raise VitkError(f"window attention: qkv {tuple(qkv.shape)} does not match B={B} H={H} W={W} heads={heads}")
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        tab = _f32(table)
        out = torch.empty((T, C), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((T * heads,), dtype=torch.float32, device=qkv.device)
        check(lib.vitk_window_attn_fwd(_ptr(qkv), _ptr(tab), _ptr(out), _ptr(lse), B, H, W, heads, dh, window, shift,
                                       float(scale), dtype_code(qkv.dtype), _stream()), "vitk_window_attn_fwd")
        ctx.save_for_backward(qkv, tab, out, lse)
        ctx.meta = (B, H, W, heads, dh, window, shift, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tab, out, lse = ctx.saved_tensors
        B, H, W, heads, dh, window, shift, scale = ctx.meta
        dout = dout if dout.is_contiguous() else dout.contiguous()
        if dout.dtype != qkv.dtype:
            dout = convert(dout, qkv.dtype)
        dqkv = torch.empty_like(qkv)
        reps = 64                 # copies of the table gradient the workgroups spread their atomic flushes over
        dtab = zeros_f32(reps * tab.numel(), qkv.device).view(reps, *tab.shape)
        check(lib.vitk_window_attn_bwd(_ptr(qkv), _ptr(tab), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _ptr(dtab),
                                       reps, B, H, W, heads, dh, window, shift, scale, dtype_code(qkv.dtype),
                                       _stream()), "vitk_window_attn_bwd")
        return dqkv, (dtab.sum(0) if ctx.needs_input_grad[1] else None), None, None, None, None, None, None, None


def window_attention(qkv, table, B, H, W, heads, window, shift, scale=None):
    if scale is None:
        scale = (qkv.shape[1] // (3 * heads)) ** -0.5
    return WindowAttentionFn.apply(qkv, table, B, H, W, heads, window, shift, scale)


# ----------------------------------------------------------------------------- RKR weight fold
class RkrFoldFn(torch.autograd.Function):
    """(W_t, b_t) = ((W + (LM . RM)^T) * F[:, None], b * F): the task-conditioned weight operand of an
    RG_FC (+ SFG_FC) layer (reference models/ViT/modeling_RKR.py:56-78) in one launch, its five
    gradients in two."""

    @staticmethod
    def forward(ctx, W, b, LM, RM, F):
        _need_gpu(W, b, LM, RM, F)
        W, b, LM, RM = _f32(W), _f32(b), _f32(LM), _f32(RM)
        Fv = _f32(F) if F is not None else None
        O, I = W.shape
        K = LM.shape[1]
        Wt, bt = torch.empty_like(W), torch.empty_like(b)
        check(lib.vitk_rkr_fold_fwd(_ptr(W), _ptr(b), _ptr(LM), _ptr(RM), _ptr(Fv), _ptr(Wt), _ptr(bt), O, I, K, _stream()),
              "vitk_rkr_fold_fwd")
        ctx.save_for_backward(W, b, LM, RM, Fv)
        return Wt, bt

    @staticmethod
    def backward(ctx, dWt, dbt):
        W, b, LM, RM, Fv = ctx.saved_tensors
        O, I = W.shape
        K = LM.shape[1]
        dWt = torch.zeros_like(W) if dWt is None else (dWt if dWt.is_contiguous() else dWt.contiguous())
        dbt = torch.zeros_like(b) if dbt is None else (dbt if dbt.is_contiguous() else dbt.contiguous())
        dW = torch.empty_like(W) if ctx.needs_input_grad[0] else None
        db, dRM = torch.empty_like(b), torch.empty_like(RM)
        dLM = zeros_f32(LM.numel(), W.device).view(LM.shape)
        dF = torch.empty_like(Fv) if Fv is not None else None
        check(lib.vitk_rkr_fold_bwd(_ptr(W), _ptr(b), _ptr(LM), _ptr(RM), _ptr(Fv), _ptr(dWt), _ptr(dbt), _ptr(dW), _ptr(db),
                                    _ptr(dLM), _ptr(dRM), _ptr(dF), O, I, K, _stream()), "vitk_rkr_fold_bwd")
        return dW, db, dLM, dRM, dF


def rkr_fold(W, b, LM, RM, F=None):
    return RkrFoldFn.apply(W, b, LM, RM, F)


class MaskFoldFn(torch.autograd.Function):
    """out = hard(mask_real) * w: a piggyback mask applied to a weight, a factor or a scale (reference
    models/ViT/layers_PB.py:14-54 and :105-111) in one launch; both straight-through gradients in one."""

    @staticmethod
    def forward(ctx, mask_real, w, ternary, threshold):
        _need_gpu(mask_real, w)
        if mask_real.shape != w.shape:
            raise VitkError("mask_fold: the mask and the tensor it masks differ in shape")
        m, w = _f32(mask_real), _f32(w)
        out = torch.empty_like(w)
        check(lib.vitk_mask_fold_fwd(_ptr(m), _ptr(w), _ptr(out), w.numel(), int(ternary), float(threshold), _stream()),
              "vitk_mask_fold_fwd")
        ctx.save_for_backward(m, w)
        ctx.args = (int(ternary), float(threshold))
        return out

    @staticmethod
    def backward(ctx, g):
        m, w = ctx.saved_tensors
        need_m, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_m or need_w):
            return None, None, None, None
        g = g if g.is_contiguous() else g.contiguous()
        dm = torch.empty_like(m) if need_m else None
        dw = torch.empty_like(w) if need_w else None
        check(lib.vitk_mask_fold_bwd(_ptr(m), _ptr(w), _ptr(g), _ptr(dm), _ptr(dw), w.numel(), ctx.args[0], ctx.args[1],
                                     _stream()), "vitk_mask_fold_bwd")
        return dm, dw, None, None


def mask_fold(mask_real, w, ternary=False, threshold=5e-3):
    return MaskFoldFn.apply(mask_real, w, ternary, threshold)


# ----------------------------------------------------------------------------- stochastic depth
class ScaleAddFn(torch.autograd.Function):
    """out = h + scale[sample] * y: a residual branch kept or dropped per sample (timm's DropPath as used by
    reference models/Swin/swin_RKR.py:340-341) added to the fp32 stream in one pass."""

    @staticmethod
    def forward(ctx, h, y, scale):
        _need_gpu(h, y, scale)
        if h.dtype != torch.float32 or scale.dtype != torch.float32 or h.shape != y.shape or scale.numel() != h.shape[0]:
            raise VitkError("scale_add: h fp32 [S, ...], y same shape, scale fp32 [S]")
        h = h if h.is_contiguous() else h.contiguous()
        y = y if y.is_contiguous() else y.contiguous()
        scale = scale.contiguous()
        out = torch.empty_like(h)
        S = h.shape[0]
        check(lib.vitk_scale_add(_ptr(h), _ptr(y), _ptr(scale), _ptr(out), S, h.numel() // S, dtype_code(y.dtype),
                                 _stream()), "vitk_scale_add")
        ctx.save_for_backward(scale)
        ctx.ydtype = y.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (scale,) = ctx.saved_tensors
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dy = None
        if ctx.needs_input_grad[1]:
            dy = torch.empty(dout.shape, dtype=ctx.ydtype, device=dout.device)
            S = dout.shape[0]
            check(lib.vitk_scale_cast(_ptr(dout), _ptr(scale), _ptr(dy), S, dout.numel() // S, dtype_code(ctx.ydtype),
                                      _stream()), "vitk_scale_cast")
        return (dout if ctx.needs_input_grad[0] else None), dy, None


def scale_add(h, y, scale):
    return ScaleAddFn.apply(h, y, scale)


# ----------------------------------------------------------------------------- MLP
class MlpFn(torch.autograd.Function):
    """y = drop(fc2(drop(gelu(fc1(x))))) [+ residual]  (reference Mlp.forward,
    models/ViT/modeling.py:113-119, exact erf GELU `:47`, residual add `:187`).  Two GEMMs:
    bias+GELU(+dropout) fused in the first epilogue, which also leaves gelu' x dropout factor behind
    (instead of the pre-activation), bias(+dropout)+residual in the second; backward multiplies the dX
    GEMM of fc2 with that factor in its epilogue."""

    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, dropout_p):
        _need_gpu(x, residual, w1, b1, w2, b2)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T, D = x2.shape
        F_, Do = w1.shape[0], w2.shape[0]                # Do == D in the transformer blocks; D may be zero-padded
        dt = x2.dtype
        w1c, w1t = weight_operands([w1], dt)      # [F,D], [D,F]
        w2c, w2t = weight_operands([w2], dt)      # [D,F], [F,D]
        u = torch.empty((T, F_), dtype=dt, device=x.device)      # gelu'(pre-activation) - 1/2 (-1/2: dropped), for backward
        g = torch.empty((T, F_), dtype=dt, device=x.device)
        y = torch.empty((T, Do), dtype=residual.dtype if residual is not None else dt, device=x.device)
        s1 = next_seed() if dropout_p > 0 else 0
        s2 = next_seed() if dropout_p > 0 else 0
        ticket = 0
        gemm_nt(x2, w1c, g, EPI_BIAS_GELU, bias=_f32(b1), aux=u, dropout_p=dropout_p, seed=s1)
        if residual is not None:
            r2 = residual.reshape(T, Do)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
            gemm_nt(g, w2c, y, EPI_BIAS_RES, bias=_f32(b2), residual=r2, dropout_p=dropout_p, seed=s2)
            if any(ctx.needs_input_grad):
                ticket = _note_producer(y, dropout_p, s2)
        else:
            gemm_nt(g, w2c, y, EPI_BIAS, bias=_f32(b2), dropout_p=dropout_p, seed=s2)
        ctx.save_for_backward(x2, u, g, w1t, w2t)
        ctx.gparams = (w1, b1, w2, b2)
        ctx.meta = (x.shape, dropout_p, s1, s2, residual.shape if residual is not None else None, ticket)
        return y.view(*x.shape[:-1], Do)

    @staticmethod
    def backward(ctx, dy):
        x2, u, g, w1t, w2t = ctx.saved_tensors
        xshape, p, s1, s2, res_shape, ticket = ctx.meta
        T, D = x2.shape
        F_, Do = u.shape[1], w2t.shape[1]
        dy2 = dy.reshape(T, Do)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = dy.reshape(res_shape) if res_shape is not None else None
        dz = to_operand(dy2, x2.dtype, p, s2, ticket)
        # du = (dz @ W2) * mask1/keep * gelu'(pre-activation), mask and derivative as the forward epilogue left them
        du = torch.empty((T, F_), dtype=x2.dtype, device=x2.device)
        gemm_nt(dz, w2t, du, EPI_DGELU, aux=u, dropout_p=p)
        dx = dw1 = db1 = dw2 = db2 = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((T, D), dtype=x2.dtype, device=x2.device)
            gemm_nt(du, w1t, dx, EPI_NONE)
            dx = dx.view(xshape)
        if any(ctx.needs_input_grad[2:6]):
            pw1, pb1, pw2, pb2 = ctx.gparams
            need = ctx.needs_input_grad
            if any(getattr(q, "_vitk_grad_slot", None) is not None for q in ctx.gparams):
                # a gradient bucketer owns (some of) the parameters: accumulate into their slices
                bw1 = grad_buf([pw1], F_ * D, x2.device) if need[2] else None
                db1 = grad_buf([pb1], F_, x2.device) if need[3] else None
                bw2 = grad_buf([pw2], Do * F_, x2.device) if need[4] else None
                db2 = grad_buf([pb2], Do, x2.device) if need[5] else None
            else:
                flat = zeros_f32(F_ * D + F_ + Do * F_ + Do, x2.device)  # dW1 | db1 | dW2 | db2
                o1, o2, o3 = F_ * D, F_ * D + F_, F_ * D + F_ + Do * F_
                bw1, bw2 = flat[:o1], flat[o2:o3]
                db1 = flat[o1:o2] if need[3] else None
                db2 = flat[o3:] if need[5] else None
            if need[2]:
                dw1 = bw1.view(F_, D)
                gemm_tn_acc(du, x2, dw1, colsum=db1)       # bias gradients ride along with dW
            elif db1 is not None:
                colsum_acc(du, db1)
            if need[4]:
                dw2 = bw2.view(Do, F_)
                gemm_tn_acc(dz, g, dw2, colsum=db2)
            elif db2 is not None:
                colsum_acc(dz, db2)
        return dx, dres, dw1, db1, dw2, db2, None


def mlp(x, w1, b1, w2, b2, residual=None, dropout_p=0.0):
    return MlpFn.apply(x, residual, w1, b1, w2, b2, float(dropout_p))


# ----------------------------------------------------------------------------- patch embedding
class PatchEmbedFn(torch.autograd.Function):
    """Token assembly of reference Embeddings.forward (models/ViT/modeling.py:153-166):
    Conv2d(kernel=stride=P) as im2col + GEMM whose epilogue adds bias and position embedding
    and scatters past the class-token row; class row and dropout share the same mask stream."""

    @staticmethod
    def forward(ctx, img, conv_w, conv_b, cls, pos, dropout_p, dtype):
        _need_gpu(img, conv_w, conv_b, cls, pos)
        B, C, Hi, Wi = img.shape
        D, _, P, P2 = conv_w.shape
        assert P == P2
        gh, gw = Hi // P, Wi // P
        n = gh * gw
        if pos.shape[1] != n + 1:
            raise VitkError(f"position embeddings hold {pos.shape[1]} tokens, image gives {n + 1}")
        imgc = img.detach()
        if imgc.dtype != torch.float32:
            imgc = imgc.float()
        if not imgc.is_contiguous():
            imgc = imgc.contiguous()
        K = C * P * P
        cols = torch.empty((B * n, K), dtype=dtype, device=img.device)
        check(lib.vitk_im2col(_ptr(imgc), _ptr(cols), B, C, Hi, Wi, P, dtype_code(dtype), _stream()),
              "vitk_im2col")
        wc, _ = weight_operands([conv_w], dtype, want=True, want_t=False)
        out = torch.empty((B, n + 1, D), dtype=torch.float32, device=img.device)   # residual stream
        seed = next_seed() if dropout_p > 0 else 0
        posf, clsf = _f32(pos).view(n + 1, D), _f32(cls).view(D)
        gemm_nt(cols, wc, out.view(B * (n + 1), D), EPI_PATCH, bias=_f32(conv_b), pos=posf, npatch=n,
                dropout_p=dropout_p, seed=seed)
        check(lib.vitk_embed_cls(_ptr(clsf), _ptr(posf), _ptr(out), B, n + 1, D, float(dropout_p), seed,
                                 F32, _stream()), "vitk_embed_cls")
        ctx.save_for_backward(cols)
        ctx.gparams = (conv_w, conv_b, cls, pos)
        ctx.meta = (B, n, D, K, conv_w.shape, dropout_p, seed)
        return out

    @staticmethod
    def backward(ctx, dE):
        (cols,) = ctx.saved_tensors
        B, n, D, K, wshape, p, seed = ctx.meta
        dE = dE.contiguous()
        if dE.dtype != torch.float32:
            dE = dE.float()
        dev = dE.device
        dtok = torch.empty((B * n, D), dtype=cols.dtype, device=dev)
        pw, pb, pcls, ppos = ctx.gparams
        dpos = grad_buf([ppos] if ctx.needs_input_grad[4] else None, (n + 1) * D, dev).view(n + 1, D)
        dcls = grad_buf([pcls] if ctx.needs_input_grad[3] else None, D, dev)
        check(lib.vitk_embed_bwd(_ptr(dE), _ptr(dtok), _ptr(dpos), _ptr(dcls), B, n + 1, D, float(p), seed,
                                 F32, dtype_code(dtok.dtype), _stream()), "vitk_embed_bwd")
        dw = db = None
        if ctx.needs_input_grad[2]:
            db = grad_buf([pb], D, dev)
        if ctx.needs_input_grad[1]:
            dw = grad_buf([pw], D * K, dev).view(D, K)
            gemm_tn_acc(dtok, cols, dw, colsum=db)
            dw = dw.view(wshape)
        elif db is not None:
            colsum_acc(dtok, db)
        return None, dw, db, dcls.view(1, 1, D), dpos.view(1, n + 1, D), None, None


def patch_embed(img, conv_w, conv_b, cls, pos, dropout_p, dtype):
    return PatchEmbedFn.apply(img, conv_w, conv_b, cls, pos, float(dropout_p), dtype)
