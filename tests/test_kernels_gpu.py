"""Each HIP kernel, called through the C-ABI (libvitk.so via ctypes), against a plain PyTorch
fp32 computation of the same operation on the same (dtype-rounded) inputs.

Tolerances (|got - ref|_max relative to |ref|_max):
  fp32 kernels 1e-5 (summation-order noise only); bf16 kernels 2e-2 (north-star bound; observed
  values are printed) unless the output is fp32 accumulated from exact bf16 products (2e-4).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: 1e-5, torch.bfloat16: 2e-2}


@pytest.fixture(scope="module")
def ops():
    import vitk
    return vitk.ops


def rnd(shape, seed, dtype, scale=1.0):
    a = np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * scale
    return torch.from_numpy(a).to(DEV).to(dtype)


def relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def check(got, ref, tol, what):
    e = relerr(got, ref)
    print(f"{what}: rel-to-max err {e:.3e} (tol {tol:g})")
    assert math.isfinite(e) and e <= tol, f"{what}: {e:.3e} > {tol:g}"


GEMM_SHAPES = [(128, 128, 64), (394, 768, 768), (1000, 2304, 768), (300, 200, 128), (77, 3072, 192), (130, 136, 3072),
               (21900, 768, 256), (40000, 1024, 192)]   # the last two span several rounds of 256x256 tiles over 256 CUs


@pytest.fixture(params=[-1, 0, 1, 2, 3, 4, 5], ids=["auto", "128x128", "256x128", "256x256", "256x128k32", "256x256pp", "pp+tail"])
def tile_shape(request):
    """Force each workgroup tile shape of the bf16 GEMMs in turn (-1 = size-based choice)."""
    import vitk
    vitk.lib.vitk_debug_force_gemm_shape(request.param)
    yield request.param
    vitk.lib.vitk_debug_force_gemm_shape(-1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_epilogues(ops, dtype, M, N, K, tile_shape):
    import vitk
    if dtype == torch.float32 and tile_shape > 0:
        pytest.skip("tile shapes apply to the bf16 kernels")
    a = rnd((M, K), 1, dtype)
    w = rnd((N, K), 2, dtype, 1 / math.sqrt(K))
    bias = rnd((N,), 3, torch.float32)
    res = rnd((M, N), 4, dtype)
    af, wf = a.float(), w.float()
    base = af @ wf.t()
    tol = TOL[dtype]
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    check(ops.gemm_nt(a, w, out, vitk.EPI_NONE), base, tol, "none")
    check(ops.gemm_nt(a, w, out, vitk.EPI_BIAS, bias=bias), base + bias, tol, "bias")
    check(ops.gemm_nt(a, w, out, vitk.EPI_BIAS_RES, bias=bias, residual=res), base + bias + res.float(), tol, "bias_res")
    aux = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm_nt(a, w, out, vitk.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = (base + bias).requires_grad_(True)
    act = torch.nn.functional.gelu(pre)
    act.backward(torch.ones_like(act))
    # what the backward epilogue multiplies with, less 1/2; beside bf16 activations the tensor holds IEEE halves
    half = dtype == torch.bfloat16
    check(aux.view(torch.float16) if half else aux, pre.grad - 0.5, 1e-3 if half else tol, "gelu.derivative")
    check(out, act, tol, "gelu.out")
    u = rnd((M, N), 5, torch.float16 if half else dtype, 0.3)
    check(ops.gemm_nt(a, w, out, vitk.EPI_DGELU, aux=u.view(dtype)), base * (u.float() + 0.5), tol, "dgelu")


@pytest.mark.parametrize("M,N,K", [(394, 768, 768), (300, 200, 128)])
def test_gemm_nt_bf16_operands_fp32_stream_output(ops, M, N, K, tile_shape):
    """bf16 operands, fp32 C and fp32 residual: the epilogues that feed the residual stream."""
    import vitk
    a = rnd((M, K), 1, torch.bfloat16)
    w = rnd((N, K), 2, torch.bfloat16, 1 / math.sqrt(K))
    bias = rnd((N,), 3, torch.float32)
    res = rnd((M, N), 4, torch.float32, 3.0)
    base = a.float() @ w.float().t()
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    check(ops.gemm_nt(a, w, out, vitk.EPI_BIAS, bias=bias), base + bias, 2e-5, "f32out.bias")
    check(ops.gemm_nt(a, w, out, vitk.EPI_BIAS_RES, bias=bias, residual=res), base + bias + res, 2e-5, "f32out.bias_res")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt_identity_asymmetric(ops, dtype, tile_shape):
    """A = I against an asymmetric B catches transposed or permuted fragment maps exactly."""
    import vitk
    if dtype == torch.float32 and tile_shape > 0:
        pytest.skip("tile shapes apply to the bf16 kernels")
    n = 512
    a = torch.eye(n, device=DEV).to(dtype)
    w = (torch.arange(n * n, device=DEV).reshape(n, n) % 251).float().to(dtype)   # exactly representable
    out = torch.empty((n, n), dtype=dtype, device=DEV)
    ops.gemm_nt(a, w, out, vitk.EPI_NONE)
    assert torch.equal(out.float(), w.float().t())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("T,M,N,split", [(64, 128, 128, 1), (1576, 768, 768, 0), (1000, 200, 136, 3),
                                         (33, 8, 8, 0), (4096, 3072, 768, 0), (2048, 200, 136, 0),
                                         (1024, 520, 264, 2), (160, 256, 256, 1)])
def test_gemm_tn_acc(ops, dtype, T, M, N, split, tile_shape):
    if dtype == torch.float32 and tile_shape > 0:
        pytest.skip("tile shapes apply to the bf16 kernels")
    a = rnd((T, M), 1, dtype)
    b = rnd((T, N), 2, dtype)
    c0 = rnd((M, N), 3, torch.float32)
    c = c0.clone()
    ops.gemm_tn_acc(a, b, c, split)
    ref = c0.double() + a.double().t() @ b.double()
    check(c, ref, 2e-4 if dtype == torch.bfloat16 else 1e-5, "tn")
    # the same product with the bias gradient (column sums of a) riding along
    c = c0.clone()
    s0 = rnd((M,), 4, torch.float32)
    cs = s0.clone()
    ops.gemm_tn_acc(a, b, c, split, colsum=cs)
    check(c, ref, 2e-4 if dtype == torch.bfloat16 else 1e-5, "tn with colsum")
    check(cs, s0.double() + a.double().sum(0), 1e-5, "colsum")


@pytest.mark.parametrize("workspace", [False, True], ids=["atomics", "two-pass"])
@pytest.mark.parametrize("kernel", [1, 0], ids=["four-wave", "eight-wave"])
@pytest.mark.parametrize("T,M,N,split", [(4096, 768, 768, 0), (4096, 3072, 768, 7), (4128, 520, 1000, 3), (2048, 1000, 520, 0),
                                         (5024, 264, 256, 4), (3200, 768, 512, 2), (4096, 256, 1280, 0), (352, 512, 768, 2)])
def test_gemm_tn_256_kernels(ops, T, M, N, split, kernel, workspace):
    """Both weight-gradient kernels of the 256x256 path (one wave per SIMD / eight-wave ping-pong), forced: ragged
    tile edges, one to five tile columns (the owners of the column sums change with it), odd slab counts per
    slice, write-out by atomics and through the registered workspace; the products of the two kernels are the
    same bits when the workspace orders the sum."""
    import ctypes
    import vitk
    lib = vitk.lib
    lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
    lib.vitk_debug_tn_w4.restype = None
    a = rnd((T, M), 1, torch.bfloat16)
    b = rnd((T, N), 2, torch.bfloat16)
    c0 = rnd((M, N), 3, torch.float32)
    s0 = rnd((M,), 4, torch.float32)
    ref = c0.double() + a.double().t() @ b.double()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV) if workspace else None
    own = ops._tn_workspace(a.device)               # the operator module's registration, put back below
    try:
        lib.vitk_debug_force_gemm_shape(4)
        assert lib.vitk_gemm_tn_set_workspace(ctypes.c_void_p(ws.data_ptr() if workspace else None),
                                              ctypes.c_size_t(ws.numel() if workspace else 0)) == 0
        outs = {}
        for k in (kernel, 1 - kernel):
            lib.vitk_debug_tn_w4(k)
            c = c0.clone()
            ops.gemm_tn_acc(a, b, c, split)
            c2, cs = c0.clone(), s0.clone()
            ops.gemm_tn_acc(a, b, c2, split, colsum=cs)
            outs[k] = (c, c2, cs)
        c, c2, cs = outs[kernel]
        check(c, ref, 2e-4, "tn")
        check(c2, ref, 2e-4, "tn with colsum")
        check(cs, s0.double() + a.double().sum(0), 1e-5, "colsum")
        if workspace and split != 1:
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    finally:
        lib.vitk_debug_tn_w4(1)
        lib.vitk_debug_force_gemm_shape(-1)
        assert lib.vitk_gemm_tn_set_workspace(ctypes.c_void_p(own.data_ptr()), ctypes.c_size_t(own.numel())) == 0


@pytest.mark.parametrize("kernel", [1, 0], ids=["four-wave", "eight-wave"])
def test_gemm_tn_256_kernels_on_column_slices(ops, kernel):
    """Operands that are column blocks of wider matrices (row strides larger than the widths, bases off the
    row start), as the fused QKV gradient and the window-attention layers hand them over."""
    import ctypes
    import vitk
    lib = vitk.lib
    lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
    lib.vitk_debug_tn_w4.restype = None
    T, M, N = 2080, 776, 520
    a_full = rnd((T, M + 96), 1, torch.bfloat16)
    b_full = rnd((T, 3 * N), 2, torch.bfloat16)
    a, b = a_full[:, 32:32 + M], b_full[:, N:2 * N]
    c0 = rnd((M, N), 3, torch.float32)
    s0 = rnd((M,), 4, torch.float32)
    try:
        lib.vitk_debug_force_gemm_shape(4)
        lib.vitk_debug_tn_w4(kernel)
        c, cs = c0.clone(), s0.clone()
        ops.gemm_tn_acc(a, b, c, 0, colsum=cs)
        check(c, c0.double() + a.double().t() @ b.double(), 2e-4, "tn on slices")
        check(cs, s0.double() + a.double().sum(0), 1e-5, "colsum on slices")
    finally:
        lib.vitk_debug_tn_w4(1)
        lib.vitk_debug_force_gemm_shape(-1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_exact_small_integers(ops, dtype, tile_shape):
    if dtype == torch.float32 and tile_shape > 0:
        pytest.skip("tile shapes apply to the bf16 kernels")
    T, M, N = 200, 384, 264
    a = torch.randint(-3, 4, (T, M), device=DEV, generator=torch.Generator(DEV).manual_seed(1)).to(dtype)
    b = torch.randint(-3, 4, (T, N), device=DEV, generator=torch.Generator(DEV).manual_seed(2)).to(dtype)
    c = torch.zeros((M, N), device=DEV)
    cs = torch.zeros(M, device=DEV)
    ops.gemm_tn_acc(a, b, c, 1, colsum=cs)
    assert torch.equal(c, a.float().t() @ b.float())
    assert torch.equal(cs, a.float().sum(0))


@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(5, 128), (1576, 768), (333, 1024), (10, 64), (64, 2048)])
def test_layernorm(ops, dtype, xdtype, rows, cols):
    """x (residual stream, fp32 in the model) -> y in `dtype`; dy arrives in `dtype`."""
    if xdtype == torch.bfloat16 and dtype == torch.float32:
        pytest.skip("bf16 stream with fp32 activations is not a supported combination")
    x = rnd((rows, cols), 1, xdtype, 2.0) + 0.5
    g = (1 + 0.1 * rnd((cols,), 2, torch.float32)).requires_grad_(True)
    b = (0.1 * rnd((cols,), 3, torch.float32)).requires_grad_(True)
    dy = rnd((rows, cols), 4, dtype)
    dres = rnd((rows, cols), 5, xdtype)
    xg = x.clone().requires_grad_(True)
    h, y = ops.add_norm(xg, g, b, 1e-6, dtype)
    (y.float() * dy.float()).sum().backward(retain_graph=True)
    xr = x.float().requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-6)
    (yr * dy.float()).sum().backward()
    tol = TOL[dtype]
    check(y, yr, tol, "ln.y")
    check(xg.grad, xr.grad, tol, "ln.dx")
    check(g.grad, gr.grad, 2e-3 if dtype == torch.bfloat16 else 1e-5, "ln.dgamma")
    check(b.grad, br.grad, 2e-3 if dtype == torch.bfloat16 else 1e-5, "ln.dbeta")
    # fused residual-gradient add
    xg.grad = None
    ((y.float() * dy.float()).sum() + (h.float() * dres.float()).sum()).backward()
    check(xg.grad, xr.grad + dres.float(), tol, "ln.dx+dres")


def _attn_ref(qkv, B, N, H):
    T, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    q, k, v = [t.reshape(B, N, H, dh).permute(0, 2, 1, 3) for t in qkv.split(D, dim=1)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(T, D)
    return o, p, torch.logsumexp(s, dim=-1)


@pytest.fixture(params=[0, 1, 2, 3], ids=["persistent", "streaming", "resident-fwd", "fused-bwd"])
def attn_variant(request):
    """bf16 attention: 0 = the persistent feeder-wave kernels (N <= 224, the default), 1 = the
    streaming kernels, 2 = the one-shot LDS-resident forward with the streaming backward, 3 = that
    forward with the one-shot fused backward."""
    import vitk
    vitk.lib.vitk_debug_force_attn_streaming(request.param)
    yield request.param
    vitk.lib.vitk_debug_force_attn_streaming(0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (1, 2, 5), (3, 1, 64), (1, 2, 33), (1, 1, 577), (2, 2, 10),
                                   (1, 2, 224), (2, 1, 129)])
def test_attention_fwd_bwd(ops, dtype, B, H, N, attn_variant):
    if dtype == torch.float32 and attn_variant:
        pytest.skip("variants apply to the bf16 kernels")
    D = 64 * H
    qkv = rnd((B * N, 3 * D), 7, dtype, 1.0)
    dout = rnd((B * N, D), 8, dtype)
    x = qkv.clone().requires_grad_(True)
    out, lse = ops.attention(x, B, N, H)
    out.backward(dout)
    xr = qkv.float().requires_grad_(True)
    o_ref, p_ref, lse_ref = _attn_ref(xr, B, N, H)
    o_ref.backward(dout.float())
    tol = TOL[dtype]
    check(out, o_ref, tol, "attn.out")
    check(lse, lse_ref, 1e-2 if dtype == torch.bfloat16 else 1e-5, "attn.lse")
    D_ = D
    for i, nm in enumerate("qkv"):
        check(x.grad[:, i * D_:(i + 1) * D_], xr.grad[:, i * D_:(i + 1) * D_], tol * 1.5, "attn.d" + nm)
    probs = ops.attention_probs(qkv, lse, B, N, H)
    check(probs, p_ref, 2e-2 if dtype == torch.bfloat16 else 1e-5, "attn.probs")


@pytest.mark.parametrize("B,H,N", [(48, 12, 197), (90, 3, 64), (60, 5, 130), (70, 4, 33), (23, 12, 224)])
def test_attention_bwd_more_pairs_than_cus(ops, B, H, N):
    """More (image, head) pairs than compute units: the persistent backward kernel walks several
    pairs per workgroup (uneven counts), with its look-ahead running across pair boundaries."""
    D = 64 * H
    qkv = rnd((B * N, 3 * D), 17, torch.bfloat16, 1.0)
    dout = rnd((B * N, D), 18, torch.bfloat16)
    x = qkv.clone().requires_grad_(True)
    out, _ = ops.attention(x, B, N, H)
    out.backward(dout)
    xr = qkv.float().requires_grad_(True)
    o_ref, _, _ = _attn_ref(xr, B, N, H)
    o_ref.backward(dout.float())
    check(out, o_ref, TOL[torch.bfloat16], "attn.out")
    for i, nm in enumerate("qkv"):
        # per (image, head) block, so that one wrong pair cannot hide behind the global maximum
        got = x.grad[:, i * D:(i + 1) * D].float().reshape(B, N, H, 64)
        ref = xr.grad[:, i * D:(i + 1) * D].reshape(B, N, H, 64)
        err = (got - ref).abs().amax(dim=(1, 3))
        scale = ref.abs().amax(dim=(1, 3)).clamp_min(1e-6)
        worst = (err / scale).max().item()
        assert worst < 1.5 * TOL[torch.bfloat16], f"attn.d{nm}: worst pair {worst:.3e}"


@pytest.mark.parametrize("gemm_wgs,attn_over", [(1024, 4), (0, 1), (64, 2)])
def test_launch_shapes_of_the_multi_rank_path(ops, gemm_wgs, attn_over):
    """DataParallel with more than one rank starts the persistent kernels with more, smaller workgroups
    (vitk_gemm_set_persistent(1024), vitk_attn_set_oversubscription(4)): a one-GPU box never takes that
    path by itself, so it is forced here (plus one workgroup per tile, and fewer workgroups than CUs)."""
    import vitk
    vitk.check(vitk.lib.vitk_gemm_set_persistent(gemm_wgs), "vitk_gemm_set_persistent")
    vitk.check(vitk.lib.vitk_attn_set_oversubscription(attn_over), "vitk_attn_set_oversubscription")
    try:
        for (M, N, K) in [(21900, 768, 256), (40000, 1024, 192), (1000, 2304, 768)]:
            a = rnd((M, K), 1, torch.bfloat16)
            w = rnd((N, K), 2, torch.bfloat16, 1 / math.sqrt(K))
            bias = rnd((N,), 3, torch.float32)
            res = rnd((M, N), 4, torch.float32, 3.0)
            base = a.float() @ w.float().t()
            out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            check(ops.gemm_nt(a, w, out, vitk.EPI_BIAS, bias=bias), base + bias, TOL[torch.bfloat16], "bias")
            aux = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            ops.gemm_nt(a, w, out, vitk.EPI_BIAS_GELU, bias=bias, aux=aux)
            check(out, torch.nn.functional.gelu(base + bias), TOL[torch.bfloat16], "gelu.out")
            out32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
            check(ops.gemm_nt(a, w, out32, vitk.EPI_BIAS_RES, bias=bias, residual=res), base + bias + res, 2e-5, "bias_res")
        for (B, H, N) in [(48, 12, 197), (23, 12, 224)]:
            D = 64 * H
            qkv = rnd((B * N, 3 * D), 17, torch.bfloat16, 1.0)
            dout = rnd((B * N, D), 18, torch.bfloat16)
            x = qkv.clone().requires_grad_(True)
            out, _ = ops.attention(x, B, N, H)
            out.backward(dout)
            xr = qkv.float().requires_grad_(True)
            o_ref, _, _ = _attn_ref(xr, B, N, H)
            o_ref.backward(dout.float())
            check(out, o_ref, TOL[torch.bfloat16], "attn.out")
            check(x.grad, xr.grad, 1.5 * TOL[torch.bfloat16], "attn.dqkv")
    finally:
        vitk.check(vitk.lib.vitk_gemm_set_persistent(256), "vitk_gemm_set_persistent")
        vitk.check(vitk.lib.vitk_attn_set_oversubscription(1), "vitk_attn_set_oversubscription")


@pytest.mark.parametrize("wgs", [64, 256])
@pytest.mark.parametrize("M,N,K", [(40000, 1024, 192), (40000, 3072, 192), (9000, 320, 256), (9000, 576, 192), (6100, 200, 192)])
def test_persistent_gemm_after_a_partial_tile(ops, M, N, K, wgs):
    """A persistent workgroup that walks on from a tile sticking out over the edge of C: in that tile some
    waves issue no stores at all, so the next tile's first wait must not count on them.  The bare epilogue
    has no other memory operation that would drain the queue.  The one-tile-per-workgroup kernel runs the
    same arithmetic: the results must be the same bits."""
    import vitk
    a = rnd((M, K), 1, torch.bfloat16)
    w = rnd((N, K), 2, torch.bfloat16, 1 / math.sqrt(K))
    base = a.float() @ w.float().t()
    vitk.lib.vitk_debug_force_gemm_shape(4)
    try:
        vitk.check(vitk.lib.vitk_gemm_set_persistent(0), "vitk_gemm_set_persistent")
        ref = ops.gemm_nt(a, w, torch.empty((M, N), dtype=torch.bfloat16, device=DEV), vitk.EPI_NONE)
        check(ref, base, TOL[torch.bfloat16], "one tile per workgroup")
        vitk.check(vitk.lib.vitk_gemm_set_persistent(wgs), "vitk_gemm_set_persistent")
        for rep in range(3):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm_nt(a, w, out, vitk.EPI_NONE)
            assert torch.equal(out, ref), f"persistent({wgs}) differs from the one-tile kernel (run {rep})"
    finally:
        vitk.lib.vitk_debug_force_gemm_shape(-1)
        vitk.check(vitk.lib.vitk_gemm_set_persistent(256), "vitk_gemm_set_persistent")


def test_attention_online_softmax_rescale(ops, attn_variant):
    """Force the running maximum to jump at a late key tile (rescale branch of the online softmax)."""
    B, H, N = 1, 1, 197
    qkv = rnd((N, 192), 11, torch.bfloat16, 0.3)
    qkv[:, 64:128][150] = qkv[:, 0:64][7] * 40          # key 150 aligned with query 7: score spike in tile 4
    qkv[:, 64:128][3] = qkv[:, 0:64][100] * 25
    out, lse = ops.attention(qkv, B, N, H)
    o_ref, _, lse_ref = _attn_ref(qkv.float(), B, N, H)
    check(out, o_ref, 2e-2, "attn.spike.out")
    check(lse, lse_ref, 1e-2, "attn.spike.lse")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,img,P,D", [(3, 32, 16, 128), (2, 48, 16, 128), (2, 224, 16, 768)])
def test_patch_embed(ops, dtype, B, img, P, D):
    x = rnd((B, 3, img, img), 1, torch.float32)
    w = rnd((D, 3, P, P), 2, torch.float32, 0.05).requires_grad_(True)
    b = rnd((D,), 3, torch.float32, 0.1).requires_grad_(True)
    n = (img // P) ** 2
    cls = rnd((1, 1, D), 4, torch.float32, 0.1).requires_grad_(True)
    pos = rnd((1, n + 1, D), 5, torch.float32, 0.1).requires_grad_(True)
    dE = rnd((B, n + 1, D), 6, dtype)
    out = ops.patch_embed(x, w, b, cls, pos, 0.0, dtype)
    out.backward(dE)
    xr = x.to(dtype).float()
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br, cr, pr = [t.detach().clone().requires_grad_(True) for t in (b, cls, pos)]
    tok = torch.nn.functional.conv2d(xr, wr, br, stride=P).flatten(2).transpose(-1, -2)
    ref = torch.cat((cr.expand(B, -1, -1), tok), dim=1) + pr
    ref.backward(dE.float())
    tol = TOL[dtype]
    check(out, ref, tol, "embed.out")
    check(w.grad, wr.grad, 2e-3 if dtype == torch.bfloat16 else 1e-5, "embed.dW")
    check(b.grad, br.grad, 2e-3 if dtype == torch.bfloat16 else 1e-5, "embed.db")
    check(cls.grad, cr.grad, 2e-3 if dtype == torch.bfloat16 else 1e-5, "embed.dcls")
    check(pos.grad, pr.grad, 2e-3 if dtype == torch.bfloat16 else 1e-5, "embed.dpos")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cast_colsum_convert(ops, dtype):
    w = rnd((200, 136), 1, torch.float32)
    c, ct = ops.cast_weight(w, dtype, True, True)
    assert torch.equal(c, w.to(dtype)) and torch.equal(ct, w.to(dtype).t().contiguous())
    x = rnd((1576, 264), 2, dtype)
    out = torch.ones(264, device=DEV)
    ops.colsum_acc(x, out)
    check(out, 1 + x.double().sum(0), 1e-5, "colsum")
    y = ops.convert(x, torch.float32)
    assert torch.equal(y, x.float())
    assert torch.equal(ops.convert(y, dtype), x)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_linear_dropout_mask_consistency(ops, dtype):
    """Dropout in a GEMM epilogue: keep-rate, 1/keep scaling, and the SAME mask in backward."""
    T, D, F_ = 512, 128, 256
    p = 0.25
    keep = 1 - round(p * 65536) / 65536
    w2 = rnd((D, F_), 4, torch.float32, 1 / math.sqrt(F_)).requires_grad_(True)
    b2 = rnd((D,), 5, torch.float32, 0.1).requires_grad_(True)
    dy = rnd((T, D), 6, dtype)
    x2 = rnd((T, F_), 7, dtype).requires_grad_(True)
    torch.manual_seed(6)
    z = ops.linear(x2, (w2,), (b2,), None, p)
    z.backward(dy)
    mz = (z.float() != 0).float()
    assert abs(float(mz.mean()) - keep) < 0.01
    x2r = x2.detach().float().requires_grad_(True)
    w2r = w2.detach().to(dtype).float().requires_grad_(True)
    b2r = b2.detach().clone().requires_grad_(True)
    zr = (x2r @ w2r.t() + b2r) * mz / keep
    zr.backward(dy.float())
    check(z, zr, TOL[dtype], "drop.fwd")
    check(x2.grad, x2r.grad, TOL[dtype] * 1.5, "drop.dx")
    check(w2.grad, w2r.grad, 3e-3 if dtype == torch.bfloat16 else 1e-5, "drop.dW")
    check(b2.grad, b2r.grad, 3e-3 if dtype == torch.bfloat16 else 1e-5, "drop.db")


def test_mlp_hidden_dropout_statistics(ops):
    """fc1 dropout: E[y] is preserved and gradient flows only through kept hidden units."""
    T, D, F_ = 256, 128, 256
    x = rnd((T, D), 1, torch.float32).requires_grad_(True)
    w1 = rnd((F_, D), 2, torch.float32, 1 / math.sqrt(D)).requires_grad_(True)
    b1 = torch.zeros(F_, device=DEV).requires_grad_(True)
    w2 = torch.eye(F_, device=DEV)[:D].contiguous().requires_grad_(True)       # y = first D hidden units
    b2 = torch.zeros(D, device=DEV).requires_grad_(True)
    # p only on the hidden layer is not separable through the public op; use p on both and w2 = I:
    # y[t, j] = m2 * m1 * gelu(u)[t, j] / keep^2
    p = 0.5
    y = ops.mlp(x, w1, b1, w2, b2, None, p)
    g = torch.nn.functional.gelu(x.detach() @ w1.detach().t())[:, :D]
    kept = (y != 0)
    check(y, g * 4.0 * kept.float(), 1e-5, "hidden-dropout.y")
    frac = float(kept.float().mean())
    assert abs(frac - 0.25) < 0.02, frac
    y.backward(torch.ones_like(y))
    # gradient wrt b1[j] = sum_t m1*m2/keep^2 * gelu'(u): zero exactly where every row dropped -> compare sums
    u = (x.detach() @ w1.detach().t())
    ur = u.clone().requires_grad_(True)
    (torch.nn.functional.gelu(ur)[:, :D] * kept.float() * 4.0).sum().backward()
    check(b1.grad[:D], ur.grad.sum(0)[:D], 1e-4, "hidden-dropout.db1")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_weight_operands_batched_cast(ops, dtype):
    """vitk_cast_weights_batched through ops.weight_operands: [N, K] and [K, N] copies of single weights, of three
    weights concatenated along N (query | key | value), a 4-D convolution weight, and shapes that take the
    one-element-per-lane path (sides not multiples of 4, ragged tiles)."""
    g = torch.Generator().manual_seed(3)
    for shapes in ([(768, 768)], [(256, 192)] * 3, [(64, 3, 16, 16)], [(10, 50)], [(130, 68)], [(66, 64), (62, 64)], [(3072, 768)]):
        ws = [torch.randn(s, generator=g).to(DEV) for s in shapes]
        w, wt = ops.weight_operands(ws, dtype)
        ref = torch.cat([x.reshape(x.shape[0], -1) for x in ws], 0).to(dtype)
        assert torch.equal(w, ref), shapes
        assert torch.equal(wt, ref.t().contiguous()), shapes
        w2, none = ops.weight_operands(ws, dtype, want=True, want_t=False)
        assert none is None and torch.equal(w2, ref)


def test_matmul_f32_on_the_library_gemm(ops):
    """ops.matmul_f32 (the weight-sized fp32 products of the task-conditioned variants on the exact-fp32 MFMA GEMM):
    values and both gradients against torch in fp64, aligned sides and sides that are zero-extended (rank 2, 89
    columns, 50 rows)."""
    g = torch.Generator().manual_seed(9)
    for (M, K, N) in [(64, 2, 128), (128, 2, 64), (768, 8, 768), (50, 3, 89), (256, 130, 40)]:
        a = torch.randn(M, K, generator=g).to(DEV).requires_grad_(True)
        b = torch.randn(K, N, generator=g).to(DEV).requires_grad_(True)
        dc = torch.randn(M, N, generator=g).to(DEV)
        c = ops.matmul_f32(a, b)
        c.backward(dc)
        ad, bd = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        cd = ad @ bd
        cd.backward(dc.double())
        check(c, cd, 1e-5, f"matmul_f32 {M}x{K}x{N}")
        check(a.grad, ad.grad, 1e-5, "matmul_f32.dA")
        check(b.grad, bd.grad, 1e-5, "matmul_f32.dB")


def test_weight_gradient_workspace_belongs_to_one_stream(ops):
    """The two-pass write-out of the 256x256 weight-gradient kernels goes through ONE registered workspace per
    device: the first stream that uses it after registration owns it, a call on another stream of the same device
    takes the atomic write-out (two calls in flight would otherwise overwrite each other's partial tiles).  Both
    give the product; run concurrently on two streams they still do."""
    import vitk
    T, M, N = 8192, 768, 768
    a = rnd((T, M), 1, torch.bfloat16)
    b = rnd((T, N), 2, torch.bfloat16)
    ref = a.double().t() @ b.double()
    vitk.lib.vitk_debug_force_gemm_shape(4)
    try:
        ops._TN_WORKSPACE.clear()
        c0 = torch.zeros((M, N), dtype=torch.float32, device=DEV)
        ops.gemm_tn_acc(a, b, c0)                          # registers the workspace; this (default) stream claims it
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        c1 = torch.zeros((M, N), dtype=torch.float32, device=DEV)
        c2 = torch.zeros((M, N), dtype=torch.float32, device=DEV)
        with torch.cuda.stream(side):
            ops.gemm_tn_acc(a, b, c1)                      # another stream: atomics
        ops.gemm_tn_acc(a, b, c2)                          # meanwhile, on the owner
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for c in (c0, c1, c2):
            check(c, ref, 2e-4, "tn on its own / another stream")
    finally:
        vitk.lib.vitk_debug_force_gemm_shape(-1)
