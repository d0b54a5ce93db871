"""vitk_layernorm_bwd_static: the LayerNorm backward that allocates nothing (its dgamma/dbeta partials go
to an array built into the library) against vitk_layernorm_bwd with a caller-supplied workspace, through
the C-ABI, on two calls in a row, the second on a smaller input (the array keeps the first call's
contents).  dx is bit-exact; dgamma/dbeta are sums of float atomics whose order differs from launch to
launch, so they are held to the fp32 bar of 1e-5 of their scale."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _run(lib, vitk, static, dy, x, gamma, mean, rstd, dres, stream):
    rows, cols = x.shape
    dx = torch.full_like(x, float("nan"))
    dg = torch.zeros(cols, dtype=torch.float32, device=DEV)
    db = torch.zeros(cols, dtype=torch.float32, device=DEV)
    dt, dyt = vitk.dtype_code(x.dtype), vitk.dtype_code(dy.dtype)
    if static:
        rc = lib.vitk_layernorm_bwd_static(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dg),
                                           _p(db), rows, cols, dt, dyt, stream)
    else:
        ws = torch.empty(lib.vitk_layernorm_bwd_workspace(rows, cols), dtype=torch.uint8, device=DEV)
        rc = lib.vitk_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dg), _p(db),
                                    _p(ws), rows, cols, dt, dyt, stream)
    assert rc == 0, vitk.lib.vitk_error_string(rc)
    torch.cuda.synchronize()
    return dx, dg, db


def _case(rows, cols, xdt, dydt, seed, with_res):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(rows, cols, generator=g).to(DEV).to(xdt)
    dy = torch.randn(rows, cols, generator=g).to(DEV).to(dydt)
    gamma = (1.0 + 0.1 * torch.randn(cols, generator=g)).to(DEV)
    xf = x.float()
    mean = xf.mean(1)
    rstd = torch.rsqrt(xf.var(1, unbiased=False) + 1e-6)
    dres = torch.randn(rows, cols, generator=g).to(DEV).to(xdt) if with_res else None
    return dy, x, gamma, mean.contiguous(), rstd.contiguous(), dres


@pytest.mark.parametrize("xdt,dydt", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                      (torch.bfloat16, torch.bfloat16)])
def test_static_matches_workspace_variant_twice(xdt, dydt):
    import vitk
    lib = vitk.lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # first a large call (every one of the 1024 partial rows is used), then a smaller one with fewer
    # blocks and fewer columns that must not see what the first left behind
    for rows, cols, seed, with_res in ((8 * 197 * 3, 768, 1, True), (37, 256, 2, False), (5000, 1024, 3, True)):
        args = _case(rows, cols, xdt, dydt, seed, with_res)
        dx0, dg0, db0 = _run(lib, vitk, False, *args, stream)
        dx1, dg1, db1 = _run(lib, vitk, True, *args, stream)
        assert torch.equal(dx0, dx1), (rows, cols)
        for a, b, what in ((dg0, dg1, "dgamma"), (db0, db1, "dbeta")):
            scale = float(a.abs().max())
            err = float((a - b).abs().max())
            assert err <= 1e-5 * scale + 1e-12, (what, rows, cols, err, scale)
        # and both agree with the formula (fp64 on the host), so "equal" is not "equally wrong"
        dy, x, gamma, mean, rstd, dres = [None if t is None else t.double().cpu() for t in args]
        xh = (x - mean[:, None]) * rstd[:, None]
        gy = dy * gamma
        ref = rstd[:, None] * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True))
        if dres is not None:
            ref = ref + dres
        tol = 1e-5 if xdt == torch.float32 else 2e-2
        assert float((dx1.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())
        np.testing.assert_allclose(dg1.double().cpu().numpy(), (dy * xh).sum(0).numpy(), rtol=0, atol=1e-5 * float((dy * xh).sum(0).abs().max()))
        np.testing.assert_allclose(db1.double().cpu().numpy(), dy.sum(0).numpy(), rtol=0, atol=1e-5 * float(dy.sum(0).abs().max()))


def test_static_rejects_oversize_and_bad_args():
    import vitk
    lib = vitk.lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dy, x, gamma, mean, rstd, _ = _case(8, 1028, torch.float32, torch.float32, 4, False)
    dx = torch.zeros_like(x)
    dg = torch.zeros(1028, device=DEV)
    rc = lib.vitk_layernorm_bwd_static(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), None, _p(dx), _p(dg), _p(dg),
                                       8, 1028, vitk.F32, vitk.F32, stream)
    assert rc == -5 and b"scratch" in lib.vitk_error_string(rc)
    rc = lib.vitk_layernorm_bwd_static(None, _p(x), _p(gamma), _p(mean), _p(rstd), None, _p(dx), _p(dg), _p(dg),
                                       8, 1024, vitk.F32, vitk.F32, stream)
    assert rc == -3
    torch.cuda.synchronize()
    assert float(dx.abs().max()) == 0.0          # nothing was launched


def test_static_allocates_nothing_and_runs_under_graph_capture():
    """What the entry point exists for: the device's free memory is the same before and after a warmed-up call
    (hipMemGetInfo), and the call can be recorded into a HIP graph, where any allocation would fail the capture;
    the replayed graph writes the same bits as the direct call."""
    import vitk
    lib = vitk.lib
    args = _case(4 * 197, 768, torch.float32, torch.bfloat16, 9, True)
    dy, x, gamma, mean, rstd, dres = args
    rows, cols = x.shape
    dt, dyt = vitk.dtype_code(x.dtype), vitk.dtype_code(dy.dtype)
    dx = torch.empty_like(x)
    dg = torch.zeros(cols, dtype=torch.float32, device=DEV)
    db = torch.zeros(cols, dtype=torch.float32, device=DEV)

    def call(stream):
        rc = lib.vitk_layernorm_bwd_static(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dg),
                                           _p(db), rows, cols, dt, dyt, stream)
        assert rc == 0, vitk.lib.vitk_error_string(rc)

    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call(cur)                                        # first call on this device takes the symbol's address
    torch.cuda.synchronize()
    want = dx.clone()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        call(cur)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    graph = torch.cuda.CUDAGraph()
    dx.fill_(float("nan"))
    with torch.cuda.graph(graph):
        call(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    dg.zero_()
    db.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(dx, want)
