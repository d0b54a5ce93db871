"""RKR variant (task-conditioned linears folded into the GEMM weight operand) on the HIP kernels against
the golden vectors of reference models/ViT/modeling_RKR.py: 1e-5 fp32, 2e-2 bf16 (max-norm per tensor).

The gradients of the rank-K factors (LM_list / RM_list) are sums over a whole weight-gradient row or
column with cancelling signs, which amplifies bf16 rounding of the operands: on the ViT-B/16 fixture
`torch.autocast(bfloat16)` applied to the reference computation itself is 3.7e-2 (LM), 2.6e-2 (RM) away
from fp32 (`python tools/bf16_calibration.py rkr`).  Those two families are held to 5e-2 in bf16; everything
else to 2e-2.  Observed: 2.0e-2 on ViT-B/16; on the 60-token tiny fixtures the worst entry is always the KEY
projection's factor in layer 0 (the key gradients of an image sum to zero over its tokens, so what is left of them
after bf16 rounding of dK is mostly rounding noise), and it moves with any change of rounding upstream: three
variants of the MLP epilogue gave 3.1e-2, 4.1e-2 and 4.4e-2 here while the error of the MLP backward itself
went DOWN (tools/gelu_aux_check.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from test_model_gpu import DEV, TOL, Report, check

pytestmark = pytest.mark.gpu
FACTOR_TOL = {torch.float32: 1e-5, torch.bfloat16: 5e-2}


def _is_factor(name):
    return "LM_list" in name or "RM_list" in name


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


def build_rkr(geom, img, ncls, seed, dtype, vis=False):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling_RKR import VisionTransformer
    vitk.set_compute_dtype(dtype)
    cfg = cfgmod.get_tiny_config() if geom == "tiny" else cfgmod.get_config("ViT-B_16_RKR")
    cfg.transformer.dropout_rate = 0.0
    cfg.RG, cfg.SFG, cfg.K, cfg.rkr_scale, cfg.multi_head, cfg.task_num = True, True, 2, 1e-1, True, len(ncls)
    model = VisionTransformer(cfg, img, num_classes=ncls, zero_head=False, vis=vis)
    model.load_state_dict(synth.synth_rkr_params(geom, img, ncls, seed=seed), strict=True)
    return model.to(DEV).train()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rkr_tiny_against_reference_golden(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, "rkr_tiny.npz"))
    ncls = [10, 7, 5]
    model = build_rkr("tiny", 32, ncls, 17, dtype, vis=True)
    # 60 tokens in the whole batch: weight gradients are sums of 60 bf16-rounded outer products, the
    # same small-sample situation as the 20-token fixture of test_model_gpu (3e-2 there as well)
    tol = TOL[dtype] if dtype == torch.float32 else 3e-2
    ftol = FACTOR_TOL[dtype]
    for task in (0, 2):
        rep, frep = Report(tol), Report(ftol)
        x, y = synth.synth_batch(12, 32, ncls[task], seed=40 + task)
        x, y = x.to(DEV), y.to(DEV)
        logits, ws = model(x, task=task)
        check(logits, g[f"t{task}.logits"], tol, "logits", report=rep)
        check(ws[0], g[f"t{task}.probs0"], tol, "probs0", report=rep)
        model.zero_grad(set_to_none=True)
        loss = model(x, y, task=task)
        loss.backward()
        assert abs(float(loss) - float(g[f"t{task}.loss"])) <= tol * abs(float(g[f"t{task}.loss"]))
        for k, p in model.named_parameters():
            ref = g[f"t{task}.grad.{k}"]
            if p.grad is None:
                assert float(np.abs(ref).max()) == 0.0, k        # other tasks' factors and heads
                continue
            check(p.grad, ref, tol, "grad " + k, report=frep if _is_factor(k) else rep)
        rep.finish(f"[{dtype}] rkr tiny task {task}")
        frep.finish(f"[{dtype}] rkr tiny task {task} rank-K factors")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rkr_b16_against_reference_golden(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, "rkr_b16_bs8.npz"))
    ncls = [10, 6]
    model = build_rkr("B_16", 224, ncls, 23, dtype)
    x, y = synth.synth_batch(8, 224, ncls[1], seed=61)
    x, y = x.to(DEV), y.to(DEV)
    tol = TOL[dtype]
    rep, frep = Report(tol), Report(FACTOR_TOL[dtype] if dtype == torch.float32 else 3e-2)    # ViT-B/16: 1.8e-2 observed
    logits, _ = model(x, task=1)
    check(logits, g["logits"], tol, "logits", report=rep)
    loss = model(x, y, task=1)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= tol * abs(float(g["loss"]))
    for k, p in model.named_parameters():
        idx = torch.from_numpy(g[f"grad.{k}__idx"])
        amax = float(g[f"grad.{k}__amax"])
        if p.grad is None:
            assert amax == 0.0, k
            continue
        got = p.grad.detach().float().cpu().reshape(-1)[idx].double()
        ref = torch.from_numpy(g[f"grad.{k}__val"]).double()
        (frep if _is_factor(k) else rep).add("grad " + k, float((got - ref).abs().max()), amax)
    rep.finish(f"[{dtype}] rkr b16")
    frep.finish(f"[{dtype}] rkr b16 rank-K factors")


@pytest.mark.parametrize("O,I,K,sfg,frozen", [(96, 64, 2, True, False), (768, 3072, 2, True, True), (40, 12, 3, False, False)])
def test_rkr_fold_matches_torch_autograd(O, I, K, sfg, frozen):
    """Fused weight fold (vitk_rkr_fold_fwd / _bwd) against the tensor-op formulation of reference
    modeling_RKR.py:56-78: W_t = (W + (LM.RM)^T) * F[:, None], b_t = b * F, and all five gradients."""
    from vitk import ops
    g = torch.Generator().manual_seed(O + I)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    W, b, LM, RM = mk(O, I), mk(O), mk(I, K), mk(K, O)
    F = (1.0 + 0.2 * mk(O)) if sfg else None
    dWt, dbt = mk(O, I), mk(O)
    leaves = [t.clone().requires_grad_(not (frozen and n == 0)) if t is not None else None
              for n, t in enumerate((W, b, LM, RM, F))]
    ref = [t.clone().requires_grad_(True) if t is not None else None for t in (W, b, LM, RM, F)]
    Wt, bt = ops.rkr_fold(*leaves)
    wr = ref[0] + torch.matmul(ref[2], ref[3]).permute(1, 0)
    br = ref[1]
    if sfg:
        wr, br = wr * ref[4].unsqueeze(1), br * ref[4]
    (Wt * dWt).sum().add((bt * dbt).sum()).backward()
    (wr * dWt).sum().add((br * dbt).sum()).backward()
    tol = lambda t: 2e-5 * float(t.abs().max()) + 1e-6  # noqa: E731
    assert float((Wt - wr).abs().max()) <= tol(wr) and float((bt - br).abs().max()) <= tol(br)
    for n, (a, r) in enumerate(zip(leaves, ref)):
        if a is None:
            continue
        if frozen and n == 0:
            assert a.grad is None
            continue
        assert float((a.grad - r.grad).abs().max()) <= tol(r.grad), ("W", "b", "LM", "RM", "F")[n]
