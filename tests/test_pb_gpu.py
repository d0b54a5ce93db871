"""Piggyback variant (thresholded per-task masks folded into the GEMM weight operand) on the HIP
kernels against the golden vectors of reference models/ViT/modeling_PB.py + layers_PB.py: 1e-5 fp32;
bf16 at the 2e-2 contract (observed 1.4e-2)."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from test_model_gpu import DEV, TOL, Report, check

pytestmark = pytest.mark.gpu
NCLS = [10, 7, 5]


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


def build_pb(fn, seed, dtype, vis=False):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling_PB import VisionTransformer
    vitk.set_compute_dtype(dtype)
    cfg = cfgmod.get_tiny_config()
    cfg.transformer.dropout_rate = 0.0
    cfg.threshold_fn, cfg.mask_scale, cfg.mask_init = fn, 1e-2, "1s"
    cfg.multi_head, cfg.task_num = True, len(NCLS)
    model = VisionTransformer(cfg, 32, num_classes=NCLS, zero_head=False, vis=vis)
    model.load_state_dict(synth.synth_pb_params("tiny", 32, NCLS, seed=seed, ternary=fn == "ternarizer"), strict=True)
    return model.to(DEV).train()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fn,task,seed", [("binarizer", 0, 29), ("binarizer", 2, 29), ("ternarizer", 1, 31)])
def test_pb_tiny_against_reference_golden(golden_dir, dtype, fn, task, seed):
    g = np.load(os.path.join(golden_dir, "pb_tiny.npz"))
    model = build_pb(fn, seed, dtype, vis=True)
    tol = TOL[dtype]          # bf16: the 2e-2 contract (observed 1.4e-2)
    rep = Report(tol)
    x, y = synth.synth_batch(12, 32, NCLS[task], seed=70 + task)
    x, y = x.to(DEV), y.to(DEV)
    logits, ws = model(x, task=task)
    check(logits, g[f"{fn}.t{task}.logits"], tol, "logits", report=rep)
    check(ws[0], g[f"{fn}.t{task}.probs0"], tol, "probs0", report=rep)
    model.zero_grad(set_to_none=True)
    loss = model(x, y, task=task)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"{fn}.t{task}.loss"])) <= tol * abs(float(g[f"{fn}.t{task}.loss"]))
    for k, p in model.named_parameters():
        if ".mask_reals." in k and not k.endswith(f".{task}"):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k     # other tasks' masks
            continue
        name = f"{fn}.t{task}.grad.{k}"
        if p.grad is None:
            assert float(g[name + "__amax"]) == 0.0, k                       # other tasks' heads
            continue
        idx = torch.from_numpy(g[name + "__idx"]).to(DEV)
        got = p.grad.reshape(-1)[idx].float().cpu().numpy()
        e = float(np.abs(got - g[name + "__val"]).max())
        rep.add(k, e, float(g[name + "__amax"]))
    rep.finish(f"pb {fn} task {task} {dtype}")


@pytest.mark.parametrize("ternary", [False, True])
@pytest.mark.parametrize("shape,frozen", [((96, 128), True), ((7, 9), False), ((3072,), False), ((768, 2), True)])
def test_mask_fold_against_formula(ternary, shape, frozen):
    """vitk_mask_fold_fwd/bwd (threshold * tensor, straight-through gradients) against the reference's
    Binarizer / Ternarizer arithmetic (layers_PB.py:14-54) written with tensor ops; sizes with and without
    a tail of fewer than four elements; masked tensor frozen (the piggyback weight) or trained (a factor)."""
    from models.ViT.layers_PB import DEFAULT_THRESHOLD
    from vitk import ops
    g = torch.Generator().manual_seed(sum(shape))
    m = (torch.randn(shape, generator=g) * 1e-2).to(DEV)
    m.view(-1)[0], m.view(-1)[1] = DEFAULT_THRESHOLD, 0.0          # the boundary values themselves
    m.requires_grad_(True)
    w = torch.randn(shape, generator=g).to(DEV).requires_grad_(not frozen)
    up = torch.randn(shape, generator=g).to(DEV)
    out = ops.mask_fold(m, w, ternary, DEFAULT_THRESHOLD)
    out.backward(up)
    md = m.detach()
    hard = (md > DEFAULT_THRESHOLD).float() - ((md < 0).float() if ternary else 0.0)
    assert torch.equal(out.detach(), hard * w.detach())
    assert torch.equal(m.grad, up * w.detach())
    if frozen:
        assert w.grad is None
    else:
        assert torch.equal(w.grad, up * hard)


def test_masked_linear_without_bias():
    """ElementWiseLinear(bias=False) (reference layers_PB.py:60-63,105-111: F.linear accepts bias None)."""
    import vitk
    from models import config as cfgmod
    from models.ViT.layers_PB import DEFAULT_THRESHOLD, ElementWiseLinear
    vitk.set_compute_dtype(torch.float32)
    cfg = cfgmod.get_tiny_config()
    cfg.threshold_fn, cfg.mask_scale, cfg.mask_init, cfg.task_num = "binarizer", 1e-2, "uniform", 2
    lin = ElementWiseLinear(64, 32, cfg, bias=False)
    torch.nn.init.normal_(lin.weight, std=0.1)
    lin = lin.to(DEV)
    x = torch.randn(5, 7, 64, device=DEV)
    y = lin(x, 1)
    want = torch.nn.functional.linear(x, (lin.mask_reals[1] > DEFAULT_THRESHOLD).float() * lin.weight)
    assert y.shape == (5, 7, 32) and torch.allclose(y.float(), want, rtol=1e-5, atol=1e-6)
