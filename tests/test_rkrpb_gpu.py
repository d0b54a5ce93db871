"""RKR+piggyback variant (shared rank-K factors specialised by thresholded per-task masks, folded
into the GEMM weight operand) on the HIP kernels against the golden vectors of reference
models/ViT/modeling_RKRPB.py + layers_PB.py: 1e-5 fp32; bf16 at 2.5e-2 (60-token tiny
fixtures: observed 1.8e-2) (weight-shaped gradients are sums of 60 bf16-rounded outer products); the gradients of the
rank-K factors and of their masks are contractions of such a weight gradient with the other factor
(128 signed terms that partly cancel), and are held to 5e-2 of their own largest entry in bf16 (observed up
to 4.4e-2, always on the key projection of layer 0: see tests/test_rkr_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from test_model_gpu import DEV, TOL, Report, check
from test_rkrpb_cpu import CASES, IDS, NCLS, rkrpb_config

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name,flags,fn,task,seed", CASES, ids=IDS)
def test_rkrpb_tiny_against_reference_golden(golden_dir, dtype, name, flags, fn, task, seed):
    import vitk
    from models.ViT.modeling_RKRPB import VisionTransformer
    g = np.load(os.path.join(golden_dir, "rkrpb_tiny.npz"))
    vitk.set_compute_dtype(dtype)
    model = VisionTransformer(rkrpb_config(flags, fn), 32, num_classes=NCLS, zero_head=False, vis=True)
    model.load_state_dict(synth.synth_rkrpb_params("tiny", 32, NCLS, seed=seed, ternary=fn == "ternarizer", **flags),
                          strict=True)
    model = model.to(DEV).train()
    tol = TOL[dtype] if dtype == torch.float32 else 2.5e-2      # observed 1.8e-2; factors and masks: 5e-2 (4.2e-2)
    rep = Report(tol)
    x, y = synth.synth_batch(12, 32, NCLS[task], seed=90 + task)
    x, y = x.to(DEV), y.to(DEV)
    logits, ws = model(x, task=task)
    check(logits, g[f"{name}.t{task}.logits"], tol, "logits", report=rep)
    check(ws[0], g[f"{name}.t{task}.probs0"], tol, "probs0", report=rep)
    model.zero_grad(set_to_none=True)
    loss = model(x, y, task=task)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"{name}.t{task}.loss"])) <= tol * abs(float(g[f"{name}.t{task}.loss"]))
    seen = 0
    for k, p in model.named_parameters():
        key = f"{name}.t{task}.grad.{k}"
        amax = float(g[key + "__amax"])
        if p.grad is None:
            assert amax == 0.0, k                       # other tasks' heads / masks / lists, dead q-k-v scales
            continue
        idx = torch.from_numpy(g[key + "__idx"]).to(DEV)
        got = p.grad.reshape(-1)[idx].float().cpu().numpy()
        factor = any(t in k for t in ("LM_", "RM_", "mask_reals")) and dtype == torch.bfloat16
        rep.add(k, float(np.abs(got - g[key + "__val"]).max()), amax * (2.0 if factor else 1.0))
        seen += amax > 0
    assert seen > 20
    rep.finish(f"rkrpb {name} task {task} {dtype}")
