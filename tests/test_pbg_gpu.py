"""Filter-bank variants (reference models/ViT/modeling_PBG.py, modeling_RKRPBG.py) on the HIP kernels against the
golden vectors of the imported reference files: 1e-5 fp32; bf16 3e-2 for the plain bank (observed 1.9e-2) and 6e-2 for the factorised one (observed 4.7e-2;
torch.autocast(bfloat16) applied to the oracle on the same inputs moves these fixtures' gradients by 4e-2 .. 9e-2:
the rank-K factors and the feature scales multiply into every weight, as in the RKR variant)."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from test_model_gpu import DEV, TOL, Report, check
from test_pbg_cpu import CASES, NCLS, build, params_for

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,name,task,lamb", CASES)
def test_bank_variants_against_reference_golden(golden_dir, dtype, kind, name, task, lamb):
    import vitk
    g = np.load(os.path.join(golden_dir, f"{kind}_tiny.npz"))
    vitk.set_compute_dtype(dtype)
    model = build(kind, task, lamb)
    model.load_state_dict(params_for(kind, task, lamb), strict=True)
    model = model.to(DEV).train()
    # bf16: the plain filter bank at 3e-2 (observed 1.9e-2), the factorised one at 6e-2 (observed 4.7e-2)
    tol = TOL[dtype] if dtype == torch.float32 else (3e-2 if kind == "pbg" else 6e-2)
    rep = Report(tol)
    x, y = synth.synth_batch(12, 32, NCLS[task], seed=90 + task)
    x, y = x.to(DEV), y.to(DEV)
    logits, ws = model(x, task=task)
    check(logits, g[f"{name}.logits"], tol, "logits", report=rep)
    check(ws[0], g[f"{name}.probs0"], tol, "probs0", report=rep)
    loss = model(x, y, task=task)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"{name}.loss"])) <= tol * abs(float(g[f"{name}.loss"]))
    for k, p in model.named_parameters():
        key = f"{name}.grad.{k}"
        if p.grad is None:
            assert float(g[key + "__amax"]) == 0.0, k
            continue
        idx = torch.from_numpy(g[key + "__idx"]).to(DEV)
        got = p.grad.reshape(-1)[idx].float().cpu().numpy()
        rep.add(k, float(np.abs(got - g[key + "__val"]).max()), float(g[key + "__amax"]))
    rep.finish(f"{kind} {name} {dtype}")
