"""Filter-bank variants on the CPU: the oracle's restatement of reference models/ViT/modeling_PBG.py
(PiggybackFC) and modeling_RKRPBG.py (RG_FC / SFG_FC) against golden vectors produced by importing those files
(oracle/gen_golden.py pbg rkrpbg), and the drop-in modules' parameter / buffer names and shapes against the
same state_dict layout."""
import os

import numpy as np
import pytest
import torch

from oracle import synth, vit_oracle as O

RTOL = 2e-5
NCLS = [10, 7, 5]


def _close(a, b, what, rtol=RTOL, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, lim = np.abs(a - b).max(), rtol * np.abs(b).max() + atol
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


def params_for(kind, task, lamb):
    if kind == "pbg":
        return synth.synth_pbg_params("tiny", 32, NCLS[task], task, lamb, seed=67 + task)
    return synth.synth_rkrpbg_params("tiny", 32, NCLS[task], task, lamb, K=2, seed=71 + task)


CASES = [("pbg",) + c for c in synth.PBG_CASES] + [("rkrpbg",) + c for c in synth.RKRPBG_CASES]


@pytest.mark.parametrize("kind,name,task,lamb", CASES)
def test_oracle_against_reference(golden_dir, kind, name, task, lamb):
    g = np.load(os.path.join(golden_dir, f"{kind}_tiny.npz"))
    params = params_for(kind, task, lamb)
    buffers = {k for k in params if k.endswith("concat_unc_filter")}
    cfg = synth.oracle_cfg("tiny")
    x, y = synth.synth_batch(12, 32, NCLS[task], seed=90 + task)
    logits, ws = O.rkr_vit_forward(params, x, cfg, task, vis=True)
    _close(logits.detach(), g[f"{name}.logits"], "logits")
    _close(ws[0].detach(), g[f"{name}.probs0"], "probs0")
    loss, _, grads = O.rkr_loss_and_grads(params, x, y, cfg, task)
    assert abs(float(loss) - float(g[f"{name}.loss"])) <= RTOL * abs(float(g[f"{name}.loss"]))
    seen = 0
    for k, v in grads.items():
        if k in buffers:
            continue
        key = f"{name}.grad.{k}"
        got = v.reshape(-1)[torch.from_numpy(g[key + "__idx"])].numpy()
        assert np.abs(got - g[key + "__val"]).max() <= RTOL * float(g[key + "__amax"]) + 1e-7, k
        seen += ("weights_mat" in k or "unc_filt" in k) and float(g[key + "__amax"]) > 0
    assert seen > 0
    assert (lamb < 1.0 and task > 0) == bool(buffers)


def build(kind, task, lamb):
    from models import config as cfgmod
    cfg = cfgmod.get_tiny_config()
    cfg.transformer.dropout_rate = 0.0
    cfg.lamb, cfg.multi_head, cfg.task_num = lamb, False, len(NCLS)
    if kind == "pbg":
        from models.ViT.modeling_PBG import VisionTransformer
        return VisionTransformer(cfg, 32, num_classes=NCLS[task], zero_head=False, vis=True, task=task)
    from models.ViT.modeling_RKRPBG import VisionTransformer
    cfg.RG, cfg.SFG, cfg.K, cfg.rkr_scale = True, True, 2, 1e-1
    return VisionTransformer(cfg, 32, num_classes=NCLS, zero_head=False, vis=True, task=task)


@pytest.mark.parametrize("kind,name,task,lamb", CASES)
def test_drop_in_state_dict_layout(kind, name, task, lamb):
    model = build(kind, task, lamb)
    want = params_for(kind, task, lamb)
    have = model.state_dict()
    assert set(have) == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    model.load_state_dict(want, strict=True)
    named = dict(model.named_parameters())
    assert all(k in named for k in want if not k.endswith("concat_unc_filter"))
    assert not any(k.endswith("concat_unc_filter") for k in named)          # the bank is a buffer, not trained


def test_weight_of_a_later_task_follows_the_reference_reshapes():
    """The bank is reinterpreted, not transposed (modeling_PBG.py:67-69): checked on numbers small enough to
    follow by hand."""
    from models.ViT.modeling_PBG import bank_filters
    bank = torch.arange(12.0).view(4, 3)                    # C = 4 filters of 3 inputs
    wm = torch.eye(4)[:, :2]                                # picks "columns" 0 and 1 of the reshaped bank
    got = bank_filters(bank, wm, 3)
    reshaped = bank.reshape(3, 4)                           # rows [0..3], [4..7], [8..11]
    assert torch.equal(got, reshaped[:, :2].reshape(2, 3))
