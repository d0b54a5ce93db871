"""RKR variant on the CPU: the oracle's restatement of reference models/ViT/modeling_RKR.py against the
golden vectors produced by importing that file (oracle/gen_golden.py rkr), and the drop-in module's
parameter names / shapes against the same state_dict layout."""
import os

import numpy as np
import torch

from oracle import synth, vit_oracle as O

RTOL = 2e-5


def _close(a, b, what, rtol=RTOL, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, lim = np.abs(a - b).max(), rtol * np.abs(b).max() + atol
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


def test_oracle_rkr_tiny_against_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "rkr_tiny.npz"))
    ncls = [10, 7, 5]
    params = synth.synth_rkr_params("tiny", 32, ncls, seed=17)
    cfg = synth.oracle_cfg("tiny")
    for task in (0, 2):
        x, y = synth.synth_batch(12, 32, ncls[task], seed=40 + task)
        logits, ws = O.rkr_vit_forward(params, x, cfg, task, vis=True)
        _close(logits.detach(), g[f"t{task}.logits"], f"t{task} logits")
        _close(ws[0].detach(), g[f"t{task}.probs0"], f"t{task} probs0")
        loss, _, grads = O.rkr_loss_and_grads(params, x, y, cfg, task)
        assert abs(float(loss) - float(g[f"t{task}.loss"])) <= RTOL * abs(float(g[f"t{task}.loss"]))
        for k, v in grads.items():
            _close(v, g[f"t{task}.grad.{k}"], f"t{task} grad {k}")
        # parameters of the other tasks receive no gradient
        other = 2 - task
        assert float(grads[f"transformer.encoder.layer.0.attn.query.LM_list.{other}"].abs().max()) == 0.0


def test_oracle_rkr_b16_against_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "rkr_b16_bs8.npz"))
    ncls = [10, 6]
    params = synth.synth_rkr_params("B_16", 224, ncls, seed=23)
    x, y = synth.synth_batch(8, 224, ncls[1], seed=61)
    loss, logits, grads = O.rkr_loss_and_grads(params, x, y, synth.oracle_cfg("B_16"), 1)
    _close(logits, g["logits"], "logits")
    assert abs(float(loss) - float(g["loss"])) <= RTOL * abs(float(g["loss"]))
    for k, v in grads.items():
        idx = g[f"grad.{k}__idx"]
        got = v.reshape(-1)[torch.from_numpy(idx)].numpy()
        amax = float(g[f"grad.{k}__amax"])
        assert np.abs(got - g[f"grad.{k}__val"]).max() <= 5e-5 * amax + 1e-7, k


def test_rkr_module_state_dict_layout():
    """Same parameter names and shapes as the reference RKR model (names are API: the training driver
    selects trainable parameters by substrings such as 'LM_list', 'F_list', 'head')."""
    from models import config as cfgmod
    from models.ViT.modeling_RKR import RG_FC, SFG_FC, VisionTransformer
    ncls = [10, 7, 5]
    cfg = cfgmod.get_tiny_config()
    cfg.RG, cfg.SFG, cfg.K, cfg.rkr_scale, cfg.multi_head, cfg.task_num = True, True, 2, 1e-1, True, len(ncls)
    model = VisionTransformer(cfg, 32, num_classes=ncls, zero_head=False)
    want = synth.synth_rkr_params("tiny", 32, ncls, seed=17)
    have = model.state_dict()
    assert set(have) == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    model.load_state_dict(want, strict=True)
    fc = model.transformer.encoder.layer[0].attn.query
    assert isinstance(fc, RG_FC) and isinstance(model.transformer.encoder.layer[0].attn.SFG_query, SFG_FC)
    # the folded operand equals diag(F) (W + (LM RM)^T), F * b
    w, b = fc.effective(1, model.transformer.encoder.layer[0].attn.SFG_query)
    f = want["transformer.encoder.layer.0.attn.SFG_query.F_list.1"]
    pre = "transformer.encoder.layer.0.attn.query."
    ref_w = (want[pre + "weight"] + (want[pre + "LM_list.1"] @ want[pre + "RM_list.1"]).t()) * f[:, None]
    assert torch.allclose(w, ref_w, rtol=1e-6, atol=1e-7) and torch.allclose(b, want[pre + "bias"] * f)
    # variants without RG / SFG carry no such parameters
    cfg.RG, cfg.SFG = False, False
    plain = VisionTransformer(cfg, 32, num_classes=ncls)
    assert not any("LM_list" in k or "F_list" in k for k in plain.state_dict())
