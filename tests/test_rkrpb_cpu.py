"""RKR+piggyback variant on the CPU: the oracle's restatement of reference models/ViT/modeling_RKRPB.py
+ layers_PB.py (ElementWiseRG / ElementWiseSFG) against the golden vectors produced by importing those
files (oracle/gen_golden.py rkrpb), every configuration branch of ElementWiseRG.forward, and the
drop-in module's parameter names / shapes / weight folding against the same state_dict layout."""
import os

import numpy as np
import pytest
import torch

from oracle import synth, vit_oracle as O

RTOL = 2e-5
NCLS = [10, 7, 5]
CASES = [(name, flags, fn, task, seed) for name, flags, fn, tasks, seed in synth.RKRPB_CASES for task in tasks]
IDS = [f"{c[0]}-t{c[3]}" for c in CASES]


def _close(a, b, what, rtol=RTOL, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, lim = np.abs(a - b).max(), rtol * np.abs(b).max() + atol
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


@pytest.fixture(autouse=True)
def _restore_threshold_fn():
    yield
    O.PB_THRESHOLD_FN = "binarizer"


def rkrpb_config(flags, fn, dropout=0.0):
    from models import config as cfgmod
    cfg = cfgmod.get_tiny_config()
    cfg.transformer.dropout_rate = dropout
    cfg.RG, cfg.SFG, cfg.K, cfg.rkr_scale, cfg.multi_head, cfg.task_num = True, True, 2, 1e-1, True, len(NCLS)
    cfg.threshold_fn, cfg.mask_scale, cfg.mask_init = fn, 1e-2, "1s"
    full = dict(RGnoPB=False, LMnoPB=False, RMnoPB=False, SFGnoPB=False, PBwR=False)
    full.update(flags)
    for k, v in full.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("name,flags,fn,task,seed", CASES, ids=IDS)
def test_oracle_rkrpb_tiny_against_reference(golden_dir, name, flags, fn, task, seed):
    g = np.load(os.path.join(golden_dir, "rkrpb_tiny.npz"))
    O.PB_THRESHOLD_FN = fn
    params = synth.synth_rkrpb_params("tiny", 32, NCLS, seed=seed, ternary=fn == "ternarizer", **flags)
    cfg = synth.oracle_cfg("tiny")
    x, y = synth.synth_batch(12, 32, NCLS[task], seed=90 + task)
    logits, ws = O.rkr_vit_forward(params, x, cfg, task, vis=True)
    _close(logits.detach(), g[f"{name}.t{task}.logits"], "logits")
    _close(ws[0].detach(), g[f"{name}.t{task}.probs0"], "probs0")
    loss, _, grads = O.rkr_loss_and_grads(params, x, y, cfg, task)
    assert abs(float(loss) - float(g[f"{name}.t{task}.loss"])) <= RTOL * abs(float(g[f"{name}.t{task}.loss"]))
    live = set()
    for k, v in grads.items():
        key = f"{name}.t{task}.grad.{k}"
        got = v.reshape(-1)[torch.from_numpy(g[key + "__idx"])].numpy()
        amax = float(g[key + "__amax"])
        assert np.abs(got - g[key + "__val"]).max() <= RTOL * amax + 1e-7, k
        if amax > 0:
            live.add(k.split(".")[-2] if k.split(".")[-1].isdigit() else k.split(".")[-1])
    # the scales of query / key / value never act in this model (reference modeling_RKRPB.py:90-97)
    assert all(float(v.abs().max()) == 0.0 for k, v in grads.items() if ".SFG_query." in k or ".SFG_value." in k)
    if task > 0 and not flags.get("RGnoPB"):
        assert live & {"mask_reals", "mask_reals_LM", "mask_reals_RM"}, live


@pytest.mark.parametrize("name,flags,fn,task,seed", CASES, ids=IDS)
def test_rkrpb_effective_weight_matches_oracle(name, flags, fn, task, seed):
    """The weight operand the module hands to the fused GEMM is the one the reference multiplies by."""
    from models.ViT.modeling_RKRPB import VisionTransformer
    O.PB_THRESHOLD_FN = fn
    model = VisionTransformer(rkrpb_config(flags, fn), 32, num_classes=NCLS, zero_head=False)
    want = synth.synth_rkrpb_params("tiny", 32, NCLS, seed=seed, ternary=fn == "ternarizer", **flags)
    have = model.state_dict()
    assert set(have) == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    model.load_state_dict(want, strict=True)
    blk = model.transformer.encoder.layer[1]
    eye = torch.eye(blk.hidden_size)
    for fc, sfg, pre, spre in ((blk.attn.out, blk.attn.SFG_out, "attn.out.", "attn.SFG_out."),
                               (blk.ffn.fc1, blk.ffn.SFG1, "ffn.fc1.", "ffn.SFG1."),
                               (blk.attn.key, None, "attn.key.", None)):
        p = "transformer.encoder.layer.1."
        w, b = fc.effective(task, sfg)
        # oracle: rows of the identity through the layer give W_t^T + b
        y = O.rg_fc(want, eye, p + pre, task)
        if spre is not None:
            y = O.sfg_fc(want, y, p + spre, task)
        zero = O.rg_fc(want, torch.zeros(1, blk.hidden_size), p + pre, task)
        if spre is not None:
            zero = O.sfg_fc(want, zero, p + spre, task)
        _close(b.detach(), zero[0].detach(), pre + "bias", rtol=1e-6)
        _close(w.detach().t(), (y - zero).detach(), pre + "weight", rtol=1e-5, atol=1e-6)


def test_rkrpb_layers_init_and_errors():
    from models.ViT.layers_PB import ElementWiseRG, ElementWiseSFG
    cfg = rkrpb_config({}, "binarizer")
    fc = ElementWiseRG(8, 4, cfg)
    one = float(torch.tensor(1e-2))
    assert len(fc.mask_reals_LM) == len(fc.mask_reals_RM) == len(NCLS) - 1
    assert all(float(m.detach().min()) == float(m.detach().max()) == one for m in list(fc.mask_reals_LM) + list(fc.mask_reals_RM))
    assert tuple(fc.LM_base.shape) == (8, 2) and tuple(fc.RM_base.shape) == (2, 4)
    sf = ElementWiseSFG(4, cfg)
    assert torch.equal(sf.scale(0), sf.F_base) and len(sf.mask_reals_F) == len(NCLS) - 1
    assert torch.equal(sf.scale(1).detach(), torch.ones(4))           # mask 1e-2 > threshold: all kept
    cfg.threshold_fn = "sign"
    with pytest.raises(ValueError):
        ElementWiseRG(8, 4, cfg)
