"""RKR-TSN variant on the CPU: the oracle's restatement of reference models/ViT/modeling_RKRTSN.py (rank-K
factors generated from a task embedding, L1 distillation of the embeddings of earlier tasks) against
the golden vectors produced by importing that file (oracle/gen_golden.py rkrtsn), and the drop-in
module's parameter names / shapes against the same state_dict layout."""
import os

import numpy as np
import pytest
import torch

from oracle import synth, vit_oracle as O

RTOL = 2e-5
NCLS = [10, 7, 5]
CASES = [("sfg", True, 67), ("nosfg", False, 71)]


def _close(a, b, what, rtol=RTOL, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, lim = np.abs(a - b).max(), rtol * np.abs(b).max() + atol
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


def tsn_config(sfg, dropout=0.0):
    from models import config as cfgmod
    cfg = cfgmod.get_tiny_config()
    cfg.transformer.dropout_rate = dropout
    cfg.RG, cfg.SFG, cfg.K, cfg.multi_head, cfg.task_num, cfg.task_emb_d, cfg.lamb = True, sfg, 2, True, len(NCLS), 6, 1.0
    return cfg


@pytest.mark.parametrize("name,sfg,seed", CASES)
def test_oracle_rkrtsn_tiny_against_reference(golden_dir, name, sfg, seed):
    g = np.load(os.path.join(golden_dir, "rkrtsn_tiny.npz"))
    params = synth.synth_rkrtsn_params("tiny", 32, NCLS, SFG=sfg, seed=seed)
    cfg = synth.oracle_cfg("tiny")
    tv = synth.synth_task_vecs(len(NCLS), 6, seed=seed)
    emb_labels = []
    for task in (0, 1):
        if task == 1:
            params["task_embedding.fc.7.bias"] = params["task_embedding.fc.7.bias"] + synth.tsn_drift(len(NCLS))
        x, y = synth.synth_batch(12, 32, NCLS[task], seed=110 + task)
        logits, ws, embs = O.tsn_forward(params, x, cfg, task, tv, vis=True)
        _close(logits.detach(), g[f"{name}.t{task}.logits"], "logits")
        _close(ws[0].detach(), g[f"{name}.t{task}.probs0"], "probs0")
        _close(embs[task].detach(), g[f"{name}.t{task}.emb_label"], "embedding")
        loss, distill, _, grads = O.tsn_loss_and_grads(params, x, y, cfg, task, tv, emb_labels)
        emb_labels.append(embs[task].detach())
        assert abs(float(loss) - float(g[f"{name}.t{task}.loss"])) <= RTOL * abs(float(g[f"{name}.t{task}.loss"]))
        assert abs(float(distill) - float(g[f"{name}.t{task}.distill"])) <= 1e-6
        live = 0
        for k, v in grads.items():
            key = f"{name}.t{task}.grad.{k}"
            got = v.reshape(-1)[torch.from_numpy(g[key + "__idx"])].numpy()
            amax = float(g[key + "__amax"])
            assert np.abs(got - g[key + "__val"]).max() <= RTOL * amax + 1e-7, k
            live += ("LM_fc" in k or "RM_fc" in k or "task_embedding" in k) and amax > 0
        assert live >= 16 + 4 * 12
    assert float(g[f"{name}.t1.distill"]) > 0.01          # the distillation term was live in the golden run


def test_rkrtsn_module_state_dict_layout():
    from models.ViT.modeling_RKRTSN import RG_FC, VisionTransformer
    model = VisionTransformer(tsn_config(True), 32, num_classes=NCLS, zero_head=False)
    want = synth.synth_rkrtsn_params("tiny", 32, NCLS, SFG=True, seed=67)
    have = model.state_dict()
    assert set(have) == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    assert "transformer.encoder.layer.0.attn.SFG_out.F_list.2" in have
    no_sfg = VisionTransformer(tsn_config(False), 32, num_classes=NCLS, zero_head=False).state_dict()
    assert set(no_sfg) == set(synth.synth_rkrtsn_params("tiny", 32, NCLS, SFG=False, seed=71))
    # the weight operand handed to the GEMM is W + (LM.RM)^T with LM, RM from the embedding
    model.load_state_dict(want, strict=True)
    fc = model.transformer.encoder.layer[1].ffn.fc1
    assert isinstance(fc, RG_FC)
    emb = torch.tensor([0.3, -0.2, 0.5])
    w, b = fc.effective((1, emb))
    p = dict(want)
    p[O.TASK_EMB] = emb
    pre = "transformer.encoder.layer.1.ffn.fc1."
    eye = torch.eye(fc.h_in)
    zero = O.rg_fc(p, torch.zeros(1, fc.h_in), pre, 1)
    _close(w.detach().t(), (O.rg_fc(p, eye, pre, 1) - zero).detach(), "weight", rtol=1e-5, atol=1e-6)
    _close(b.detach(), zero[0].detach(), "bias", rtol=1e-6)
