"""Piggyback variant on the CPU: the oracle's restatement of reference models/ViT/modeling_PB.py +
layers_PB.py against the golden vectors produced by importing those files (oracle/gen_golden.py pb),
and the drop-in module's parameter names / shapes / thresholding against the same state_dict layout."""
import os

import numpy as np
import pytest
import torch

from oracle import synth, vit_oracle as O

RTOL = 2e-5
CASES = [("binarizer", 0, 29), ("binarizer", 2, 29), ("ternarizer", 1, 31)]
NCLS = [10, 7, 5]


def _close(a, b, what, rtol=RTOL, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, lim = np.abs(a - b).max(), rtol * np.abs(b).max() + atol
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


@pytest.fixture(autouse=True)
def _restore_threshold_fn():
    yield
    O.PB_THRESHOLD_FN = "binarizer"


@pytest.mark.parametrize("fn,task,seed", CASES)
def test_oracle_pb_tiny_against_reference(golden_dir, fn, task, seed):
    g = np.load(os.path.join(golden_dir, "pb_tiny.npz"))
    O.PB_THRESHOLD_FN = fn
    params = synth.synth_pb_params("tiny", 32, NCLS, seed=seed, ternary=fn == "ternarizer")
    cfg = synth.oracle_cfg("tiny")
    x, y = synth.synth_batch(12, 32, NCLS[task], seed=70 + task)
    logits, ws = O.rkr_vit_forward(params, x, cfg, task, vis=True)
    _close(logits.detach(), g[f"{fn}.t{task}.logits"], "logits")
    _close(ws[0].detach(), g[f"{fn}.t{task}.probs0"], "probs0")
    loss, _, grads = O.rkr_loss_and_grads(params, x, y, cfg, task)
    assert abs(float(loss) - float(g[f"{fn}.t{task}.loss"])) <= RTOL * abs(float(g[f"{fn}.t{task}.loss"]))
    seen_mask = False
    for k, v in grads.items():
        if ".mask_reals." in k and not k.endswith(f".{task}"):
            assert float(v.abs().max()) == 0.0, k          # other tasks' masks receive no gradient
            continue
        name = f"{fn}.t{task}.grad.{k}"
        got = v.reshape(-1)[torch.from_numpy(g[name + "__idx"])].numpy()
        amax = float(g[name + "__amax"])
        assert np.abs(got - g[name + "__val"]).max() <= RTOL * amax + 1e-7, k
        seen_mask |= ".mask_reals." in k and amax > 0
    assert seen_mask


def test_pb_thresholding_functions():
    from models.ViT.layers_PB import Binarizer, Ternarizer
    x = torch.tensor([-1.0, -1e-4, 0.0, 4e-3, 5e-3, 5.1e-3, 0.3], requires_grad=True)
    b = Binarizer.apply(x)
    assert b.tolist() == [0, 0, 0, 0, 0, 1, 1]
    t = Ternarizer.apply(x)
    assert t.tolist() == [-1, -1, 0, 0, 0, 1, 1]
    (b * torch.arange(7.0)).sum().backward()
    assert x.grad.tolist() == list(range(7))                # straight-through


def test_pb_module_state_dict_layout():
    """Same parameter names and shapes as the reference piggyback model (the training driver selects
    trainable parameters by substrings such as 'mask_reals', 'head')."""
    from models import config as cfgmod
    from models.ViT.layers_PB import ElementWiseLinear
    from models.ViT.modeling_PB import VisionTransformer
    cfg = cfgmod.get_tiny_config()
    cfg.threshold_fn, cfg.mask_scale, cfg.mask_init = "binarizer", 1e-2, "1s"
    cfg.multi_head, cfg.task_num = True, len(NCLS)
    model = VisionTransformer(cfg, 32, num_classes=NCLS, zero_head=False)
    want = synth.synth_pb_params("tiny", 32, NCLS, seed=29)
    have = model.state_dict()
    assert set(have) == set(want)
    for k, v in want.items():
        assert tuple(have[k].shape) == tuple(v.shape), k
    fc = model.transformer.encoder.layer[0].attn.query
    assert isinstance(fc, ElementWiseLinear)
    one = float(torch.tensor(1e-2))
    assert all(float(m.detach().min()) == float(m.detach().max()) == one for m in fc.mask_reals)   # mask_init '1s'
    model.load_state_dict(want, strict=True)
    w, b = fc.effective(2)
    pre = "transformer.encoder.layer.0.attn.query."
    hard = (want[pre + "mask_reals.2"] > 5e-3).float()
    assert 0.3 < float(hard.mean()) < 0.9
    assert torch.equal(w, hard * want[pre + "weight"]) and torch.equal(b, want[pre + "bias"])
    cfg.mask_init = "uniform"
    m = ElementWiseLinear(8, 4, cfg).mask_reals[0]
    m = m.detach()
    assert float(m.abs().max()) <= one and float(m.min()) < 0 < float(m.max())
    cfg.threshold_fn = "sign"
    with pytest.raises(ValueError):
        ElementWiseLinear(8, 4, cfg)
