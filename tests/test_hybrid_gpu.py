"""R50-style hybrid on the GPU against the golden vectors of the reference's hybrid branch.

The convolutional stem runs on PyTorch-ROCm's own operators (MIOpen convolutions, GroupNorm), not on
this package's kernels, always in fp32; what is under test here is the hand-over in both directions:
the feature map becomes tokens through the kernel library's linear operator, the encoder runs as in
the plain model, and the gradient of the feature map that the linear operator's dX GEMM hands back to
the stem (`dfeat`) is compared with the reference's.  Tolerances: fp32 instance 2e-5 (a shade over
the 1e-5 of the plain model: the feature map itself already differs from the CPU's by 3e-6 after nine
stock convolution layers); bf16 instance the 3e-2 of the other small-token fixtures (17 tokens per
image).

The stem's OWN parameter gradients are not pinned by this test ("parity unpinned" for those
tensors): on this stack (PyTorch 2.10 + ROCm 7.0, MI355X) the stock fp32 convolution backward gives
weight gradients that differ from the CPU's by 1-8 % in about every second process, with identical
inputs and an identical, verified upstream gradient (tools/stem_check.py probes operators and the
stem alone).  They are held to 0.15 only, which still catches a wrong or missing upstream gradient."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle.gen_golden import HYBRID_TINY as H
from test_model_gpu import DEV, Report, check

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 2e-5, torch.bfloat16: 3e-2}
STEM_GRAD_TOL = 0.15


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hybrid_tiny_against_reference_golden(golden_dir, dtype):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    g = np.load(os.path.join(golden_dir, "hybrid_tiny.npz"))
    vitk.set_compute_dtype(dtype)
    cfg = cfgmod.get_tiny_config()
    del cfg.patches["size"]
    cfg.patches.grid = (H["img"] // 16, H["img"] // 16)
    cfg.resnet = type(cfg)(dict(num_layers=H["block_units"], width_factor=H["width_factor"]))
    model = VisionTransformer(cfg, H["img"], num_classes=H["ncls"], zero_head=False, vis=True)
    model.load_state_dict(synth.synth_hybrid_params("tiny", H["img"], H["ncls"], H["block_units"],
                                                    H["width_factor"], seed=H["seed"]), strict=True)
    model = model.to(DEV).train()
    tol = TOL[dtype]
    rep, stem_rep = Report(tol), Report(STEM_GRAD_TOL)
    x, y = synth.synth_batch(H["batch"], H["img"], H["ncls"], seed=80)
    x, y = x.to(DEV), y.to(DEV)
    stem = model.transformer.embeddings.hybrid_model
    check(stem(x), g["feat"], 2e-5, "stem feature map", report=rep)
    logits, ws = model(x)
    check(logits, g["logits"], tol, "logits", report=rep)
    check(ws[0], g["probs0"], tol, "probs0", report=rep)
    kept = {}

    def keep_grad(mod, inp, out):
        out.retain_grad()
        kept["feat"] = out
    handle = stem.register_forward_hook(keep_grad)
    loss = model(x, y)
    loss.backward()
    handle.remove()
    assert abs(float(loss.detach()) - float(g["loss"])) <= tol * abs(float(g["loss"]))
    check(kept["feat"].grad, g["dfeat"], tol, "gradient of the stem feature map", report=rep)
    for k, p in model.named_parameters():
        name = "grad." + k
        idx = torch.from_numpy(g[name + "__idx"]).to(DEV)
        got = p.grad.reshape(-1)[idx].float().cpu().numpy()
        (stem_rep if ".hybrid_model." in k else rep).add(k, float(np.abs(got - g[name + "__val"]).max()),
                                                          float(g[name + "__amax"]))
    rep.finish(f"hybrid {dtype}")
    stem_rep.finish(f"hybrid stem parameters {dtype}")
