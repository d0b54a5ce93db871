"""ViT-L/16 geometry at 384 px (577 tokens: the streaming attention kernels, hidden 1024, 16 heads,
MLP 4096) on the HIP kernels against the CPU oracle on the same seeded weights and batch.  Two of the
24 layers keep the oracle's CPU run to seconds; per-layer work and shapes are those of the preset."""
import pytest
import torch

from oracle import synth, vit_oracle as O
from test_model_gpu import DEV, TOL, Report, check

pytestmark = pytest.mark.gpu
GEOM = dict(hidden=1024, heads=16, mlp=4096, layers=2, patch=16)


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_l16_geometry_384px_against_oracle(dtype):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    vitk.set_compute_dtype(dtype)
    cfg = cfgmod.get_config("ViT-L_16")
    assert (cfg.hidden_size, cfg.transformer.num_heads, cfg.transformer.mlp_dim) == (1024, 16, 4096)
    cfg.transformer.num_layers = GEOM["layers"]
    cfg.transformer.dropout_rate = 0.0
    params = synth.synth_params(GEOM, 384, 10, seed=51)
    model = VisionTransformer(cfg, 384, num_classes=10, zero_head=False)
    model.load_state_dict(params, strict=True)
    model = model.to(DEV).train()
    assert model.transformer.embeddings.position_embeddings.shape[1] == 577
    x, y = synth.synth_batch(2, 384, 10, seed=52)
    loss_ref, logits_ref, grads_ref = O.vit_loss_and_grads(params, x, y, dict(patch=16, heads=16, layers=2))
    tol = TOL[dtype]
    rep = Report(tol)
    logits, _ = model(x.to(DEV))
    check(logits, logits_ref, tol, "logits", report=rep)
    loss = model(x.to(DEV), y.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) <= tol * abs(float(loss_ref))
    for k, p in model.named_parameters():
        check(p.grad, grads_ref[k], tol, "grad " + k, report=rep)
    rep.finish(f"L/16 geometry 384px {dtype}")
