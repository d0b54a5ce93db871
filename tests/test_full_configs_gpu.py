"""The two full-size configurations that the benchmark runs besides ViT-B/16, checked numerically at full depth
and width against the CPU oracle on the same seeded weights and batch:

* ViT-L/16 at 384 px: all 24 layers, 577 tokens, hidden 1024, 16 heads, MLP 4096 (preset of reference
  models/ViT/_configs.py:242-255), batch 1;
* R50 + ViT-B/16 hybrid at 224 px: ResNetV2 stem with (3, 4, 9) units at full width feeding the 12-layer
  transformer (reference models/ViT/_configs.py:224-232, modeling.py:130-158, modeling_resnet.py:129-164), batch 2.

fp32 kernels: 1e-5 (plain) / 2e-5 (hybrid: thirteen + thirty-six convolution layers deep) max-norm per tensor.
bf16 kernels: the 2e-2 contract for ViT-L/16; the hybrid at the bounds of tests/test_hybrid_gpu.py (its stem
rounds the operands of 49 convolutions to bf16, which moves the reference arithmetic itself by several per cent).
Stem parameter gradients are compared L2-relative per tensor in the fp32 instance as well: with random weights a
few of the ~10^7 ReLU inputs of a pass lie within fp32 rounding of zero, and whichever way an implementation
rounds them decides single entries of the sums upstream (see tests/test_hybrid_cpu.py)."""
import numpy as np
import pytest
import torch

from oracle import synth, vit_oracle as O
from test_model_gpu import DEV, TOL, Report, check

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_dtype():
    import vitk
    yield
    vitk.set_compute_dtype(torch.bfloat16)


@pytest.fixture(scope="module")
def l16_case():
    params = synth.synth_params("L_16", 384, 10, seed=151)
    x, y = synth.synth_batch(1, 384, 10, seed=152)
    loss, logits, grads = O.vit_loss_and_grads(params, x, y, dict(patch=16, heads=16, layers=24))
    return params, x, y, loss, logits, grads


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vit_l16_384px_all_24_layers_against_oracle(l16_case, dtype):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    params, x, y, loss_ref, logits_ref, grads_ref = l16_case
    vitk.set_compute_dtype(dtype)
    cfg = cfgmod.get_config("ViT-L_16")
    assert (cfg.hidden_size, cfg.transformer.num_heads, cfg.transformer.mlp_dim, cfg.transformer.num_layers) == (1024, 16, 4096, 24)
    cfg.transformer.dropout_rate = 0.0
    model = VisionTransformer(cfg, 384, num_classes=10, zero_head=False)
    model.load_state_dict(params, strict=True)
    model = model.to(DEV).train()
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for v in params.values())
    tol = TOL[dtype]
    rep = Report(tol)
    logits, _ = model(x.to(DEV))
    check(logits, logits_ref, tol, "logits", report=rep)
    loss = model(x.to(DEV), y.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) <= tol * abs(float(loss_ref))
    for k, p in model.named_parameters():
        check(p.grad, grads_ref[k], tol, "grad " + k, report=rep)
    rep.finish(f"ViT-L/16 384px, 24 layers, {dtype}")


R50 = dict(block_units=(3, 4, 9), width_factor=1, img=224, batch=2, ncls=10, seed=161, batch_seed=162)
_QK = (".attn.query.", ".attn.key.")


@pytest.fixture(scope="module")
def r50_case():
    """fp32 oracle, and the same oracle evaluated in fp64: with random weights this 49-convolution stem is badly
    conditioned in backward (GroupNorm's backward subtracts the two group projections from gradients that lie
    almost entirely inside them, unit after unit), so two correct fp32 evaluations differ by ~1.6e-2 (L2) on every
    stem parameter gradient, and by ~7e-5 on the query / key gradients (nearly uniform attention).  The fp64
    run is the comparison target for those; `cond` is the distance of the fp32 oracle from it per tensor."""
    params = synth.synth_hybrid_params("B_16", R50["img"], R50["ncls"], R50["block_units"], R50["width_factor"], seed=R50["seed"])
    x, y = synth.synth_batch(R50["batch"], R50["img"], R50["ncls"], seed=R50["batch_seed"])
    cfg = synth.oracle_cfg("B_16")
    loss, logits, grads = O.vit_loss_and_grads(params, x, y, cfg)
    _, _, g64 = O.vit_loss_and_grads({k: v.double() for k, v in params.items()}, x.double(), y, cfg)
    cond = {k: float((grads[k].double() - g64[k]).norm() / g64[k].norm().clamp_min(1e-300)) for k in grads}
    return params, x, y, loss, logits, grads, g64, cond


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_r50_vit_b16_full_width_224px_against_oracle(r50_case, dtype):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    params, x, y, loss_ref, logits_ref, grads_ref, g64, cond = r50_case
    vitk.set_compute_dtype(dtype)
    cfg = cfgmod.get_config("R50-ViT-B_16")
    assert tuple(cfg.resnet.num_layers) == R50["block_units"] and cfg.resnet.width_factor == 1
    cfg.transformer.dropout_rate = 0.0
    model = VisionTransformer(cfg, R50["img"], num_classes=R50["ncls"], zero_head=False)
    model.load_state_dict(params, strict=True)
    model = model.to(DEV).train()
    fp32 = dtype == torch.float32
    tol = 2e-5 if fp32 else 1e-1
    rep = Report(tol)
    logits, _ = model(x.to(DEV))
    check(logits, logits_ref, tol, "logits", report=rep)
    loss = model(x.to(DEV), y.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) <= tol * abs(float(loss_ref))
    hard = []             # badly conditioned tensors: L2-relative distance from the fp64 oracle, next to the fp32 oracle's own
    for k, p in model.named_parameters():
        if ".hybrid_model." in k or any(t in k for t in _QK) or cond[k] > 1e-5:
            d = float((p.grad.double().cpu() - g64[k]).norm() / g64[k].norm().clamp_min(1e-300))
            hard.append((d, cond[k], k))
        else:
            check(p.grad, grads_ref[k], tol, "grad " + k, report=rep)
    rep.finish(f"R50+ViT-B/16 224px, well-conditioned tensors, {dtype}")
    hard.sort(reverse=True)
    stem = [t for t in hard if ".hybrid_model." in t[2]]
    qk = [t for t in hard if any(q in t[2] for q in _QK) and not t[2].endswith("key.bias")]
    other = [t for t in hard if t not in stem and t not in qk and not t[2].endswith("key.bias")]
    for name, rows in (("stem", stem), ("query/key", qk), ("other tensors the fp32 oracle holds to worse than 1e-5", other)):
        print(f"{name} gradients vs the fp64 oracle, L2-relative, {dtype}: worst",
              ", ".join(f"{k.split('hybrid_model.')[-1].replace('transformer.encoder.', '')}={d:.2e} (fp32 oracle {c:.2e})"
                        for d, c, k in rows[:4]))
    if fp32:
        # an fp32 implementation may be as far from fp64 as the fp32 oracle is, times a margin of 3
        bad = [(d, c, k) for d, c, k in stem + qk + other if d > 3.0 * max(c, 1e-5)]
        assert not bad, bad[:4]
    # bf16: rounding the stem's operands is amplified the same way (1e-7 -> 1.6e-2 in fp32 means 4e-3 -> noise):
    # its stem and query/key gradients, and the few others that two fp32 evaluations already hold to worse than
    # 1e-5 (position embeddings and class token: 2.6e-5), are printed, not judged; tests/test_hybrid_gpu.py pins
    # the bf16 stem on a geometry where bf16 is meaningful
