"""FusedSGD (two HIP launches through the C-ABI) against the CPU optimizer oracle."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
from oracle import optim_oracle as OO  # noqa: E402

pytestmark = pytest.mark.gpu
# odd sizes and a tensor larger than one 8192-element block; (3,), (10,) take the scalar path
SHAPES = [(768, 768), (3072,), (10,), (197, 768), (3,), (5, 7), (8193,), (4, 2049)]


def _grads(shapes, gen, scale):
    return [torch.randn(s, generator=gen) * scale for s in shapes]


@pytest.mark.parametrize("momentum,wd,max_norm", [(0.9, 0.0, 1.0), (0.9, 1e-3, 1.0), (0.0, 0.0, None),
                                                   (0.5, 0.0, 0.25), (0.9, 0.0, None)])
def test_fused_sgd_matches_oracle(momentum, wd, max_norm):
    from utils.optim import FusedSGD
    gen = torch.Generator().manual_seed(11)
    init = [torch.randn(s, generator=gen) for s in SHAPES]
    ref = [p.clone() for p in init]
    bufs = [None] * len(ref)
    params = [torch.nn.Parameter(p.clone().cuda()) for p in init]
    opt = FusedSGD(params, lr=3e-2, momentum=momentum, weight_decay=wd)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0 / (1 + k))
    for k in range(4):
        grads = _grads(SHAPES, gen, 3.0 if k == 1 else 0.01)      # step 1 is clipped hard, the others are not
        for p, g in zip(params, grads):
            p.grad = g.cuda()
        lr = opt.param_groups[0]["lr"]
        n_ref = OO.sgd_clip_step(ref, grads, bufs, lr, momentum, 0.0, wd, max_norm)
        n = opt.step(max_grad_norm=max_norm)
        sched.step()
        opt.zero_grad()
        if max_norm is not None:
            assert abs(float(n) - float(n_ref)) <= 1e-5 * float(n_ref)
        for a, b in zip(params, ref):
            assert torch.allclose(a.detach().cpu(), b, rtol=2e-6, atol=1e-7), (k, a.shape)


def test_fused_sgd_two_groups_and_missing_grads():
    from utils.optim import FusedSGD
    gen = torch.Generator().manual_seed(12)
    init = [torch.randn(s, generator=gen) for s in SHAPES]
    ref = [p.clone() for p in init]
    params = [torch.nn.Parameter(p.clone().cuda()) for p in init]
    opt = FusedSGD([dict(params=params[:3], lr=1e-2), dict(params=params[3:], lr=5e-2, weight_decay=1e-3)],
                   lr=1.0, momentum=0.9)
    bufs = [None] * len(ref)
    for k in range(3):
        grads = _grads(SHAPES, gen, 0.5)
        skip = 4 if k == 0 else -1                      # a parameter without a gradient is left alone
        for i, (p, g) in enumerate(zip(params, grads)):
            p.grad = None if i == skip else g.cuda()
        idx = [i for i in range(len(ref)) if i != skip]
        total, coef = OO.clip_coef([grads[i] for i in idx], 1.0)
        for i in idx:
            lr, wd = (1e-2, 0.0) if i < 3 else (5e-2, 1e-3)
            one_p, one_b = [ref[i]], [bufs[i]]
            OO.sgd_clip_step(one_p, [grads[i] * coef], one_b, lr, 0.9, 0.0, wd, None)
            bufs[i] = one_b[0]
        n = opt.step(max_grad_norm=1.0)
        assert abs(float(n) - float(total)) <= 1e-5 * float(total)
        for a, b in zip(params, ref):
            assert torch.allclose(a.detach().cpu(), b, rtol=2e-6, atol=1e-7), (k, a.shape)


def test_fused_sgd_rejects_unsupported_tensors():
    from utils.optim import FusedSGD
    p = torch.nn.Parameter(torch.randn(8, 8).cuda().half())
    p.grad = torch.randn(8, 8).cuda().half()
    with pytest.raises(ValueError):
        FusedSGD([p], lr=0.1).step()


def test_gradient_norm_is_bit_reproducible():
    """The clip coefficient scales every update, so replicas that received the same reduced gradients stay identical
    only if the norm of those gradients is the same number on each of them: it is summed in a fixed order (no atomics)
    and must come out bit-identical call after call, also when other kernels ran in between."""
    from utils.optim import FusedSGD
    torch.manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (5_000_003, 1_200_000, 77, 8192, 3_000_001)]
    for p in params:
        p.grad = torch.randn_like(p) * 1e-2
    norms = []
    for rep in range(6):
        opt = FusedSGD(params, lr=0.0, momentum=0.0)      # lr 0: parameters and gradients stay as they are
        norms.append(opt.step(max_grad_norm=1.0).item())
        torch.randn(1 << 20, device="cuda").sum()          # unrelated work between the calls
    assert len(set(norms)) == 1, norms
    ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)).item()
    assert abs(norms[0] - ref) <= 1e-5 * ref


def _tiny_model(seed):
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    from oracle import synth
    vitk.set_compute_dtype(torch.bfloat16)
    model = VisionTransformer(cfgmod.get_tiny_config(), 32, num_classes=10, zero_head=False)
    model.load_state_dict(synth.synth_params("tiny", 32, 10, seed=seed))
    return model.cuda().train()


def _train(model, steps, fused, cache):
    """A few clipped SGD steps; returns the losses.  `cache`: keep the GEMM operands with the parameters."""
    import vitk
    from oracle import synth
    from utils.optim import FusedSGD
    old = vitk.ops._OPERAND_CACHE
    vitk.ops._OPERAND_CACHE = cache
    try:
        opt = (FusedSGD(model.parameters(), lr=3e-2, momentum=0.9) if fused
               else torch.optim.SGD(model.parameters(), lr=3e-2, momentum=0.9))
        losses = []
        for k in range(steps):
            x, y = synth.synth_batch(8, 32, 10, seed=300 + k)
            loss = model(x.cuda(), y.cuda())
            loss.backward()
            if fused:
                opt.step(max_grad_norm=1.0)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
            opt.zero_grad()
            losses.append(float(loss))
        return losses
    finally:
        vitk.ops._OPERAND_CACHE = old


@pytest.mark.parametrize("fused", [True, False], ids=["FusedSGD", "torch.optim.SGD"])
def test_kept_gemm_operands_follow_the_optimizer(fused):
    """The bf16 copies of the weights kept with the parameters (renewed by FusedSGD in one launch, re-derived after
    torch's in-place updates because the version counter moved) give the same bits as casting in every pass."""
    import vitk
    ref = _train(_tiny_model(5), 4, fused, cache=False)
    vitk.ops._OPERAND_VERIFY = True              # every hit is compared with a fresh cast
    before = dict(vitk.ops.operand_stats)
    try:
        got = _train(_tiny_model(5), 4, fused, cache=True)
    finally:
        vitk.ops._OPERAND_VERIFY = False
    after = vitk.ops.operand_stats
    assert got == pytest.approx(ref, rel=2e-6), (got, ref)        # weight gradients are summed with fp32 atomics
    if fused:
        assert after["refresh"] - before["refresh"] == 4
        assert after["hit"] - before["hit"] > 0           # steps 2.. take what the optimizer left behind
    else:
        assert after["refresh"] == before["refresh"]


def test_kept_gemm_operands_invalidation():
    """Eval passes reuse the kept operands; an in-place update is seen through the version counter; a write
    through .data is not (documented) until invalidate_operands() is called."""
    import vitk
    from oracle import synth
    model = _tiny_model(6).eval()
    x, _ = synth.synth_batch(4, 32, 10, seed=1)
    x = x.cuda()
    with torch.no_grad():
        a = model(x)[0].clone()
        h0 = vitk.ops.operand_stats["hit"]
        b = model(x)[0].clone()
        assert vitk.ops.operand_stats["hit"] > h0 and torch.equal(a, b)
        w = model.transformer.encoder.layer[0].ffn.fc1.weight
        w.mul_(1.5)                                   # version counter moves: re-derived by itself
        c = model(x)[0].clone()
        assert not torch.equal(a, c)
        w.data.mul_(1.0 / 1.5)                        # bypasses the counter: the kept copy is stale ...
        vitk.ops._OPERAND_VERIFY = True
        try:
            with pytest.raises(vitk.VitkError):
                model(x)
        finally:
            vitk.ops._OPERAND_VERIFY = False
        vitk.ops.invalidate_operands(model.parameters())      # ... until the caller says so
        d = model(x)[0]
        assert float((d - a).abs().max()) <= 2e-2 * float(a.abs().max())
