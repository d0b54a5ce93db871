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
