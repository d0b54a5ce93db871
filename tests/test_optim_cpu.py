"""The optimizer oracle against torch's own clip_grad_norm_ + SGD (what reference train.py:535-538 calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import optim_oracle as OO  # noqa: E402

SHAPES = [(768, 768), (3072,), (10,), (197, 768), (3,), (5, 7)]


def _run_torch(steps, momentum, wd, max_norm, seed):
    g = torch.Generator().manual_seed(seed)
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in SHAPES]
    opt = torch.optim.SGD(params, lr=3e-2, momentum=momentum, weight_decay=wd)
    norms = []
    for k in range(steps):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g) * (3.0 if k == 1 else 0.01)
        if max_norm is not None:
            norms.append(torch.nn.utils.clip_grad_norm_(params, max_norm))
        opt.step()
        opt.zero_grad()
    return [p.detach() for p in params], norms


def _run_oracle(steps, momentum, wd, max_norm, seed):
    g = torch.Generator().manual_seed(seed)
    params = [torch.randn(s, generator=g) for s in SHAPES]
    bufs = [None] * len(params)
    norms = []
    for k in range(steps):
        grads = [torch.randn(p.shape, generator=g) * (3.0 if k == 1 else 0.01) for p in params]
        norms.append(OO.sgd_clip_step(params, grads, bufs, 3e-2, momentum, 0.0, wd, max_norm))
    return params, norms


def test_oracle_matches_torch_sgd_with_clipping():
    for momentum, wd, max_norm in [(0.9, 0.0, 1.0), (0.9, 1e-3, 1.0), (0.0, 0.0, None), (0.5, 0.0, None)]:
        pt, nt = _run_torch(4, momentum, wd, max_norm, 5)
        po, no = _run_oracle(4, momentum, wd, max_norm, 5)
        for a, b in zip(pt, po):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
        for a, b in zip(nt, no):
            assert abs(float(a) - float(b)) <= 1e-5 * float(a)
