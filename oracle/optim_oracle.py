"""CPU restatement of the optimizer step of the reference loop.  TEST INFRASTRUCTURE ONLY: imported
by tests/ (and nothing else); the product path is vit-pytorch_amd/utils/optim.py on the HIP library.

Follows /root/reference/train.py:535-538 (clip_grad_norm_(parameters, max_grad_norm), SGD.step(),
zero_grad()) with the optimizer built at train.py:475-478 (torch.optim.SGD, momentum 0.9): plain
tensor arithmetic, one parameter at a time.  Pinned by tests/test_optim_cpu.py against
torch.nn.utils.clip_grad_norm_ + torch.optim.SGD themselves.
"""
import torch


def clip_coef(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: joint 2-norm of all gradients, coefficient clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def sgd_clip_step(params, grads, bufs, lr, momentum=0.9, dampening=0.0, weight_decay=0.0, max_norm=None):
    """In place on `params` and `bufs` (a list holding None before the first step).  Returns the norm."""
    total = None
    coef = 1.0
    if max_norm is not None:
        total, coef = clip_coef(grads, max_norm)
    for i, (p, g) in enumerate(zip(params, grads)):
        g = g * coef
        if weight_decay:
            g = g + weight_decay * p
        if momentum:
            bufs[i] = g.clone() if bufs[i] is None else momentum * bufs[i] + (1.0 - dampening) * g
            g = bufs[i]
        p -= lr * g
    return total
