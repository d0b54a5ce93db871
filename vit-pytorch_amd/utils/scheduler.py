"""Learning-rate multipliers used by the training drivers (reference utils/scheduler.py:8-63),
expressed as one LambdaLR subclass per policy with the policy as a pure function of the step."""
import math

from torch.optim.lr_scheduler import LambdaLR


def _ramp(step, warmup_steps, floor):
    return float(step) / float(max(floor, warmup_steps))


class _Schedule(LambdaLR):
    def __init__(self, optimizer, last_epoch=-1):
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)


class ConstantLRSchedule(_Schedule):
    def lr_lambda(self, step):
        return 1.0


class WarmupConstantSchedule(_Schedule):
    """0 -> 1 linearly over `warmup_steps`, then 1."""

    def __init__(self, optimizer, warmup_steps, last_epoch=-1):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, last_epoch)

    def lr_lambda(self, step):
        return _ramp(step, self.warmup_steps, 1.0) if step < self.warmup_steps else 1.0


class WarmupLinearSchedule(_Schedule):
    """Linear warmup, then linear decay to 0 at `t_total`."""

    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps, self.t_total = warmup_steps, t_total
        super().__init__(optimizer, last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return _ramp(step, self.warmup_steps, 1)
        return max(0.0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))


class WarmupCosineSchedule(_Schedule):
    """Linear warmup, then `cycles` cosine periods (default half a period: 1 -> 0)."""

    def __init__(self, optimizer, warmup_steps, t_total, cycles=.5, last_epoch=-1):
        self.warmup_steps, self.t_total, self.cycles = warmup_steps, t_total, cycles
        super().__init__(optimizer, last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return _ramp(step, self.warmup_steps, 1.0)
        progress = float(step - self.warmup_steps) / float(max(1, self.t_total - self.warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(self.cycles) * 2.0 * progress)))
