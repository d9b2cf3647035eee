"""Drop-in for protlig_dd/utils/lr_scheduler.py:51-87 (`WarmupCosineLR`): linear warm-up from base_lr to max_lr over
`warmup_steps`, then cosine decay to min_lr at `max_steps` (the cosine keeps running past max_steps, as in the
reference).  Host glue: O(1) scalar arithmetic per step.  `warmup_cosine_lr` is the same rule as a pure function of
the step, which is what `SEDDTrainer(lr_schedule=...)` consumes (the trainer owns no torch optimizer: its AdamW is the
fused kernel sedd_adamw_ema_step)."""
import math

from torch.optim.lr_scheduler import _LRScheduler


def warmup_cosine_lr(step, warmup_steps, max_steps, base_lr=1e-5, max_lr=1e-3, min_lr=1e-6):
    if step < warmup_steps:
        scale = step / float(max(1, warmup_steps))
        return base_lr + scale * (max_lr - base_lr)
    progress = (step - warmup_steps) / float(max(1, max_steps - warmup_steps))
    cosine = 0.5 * (1.0 + math.cos(math.pi * progress))
    return min_lr + (max_lr - min_lr) * cosine


class WarmupCosineLR(_LRScheduler):
    """Same constructor as the reference (lr_scheduler.py:64-73); wraps any torch optimizer."""

    def __init__(self, optimizer, warmup_steps, max_steps, base_lr=1e-5, max_lr=1e-3, min_lr=1e-6, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.max_steps = max_steps
        self.base_lr = base_lr
        self.max_lr = max_lr
        self.min_lr = min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        lr = warmup_cosine_lr(self.last_epoch, self.warmup_steps, self.max_steps, self.base_lr, self.max_lr, self.min_lr)
        return [lr for _ in self.base_lrs]

    def schedule(self):
        """step -> lr callable for SEDDTrainer(lr_schedule=...)."""
        return lambda step: warmup_cosine_lr(step, self.warmup_steps, self.max_steps, self.base_lr, self.max_lr, self.min_lr)
