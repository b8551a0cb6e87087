"""Learning-rate schedule of the reference's training recipe.  Plugin-compatible with
``captioning.utils.lr_scheduler.ExponentialDecayScheduler`` (lr_scheduler.py:5-42): linear warm-up to the base rate
over ``warmup_iters`` scheduler steps, then a geometric decay that reaches ``final_lrs`` at ``total_iters``; stepped
once per iteration before the optimiser (run.py:104-105).  Host-side arithmetic only.
"""
import torch


class ExponentialDecayScheduler(torch.optim.lr_scheduler._LRScheduler):

    def __init__(self, optimizer, total_iters, final_lrs, warmup_iters=3000, last_epoch=-1, verbose=False):
        self.total_iters = total_iters
        n = len(optimizer.param_groups)
        self.final_lrs = list(final_lrs) if isinstance(final_lrs, (list, tuple)) else [final_lrs] * n
        self.warmup_iters = warmup_iters
        self.bases = [0.0] * n
        super().__init__(optimizer, last_epoch)
        span = self.total_iters - self.warmup_iters
        self.bases = [(final / base) ** (1.0 / span) for base, final in zip(self.base_lrs, self.final_lrs)]

    def get_lr(self):
        it = self._step_count
        if it <= self.warmup_iters:
            coeff = it / self.warmup_iters if it < self.warmup_iters else 1.0
            return [coeff * base for base in self.base_lrs]
        return [base * decay ** (it - self.warmup_iters) for base, decay in zip(self.base_lrs, self.bases)]
