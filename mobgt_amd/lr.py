"""`graphormer/lr.py:7-34` -- PolynomialDecayLR (linear warm-up, then polynomial decay to `end_lr`).
Same constructor; `verbose` is accepted for signature parity and ignored (torch >= 2.7 dropped it)."""
from torch.optim.lr_scheduler import LRScheduler


class PolynomialDecayLR(LRScheduler):
    def __init__(self, optimizer, warmup_updates, tot_updates, lr, end_lr, power, last_epoch=-1, verbose=False):
        self.warmup_updates = warmup_updates
        self.tot_updates = tot_updates
        self.lr = lr
        self.end_lr = end_lr
        self.power = power
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self._step_count <= self.warmup_updates:
            self.warmup_factor = self._step_count / float(self.warmup_updates)
            lr = self.warmup_factor * self.lr
        elif self._step_count >= self.tot_updates:
            lr = self.end_lr
        else:
            warmup = self.warmup_updates
            lr_range = self.lr - self.end_lr
            pct_remaining = 1 - (self._step_count - warmup) / (self.tot_updates - warmup)
            lr = lr_range * pct_remaining ** self.power + self.end_lr
        return [lr for _ in self.optimizer.param_groups]

    def _get_closed_form_lr(self):
        assert False
