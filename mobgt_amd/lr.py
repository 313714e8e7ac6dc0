"""Learning-rate schedule of the reference (`graphormer/lr.py:7-34`): linear warm-up to `lr` over
`warmup_updates` scheduler steps, then a polynomial fall to `end_lr` at `tot_updates`, flat afterwards.

The schedule is ONE closed-form function of the step count, `polynomial_decay_lr`; the three consumers share it:
`PolynomialDecayLR` (the torch scheduler `configure_optimizers` returns, same constructor as the reference's),
`train.TrainStep` (host mirror of the value) and `adamw_flat_kernel` (csrc/layer.hip evaluates the same formula
from the device step counter for power == 1).
"""
from torch.optim.lr_scheduler import LRScheduler


def polynomial_decay_lr(step_count, warmup_updates, tot_updates, lr, end_lr, power=1.0):
    """Value of the schedule at scheduler step `step_count` (torch's `_step_count`: 1 right after construction)."""
    if step_count <= warmup_updates:
        return lr * (step_count / float(warmup_updates))
    if step_count >= tot_updates:
        return end_lr
    left = (tot_updates - step_count) / float(tot_updates - warmup_updates)      # 1 at the end of warm-up, 0 at tot_updates
    return end_lr + (lr - end_lr) * left ** power


class PolynomialDecayLR(LRScheduler):
    """Every param group follows `polynomial_decay_lr(_step_count, ...)`.  `verbose` is accepted for signature parity
    and ignored (torch >= 2.7 dropped it)."""

    def __init__(self, optimizer, warmup_updates, tot_updates, lr, end_lr, power, last_epoch=-1, verbose=False):
        self.schedule = dict(warmup_updates=warmup_updates, tot_updates=tot_updates, lr=lr, end_lr=end_lr, power=power)
        super().__init__(optimizer, last_epoch)

    def __getattr__(self, name):                 # reference attribute names (warmup_updates, tot_updates, lr, end_lr, power)
        sched = self.__dict__.get("schedule")
        if sched is not None and name in sched:
            return sched[name]
        raise AttributeError(name)

    def get_lr(self):
        return [polynomial_decay_lr(self._step_count, **self.schedule)] * len(self.optimizer.param_groups)

    def load_state_dict(self, state_dict):
        """Also accepts a state written by the REFERENCE scheduler (graphormer/lr.py:9-15 keeps warmup_updates, tot_updates,
        lr, end_lr, power as plain attributes, so its state_dict has them at top level): they are folded into `schedule`,
        which is what get_lr reads."""
        state_dict = dict(state_dict)
        sched = dict(state_dict.pop("schedule", self.schedule))
        for k in ("warmup_updates", "tot_updates", "lr", "end_lr", "power"):
            if k in state_dict:
                sched[k] = state_dict.pop(k)
        super().load_state_dict(state_dict)
        self.schedule = sched

