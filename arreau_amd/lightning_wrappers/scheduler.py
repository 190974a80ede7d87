"""Learning-rate schedule of the training preset: half a cosine period over `max_iters` epochs, scaled by a linear ramp
while `epoch <= warmup`.  Same constructor and `get_lr_factor` as the reference's class of this name
(/root/reference/lightning_wrappers/scheduler.py:5-19; stepped once per epoch, Lightning's default), built on LambdaLR;
the factors are pinned against values produced by the reference class (tests/golden/scheduler.json)."""
import math

from torch.optim.lr_scheduler import LambdaLR

_RAMP_EPS = 1e-6  # the reference's guard against warmup = 0 (it also makes the very first factor 1e-6 / warmup, not 0)


def cosine_warmup_factor(epoch: int, warmup: int, max_iters: int) -> float:
    cosine = 0.5 + 0.5 * math.cos(math.pi * epoch / max_iters)
    ramp = (epoch + _RAMP_EPS) / (warmup + _RAMP_EPS) if epoch <= warmup else 1.0
    return cosine * ramp


class CosineWarmupScheduler(LambdaLR):
    def __init__(self, optimizer, warmup, max_iters):
        self.warmup, self.max_num_iters = warmup, max_iters
        super().__init__(optimizer, lr_lambda=self.get_lr_factor)

    def get_lr_factor(self, epoch):
        return cosine_warmup_factor(epoch, self.warmup, self.max_num_iters)

    def state_dict(self):  # (LambdaLR cannot pickle a bound method; the schedule is a pure function of last_epoch)
        return {k: v for k, v in self.__dict__.items() if k not in ("optimizer", "lr_lambdas")}

    def load_state_dict(self, state):
        self.__dict__.update({k: v for k, v in state.items() if k != "lr_lambdas"})
