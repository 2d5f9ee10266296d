"""WarmupCosineLR (pc_processor/utils/warmup_lr.py:54-97) in closed form.

The reference chains torch's CyclicLR (0 -> lr over `warmup_steps`) and CosineAnnealingLR(T_max=max_steps); because
the wrapper's constructor already takes one step, after k calls of ``step()`` the learning rate is
    m = k + 1;   lr * m / warmup            if m <= warmup
                 lr * (1 + cos(pi * (m - warmup) / max_steps)) / 2   otherwise
(pinned against the reference's 15-step trace in tests/golden/g6_losses.npz)."""
import math


class WarmupCosineLR:
    def __init__(self, optimizer, lr, warmup_steps, momentum, max_steps):
        self.optimizer, self.lr, self.momentum = optimizer, lr, momentum
        self.warmup_steps = max(int(warmup_steps), 1)
        self.max_steps = max(int(max_steps), 1)
        self.k = 0
        self._apply()

    def lr_at(self, k):
        m = k + 1
        if m <= self.warmup_steps:
            return self.lr * m / self.warmup_steps
        return self.lr * (1.0 + math.cos(math.pi * (m - self.warmup_steps) / self.max_steps)) / 2.0

    def _apply(self):
        v = self.lr_at(self.k)
        for g in self.optimizer.param_groups:
            g["lr"] = v

    def step(self, epoch=None):
        self.k = self.k + 1 if epoch is None else int(epoch)
        self._apply()

    def state_dict(self):
        return {"k": self.k}

    def load_state_dict(self, sd):
        self.k = int(sd["k"])
        self._apply()
