"""Host-side learning-rate schedules of the two reference loops, driving FusedAdam.set_lr / set_beta1 (device scalars
are updated between steps, outside any captured graph).

OneCycle             = torch.optim.lr_scheduler.OneCycleLR as v2:1031-1037 configures it (cosine annealing, two phases,
                       pct_start 0.3, div_factor 25, final_div_factor 1000) stepped once per batch (v2:1130). Like torch's,
                       it also cycles AdamW's beta1 between 0.95 and 0.85 (cycle_momentum defaults to True).
CosineWarmRestarts   = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(T_0=10, T_mult=2) stepped once per epoch
                       (v2:1248, v2:1269).
Both are pinned step by step against the torch classes in tests/test_schedules_cpu.py; their position is part of the
resume side-car (checkpoint.save_train_state)."""
import math


def _cos(start, end, pct):
    return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)


class OneCycle:
    def __init__(self, opt, max_lr, total_steps, pct_start=0.3, div_factor=25.0, final_div_factor=1000.0,
                 base_momentum=0.85, max_momentum=0.95, cycle_momentum=True):
        self.opt, self.total_steps = opt, int(total_steps)
        self.initial_lr = max_lr / div_factor
        self.max_lr = max_lr
        self.min_lr = self.initial_lr / final_div_factor
        self.base_m, self.max_m, self.cycle_momentum = base_momentum, max_momentum, cycle_momentum
        self.phase_ends = (float(pct_start * self.total_steps) - 1.0, float(self.total_steps) - 1.0)
        self.step_num = 0
        self.apply()

    def values(self, step_num):
        """(lr, beta1) torch's OneCycleLR has set after `step_num` calls of scheduler.step()."""
        if step_num > self.total_steps:
            raise ValueError("OneCycle stepped %d times, total_steps is %d" % (step_num, self.total_steps))
        start = 0.0
        spec = ((self.initial_lr, self.max_lr, self.max_m, self.base_m), (self.max_lr, self.min_lr, self.base_m, self.max_m))
        for i, (end, (lr0, lr1, m0, m1)) in enumerate(zip(self.phase_ends, spec)):
            if step_num <= end or i == len(spec) - 1:
                pct = (step_num - start) / (end - start)
                return _cos(lr0, lr1, pct), _cos(m0, m1, pct)
            start = end

    def apply(self):
        lr, m = self.values(self.step_num)
        self.opt.set_lr(lr)
        if self.cycle_momentum:
            self.opt.set_beta1(m)
        return lr, m

    def step(self):
        """after optimizer.step() of every batch (v2:1130)."""
        self.step_num += 1
        return self.apply()

    def state_dict(self):
        return {"kind": "onecycle", "step_num": self.step_num}

    def load_state_dict(self, st):
        self.step_num = int(st["step_num"])
        self.apply()


class CosineWarmRestarts:
    def __init__(self, opt, base_lr, T_0=10, T_mult=2, eta_min=0.0):
        self.opt, self.base_lr, self.T_0, self.T_mult, self.eta_min = opt, base_lr, int(T_0), int(T_mult), eta_min
        self.T_i, self.T_cur, self.epoch = self.T_0, 0, 0
        self.apply()

    def value(self):
        return self.eta_min + (self.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * self.T_cur / self.T_i)) / 2.0

    def apply(self):
        lr = self.value()
        self.opt.set_lr(lr)
        return lr

    def step(self):
        """once per epoch (v2:1269)."""
        self.epoch += 1
        self.T_cur += 1
        if self.T_cur >= self.T_i:
            self.T_cur -= self.T_i
            self.T_i *= self.T_mult
        return self.apply()

    def state_dict(self):
        return {"kind": "cosine_warm_restarts", "T_i": self.T_i, "T_cur": self.T_cur, "epoch": self.epoch}

    def load_state_dict(self, st):
        self.T_i, self.T_cur, self.epoch = int(st["T_i"]), int(st["T_cur"]), int(st["epoch"])
        self.apply()
