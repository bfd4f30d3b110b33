"""The host-side LR schedules (schedules.py) step by step against the torch classes the reference instantiates
(v2:1031-1037, v2:1248), and the FusedAdam / scheduler state round trip of the resume side-car."""
import torch

from oxford_102_flower_gan_vae_latent_diffusion_amd import schedules


class Opt:
    def __init__(self):
        self.lr, self.b1 = None, 0.9

    def set_lr(self, v):
        self.lr = v

    def set_beta1(self, v):
        self.b1 = v


def test_onecycle_equals_torch():
    for total in (2, 7, 300 * 16):
        p = torch.nn.Parameter(torch.zeros(1))
        topt = torch.optim.AdamW([p], lr=1e-4, betas=(0.9, 0.999))
        tsch = torch.optim.lr_scheduler.OneCycleLR(topt, max_lr=1e-4, total_steps=total, pct_start=0.3, div_factor=25,
                                                   final_div_factor=1000)
        o = Opt()
        sch = schedules.OneCycle(o, 1e-4, total_steps=total)
        for i in range(total):
            g = topt.param_groups[0]
            assert abs(o.lr - g["lr"]) <= 1e-18 + 1e-12 * g["lr"], (total, i, o.lr, g["lr"])
            assert abs(o.b1 - g["betas"][0]) <= 1e-12, (total, i)
            topt.step()
            if i + 1 < total:
                tsch.step()
                sch.step()


def test_cosine_warm_restarts_equals_torch_and_resumes():
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=1e-3)
    tsch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(topt, T_0=10, T_mult=2)
    o = Opt()
    sch = schedules.CosineWarmRestarts(o, 1e-3)
    saved = None
    for e in range(75):
        assert abs(o.lr - topt.param_groups[0]["lr"]) <= 1e-15, e
        if e == 33:
            saved = (sch.state_dict(), o.lr)
        topt.step()
        tsch.step()
        sch.step()
    o2 = Opt()
    sch2 = schedules.CosineWarmRestarts(o2, 1e-3)
    sch2.load_state_dict(saved[0])
    assert o2.lr == saved[1]
    for _ in range(75 - 33):
        sch2.step()
    assert o2.lr == o.lr
