"""Wall times of the persistent programs (hipGraph replays, batch 256 / 512): folded and unfolded ConditionalUNet forward, the
denoiser's training program + optimizer program, one sampler step. Run twice to A/B the executor's modes:
    python tools/program_ab.py            # XCD-local phases (default)
    FL_PROG_LOCAL=0 python tools/program_ab.py   # every phase device-wide
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, program as P, steps  # noqa: E402
from oxford_102_flower_gan_vae_latent_diffusion_amd.program_train import DenoiserTrainProgram, OptimizerProgram  # noqa: E402

dev = torch.device("cuda:0")


def wall(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def graphed(fn, iters=400):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return wall(g.replay, iters) * 1e3


def main():
    torch.manual_seed(0)
    res = {"FL_PROG_LOCAL": os.environ.get("FL_PROG_LOCAL", "1")}
    B = 256
    un = M.ConditionalUNet().to(dev).eval()
    x = torch.randn(B, 256, device=dev)
    t = torch.randint(0, 1000, (B,), device=dev)
    c = torch.randint(0, 102, (B,), device=dev)
    with torch.no_grad():
        fp = P.FoldedUNetProgram(un, B, dev)
        out = torch.empty(B, 256, device=dev)
        res["unet_fwd_folded_ms"] = graphed(lambda: fp(x, t, c, out=out))
        fp.prog.check()
        res["folded_phases"] = (fp.prog.n_phases, fp.prog.n_local_phases)
        up = P.UNetProgram(un, B, dev)
        res["unet_fwd_unfolded_ms"] = graphed(lambda: up(x, t, c, out=out))
        up.prog.check()
    # training programs
    un.train()
    fpm = steps.FlatParams(un)
    tp = DenoiserTrainProgram(un, fpm.gviews, B, dev)
    eps = torch.randn(B, 256, device=dev)
    bm, hm = un.dropout_masks(B, dev)
    res["train_program_ms"] = graphed(lambda: tp.run(x, t, c, eps, bm, hm), iters=200)
    tp.prog.check()
    res["train_phases"] = (tp.prog.n_phases, tp.prog.n_local_phases)
    res["train_loss"] = float(tp.loss)
    opt = steps.FusedAdam(fpm, lr=1e-4, weight_decay=1e-2, decoupled=True, max_norm=1.0)
    op = OptimizerProgram(opt, fpm.n_active)
    res["optimizer_program_ms"] = graphed(op.run, iters=200)
    op.prog.check()
    # one sampler step at batch 512
    un.eval()
    d = M.ConditionalDenoiseDiffusion(un, n_steps=1000, device=dev)
    Bs = 512
    cs = (torch.arange(Bs, device=dev) % 102).long()
    with torch.no_grad():
        d.sample((Bs, 256), dev, cs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.sample((Bs, 256), dev, cs)
        torch.cuda.synchronize()
        res["sample_1000_steps_b512_s"] = time.perf_counter() - t0
    print(json.dumps(res))


if __name__ == "__main__":
    main()
