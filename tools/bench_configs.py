"""Secondary measurements for the other BASELINE.json configs (bench.py stays the single headline metric):
   unet_fwd  : ConditionalUNet forward at batch 256 (north-star 60%-of-MFMA-roofline target, 4.161 GFLOP of record)
   vaegan    : config 2, VAE-GAN full-loss step, batch 64 (22.75 GFLOP/img of record)
   sample    : config 5, 1000-step class-conditional sampling + VAE decode, batch 512 (15.65 GFLOP/img of record)
Prints one JSON object per config."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, steps  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 157.3


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def unet_fwd():
    torch.manual_seed(0)
    un = M.ConditionalUNet().to(dev).eval()
    B = 256
    x = torch.empty(B, 256, device=dev); ops.philox_fill(x, 1)
    t = torch.empty(B, dtype=torch.int64, device=dev); ops.philox_randint(t, 2, 1000)
    c = torch.empty(B, dtype=torch.int64, device=dev); ops.philox_randint(c, 3, 102)
    with torch.no_grad():
        from oxford_102_flower_gan_vae_latent_diffusion_amd import engine
        mk = lambda: engine.unet_forward(un, x, t, c)[0]  # the multi-kernel path (71 launches)
        eager = timeit(mk, 50)
        g = torch.cuda.CUDAGraph()
        mk()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = mk()
        graph = timeit(g.replay, 200)
        # persistent phase program: ONE launch for the whole forward (program.py / csrc/program.hip)
        from oxford_102_flower_gan_vae_latent_diffusion_amd import program as P
        up = P.UNetProgram(un, B, dev)
        o2 = up(x, t, c)
        torch.cuda.synchronize()
        up.prog.check()
        err = (o2 - out).abs().max().item()
        pers_eager = timeit(lambda: up(x, t, c, out=o2), 200)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            up(x, t, c, out=o2)
        pers = timeit(g2.replay, 500)
        up.prog.check()
    return dict(config="unet_fwd_b256", ms_eager=eager * 1e3, ms_graph=graph * 1e3, ms_persistent=pers * 1e3,
                ms_persistent_eager=pers_eager * 1e3, persistent_vs_multikernel_maxabs=err, gflop=4.161,
                tflops=4.161 / pers / 1e3, frac_fp32_mfma=4.161 / pers / 1e3 / PEAK,
                frac_fp32_mfma_multikernel=4.161 / graph / 1e3 / PEAK)


def vaegan():
    torch.manual_seed(0)
    ae = M.SimpleAutoencoder().to(dev).train()
    D = M.Discriminator64().to(dev).train()
    vgg = M.VGGPerceptualLoss(dev)  # seeded-synthetic weights (no network for the pretrained ones)
    tr = steps.VAEGANTrainer(ae, D, vgg, lr=1e-4)
    B = 64
    data = torch.empty(B, 3, 64, 64, device=dev); ops.philox_fill(data, 5, mode=2)
    labels = torch.empty(B, dtype=torch.int64, device=dev); ops.philox_randint(labels, 6, 102)
    eager = timeit(lambda: tr.step(data, labels, epoch=200, num_epochs=300), 10, warm=3)
    tr.capture(data, labels, epoch=200, num_epochs=300)
    dt = timeit(tr.replay, 20, warm=3)
    return dict(config="vaegan_step_b64_epoch200", ms_per_step_eager=eager * 1e3, ms_per_step=dt * 1e3, images_per_s=B / dt, gflop_per_img=22.75,
                tflops=22.75 * B / dt / 1e3, frac_fp32_mfma=22.75 * B / dt / 1e3 / PEAK, loss=float(tr.out["total"]))


def sample():
    torch.manual_seed(0)
    ae = M.SimpleAutoencoder().to(dev).eval()
    un = M.ConditionalUNet().to(dev).eval()
    d = M.ConditionalDenoiseDiffusion(un, n_steps=1000, device=dev)
    B = 512
    c = (torch.arange(B, device=dev) % 102).long()
    with torch.no_grad():
        d.sample((B, 256), dev, c)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = d.sample((B, 256), dev, c)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        img = ae.decode(z)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    assert img.shape == (B, 3, 64, 64) and torch.isfinite(img).all()
    tot = t2 - t0
    return dict(config="sample_1000steps_decode_b512", s_sampling=t1 - t0, s_decode=t2 - t1, images_per_s=B / tot,
                ms_per_reverse_step=(t1 - t0), gflop_per_img=15.65, tflops=15.65 * B / tot / 1e3)


if __name__ == "__main__":
    which = sys.argv[1:] or ["unet_fwd", "vaegan", "sample"]
    for w in which:
        print(json.dumps(globals()[w]()), flush=True)
