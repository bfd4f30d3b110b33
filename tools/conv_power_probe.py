"""Clock / power while the dominant conv kernel runs back to back (rocm-smi sampled from a side thread), against an idle
sample and a register-only MFMA loop of similar length: is the f16x2 conv running at the data-sheet clock?
Usage: python tools/conv_power_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops  # noqa: E402


def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in r.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "Socket"))]
    return " | ".join(keep)


def sample_during(fn, seconds, label):
    stop = False
    out = []

    def sampler():
        while not stop:
            time.sleep(0.6)
            out.append(smi())

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    print("== %s: %d launches, %.4f ms each" % (label, n, 1e3 * dt / n))
    for s in out[1:6]:
        print("   ", s)


def main():
    dev = torch.device("cuda:0")
    print("idle:", smi())
    B = 256
    for (C, H) in ((512, 8), (128, 32)):
        x = torch.randn(B, H, H, C, device=dev)
        x = x * torch.sigmoid(x)
        w = torch.randn(C, 3, 3, C, device=dev) * (2.0 / (9 * C)) ** 0.5
        wh2 = ops.pack_weight_f16x2(w)
        bias = torch.zeros(C, device=dev)
        amax = x.abs().max().reshape(1)
        y = torch.empty(B, H, H, C, device=dev)
        sample_during(lambda: ops.conv2d_fwd_f16x2(x, wh2, tuple(w.shape), bias, 1, 1, x_amax=amax, out=y), 4.0,
                      "conv3x3_h2 C=%d W=%d" % (C, H))
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    sample_during(lambda: torch.mm(a, b), 4.0, "torch.mm bf16 8192^3 (library GEMM, for the clock comparison only)")


if __name__ == "__main__":
    main()
