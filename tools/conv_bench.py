"""Micro-benchmark of the dominant kernel (implicit-GEMM conv) on the 9 encoder shapes at B=256.
Usage: python tools/conv_bench.py [reps]   (used under rocprofv3 for PMC / kernel-trace runs)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import conv_shapes  # noqa: E402
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(os.environ.get("B", "256"))
    dev = torch.device("cuda:0")
    tot_t = tot_f = 0.0
    # sustained warm-up so clocks / caches are in their steady state before anything is timed
    xw = torch.empty(256, 32, 32, 128, device=dev); ops.philox_fill(xw, seed=1, mode=0)
    ww = torch.empty(128, 3, 3, 128, device=dev); ops.philox_fill(ww, seed=2, mode=0)
    yw = torch.empty(256, 32, 32, 128, device=dev)
    for _ in range(int(os.environ.get("WARM", "300"))):
        ops.conv2d_fwd(xw, ww, None, 1, 1, out=yw)
    torch.cuda.synchronize()
    for name, b, H, W, Ci, Co, k, s, p in conv_shapes(B):
        x = torch.empty(b, H, W, Ci, device=dev)
        ops.philox_fill(x, seed=11, mode=0)
        w = torch.empty(Co, k, k, Ci, device=dev)
        ops.philox_fill(w, seed=12, mode=0)
        bias = torch.zeros(Co, device=dev)
        Ho = (H + 2 * p - k) // s + 1
        y = torch.empty(b, Ho, Ho, Co, device=dev)
        if os.environ.get("PRESPLIT"):  # weights split once (B_FRAGS): the inference-encoder path
            wsp, geom = ops.split_pack_weight(w), tuple(w.shape)
            run = lambda: ops.conv2d_fwd_presplit(x, wsp, geom, bias, s, p, out=y)
        else:
            run = lambda: ops.conv2d_fwd(x, w, bias, s, p, out=y)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * b * Ho * Ho * Co * k * k * Ci
        print("%-18s M=%7d N=%4d K=%5d  %8.3f ms  %6.1f TFLOP/s" % (name, b * Ho * Ho, Co, k * k * Ci, ms, fl / ms / 1e9))
        tot_t += ms
        tot_f += fl
    print("TOTAL %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (tot_t, tot_f / tot_t / 1e9, 100 * tot_f / tot_t / 1e9 / 157.3))


if __name__ == "__main__":
    main()
