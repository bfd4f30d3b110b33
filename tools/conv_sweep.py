"""K-sweep at fixed M,N to separate per-tile overhead from per-chunk cost: t = a*tiles_per_slot + b*chunks."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops

dev = torch.device("cuda:0")
def run(B, H, C, Co, k, reps=5):
    x = torch.empty(B, H, H, C, device=dev); ops.philox_fill(x, seed=1, mode=0)
    w = torch.empty(Co, k, k, C, device=dev); ops.philox_fill(w, seed=2, mode=0)
    bias = torch.zeros(Co, device=dev)
    y = torch.empty(B, H, H, Co, device=dev)
    ops.conv2d_fwd(x, w, bias, 1, k // 2, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d_fwd(x, w, bias, 1, k // 2, out=y)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * H * H * Co * k * k * C
    print("B=%3d H=%2d C=%3d Co=%3d k=%d  chunks=%4d tiles=%5d  %8.3f ms  %6.1f TF" % (B, H, C, Co, k, k*k*C//32, (B*H*H//128)*((Co+127)//128), ms, fl/ms/1e9))
for k in (1, 3, 5, 7):
    run(256, 32, 128, 128, k)
for B in (64, 128, 256, 512):
    run(B, 32, 128, 128, 3)
for k in (1, 3, 5):
    run(256, 8, 512, 512, k)
