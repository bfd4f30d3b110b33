"""One launch of each frozen-encoder conv (as the step launches it, f16x2 kernels with the fused pre-op) plus a streaming
calibration kernel (fl_affine_act over a 134 MB tensor: reads 134 MB, writes 134 MB), for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o fetch -- python3 tools/conv_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d DIR -o write -- python3 tools/conv_pmc.py
tools/pmc_summary.py turns the two databases into profiles/r02_conv_h2_pmc.{md,json}."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import conv_shapes  # noqa: E402
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 256
runs = []
for name, b, H, W, Ci, Co, k, s, p in conv_shapes(B):
    x = torch.empty(b, H, W, Ci, device=dev)
    ops.philox_fill(x, seed=11, mode=0)
    w = torch.empty(Co, k, k, Ci, device=dev)
    ops.philox_fill(w, seed=12, mode=0)
    bias = torch.zeros(Co, device=dev)
    Ho = (H + 2 * p - k) // s + 1
    y = torch.empty(b, Ho, Ho, Co, device=dev)
    wh2, geom = ops.pack_weight_f16x2(w), tuple(w.shape)
    amax = torch.full((1,), 8.0, device=dev)
    pre = None if name in ("down2 4x4s2", "down3 4x4s2") else (torch.ones(b, Ci, device=dev), torch.zeros(b, Ci, device=dev))
    runs.append(lambda x=x, wh2=wh2, geom=geom, bias=bias, s=s, p=p, y=y, amax=amax, pre=pre:
                ops.conv2d_fwd_f16x2(x, wh2, geom, bias, s, p, out=y, x_amax=amax, pre=pre))
cal = torch.empty(256, 32, 32, 128, device=dev)
ops.philox_fill(cal, seed=1, mode=0)
sc, sh = torch.ones(256, 128, device=dev), torch.zeros(256, 128, device=dev)
cal_out = torch.empty_like(cal)
torch.cuda.synchronize()
for rep in range(3):  # the LAST repetition is the one summarised (warm caches, steady clocks)
    ops.affine_act(cal, sc, sh, act=ops.ACT_SWISH, out=cal_out)
    for r in runs:
        r()
torch.cuda.synchronize()
print("done")
