"""Numeric check of conv dgrad / wgrad kernels against torch fp64 autograd on the Discriminator64 / VGG / encoder shapes."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for math in ("split", "fp32"):
    ops.set_gemm_math(math)
    for (B, H, Ci, Co, k, s, p, cpad) in ((4, 64, 3, 64, 4, 2, 1, 4), (4, 32, 64, 128, 4, 2, 1, 64), (4, 16, 128, 256, 4, 2, 1, 128),
                                          (4, 64, 3, 64, 3, 1, 1, 4), (4, 32, 128, 128, 3, 1, 1, 128), (4, 64, 64, 32, 3, 1, 1, 64)):
        x = torch.randn(B, Ci, H, H, generator=g, dtype=torch.float64)
        w = torch.randn(Co, Ci, k, k, generator=g, dtype=torch.float64) * 0.05
        x.requires_grad_(True); w.requires_grad_(True)
        y = F.conv2d(x, w, None, s, p)
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        (y * dy).sum().backward()
        xh = torch.zeros(B, H, H, cpad); xh[..., :Ci] = x.detach().permute(0, 2, 3, 1).float()
        wp = torch.zeros(Co, k, k, cpad); wp[..., :Ci] = w.detach().permute(0, 2, 3, 1).float()
        dyh = dy.permute(0, 2, 3, 1).float().contiguous()
        xh, wp, dyh = xh.to(dev), wp.to(dev), dyh.to(dev)
        dw = ops.conv2d_wgrad(dyh, xh, k, k, s, p).cpu().double()[..., :Ci].permute(0, 3, 1, 2)
        dx = ops.conv2d_dgrad(dyh, wp, (H, H), s, p).cpu().double()[..., :Ci].permute(0, 3, 1, 2)
        yf = ops.conv2d_fwd(xh, wp, None, s, p).cpu().double().permute(0, 3, 1, 2)
        e = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
        print("%-5s B%d H%2d %3d->%3d k%d s%d: fwd %.2e  dgrad %.2e  wgrad %.2e" % (math, B, H, Ci, Co, k, s, e(yf, y.detach()), e(dx, x.grad), e(dw, w.grad)))
