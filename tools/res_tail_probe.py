"""res_tail_out at the stage-1 shape (B=256, 32x32x128) with and without its pieces: where do 124 us go? (timing only)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
dev = torch.device("cuda:0")
B, H, C = 256, 32, 128
y = torch.randn(B, H, H, C, device=dev); x = torch.randn(B, H, H, C, device=dev)
sc = torch.rand(B, C, device=dev) + 0.5; sh = torch.randn(B, C, device=dev) * 0.1
s = torch.rand(C, device=dev); saw = torch.randn(1, 2, 7, 7, device=dev).contiguous() * 0.1
m = ops.res_tail_map(y, sc, sh, s)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("full (map, residual pre-op, final swish): %.1f us" % t(lambda: ops.res_tail_out(y, sc, sh, s, m, saw, xres=x, final_swish=True, xres_pre=(sc, sh))))
print("no 7x7 attention (map=None):             %.1f us" % t(lambda: ops.res_tail_out(y, sc, sh, s, None, None, xres=x, final_swish=True, xres_pre=(sc, sh))))
print("no residual pre-op:                      %.1f us" % t(lambda: ops.res_tail_out(y, sc, sh, s, m, saw, xres=x, final_swish=True)))
print("no scale/shift/s, no pre-op, no map:     %.1f us" % t(lambda: ops.res_tail_out(y, None, None, None, None, None, xres=x, final_swish=True)))
print("res_tail_map:                            %.1f us" % t(lambda: ops.res_tail_map(y, sc, sh, s)))
print("streaming reference affine_act:          %.1f us" % t(lambda: ops.affine_act(y, sc, sh, act=ops.ACT_SWISH)))
