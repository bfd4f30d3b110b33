"""RGB stem conv (3->64, 3x3) through fl_conv2d_fwd / fl_conv2d_fwd_instnorm at batch 256, next to the write floor."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
dev = torch.device("cuda:0")
B = 256
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for Ci in (4, 8, 32):
    x = torch.randn(B, 64, 64, Ci, device=dev)
    w = torch.randn(64, 3, 3, Ci, device=dev) * 0.1
    b = torch.zeros(64, device=dev)
    y = torch.empty(B, 64, 64, 64, device=dev)
    lw, lb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    t1 = timeit(lambda: ops.conv2d_fwd(x, w, b, 1, 1, out=y))
    t2 = timeit(lambda: ops.conv2d_fwd_instnorm(x, w, b, 1, 1, lw, lb))
    print("Cin=%d: conv2d_fwd %.1f us   conv2d_fwd_instnorm %.1f us  (write floor %.0f us at 5 TB/s)" % (Ci, t1, t2, y.numel() * 4 / 5e6))
y = torch.empty(B, 64, 64, 64, device=dev)
print("memset-like write: %.1f us" % timeit(lambda: y.fill_(1.0)))
x2 = torch.randn(B, 64, 64, 64, device=dev)
sc = torch.ones(B, 64, device=dev); sh = torch.zeros(B, 64, device=dev)
print("affine_act 268MB in+out: %.1f us" % timeit(lambda: ops.affine_act(x2, sc, sh, act=ops.ACT_SWISH, out=y)))
