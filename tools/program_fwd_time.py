"""ms of the folded ConditionalUNet forward (batch 256, hipGraph replay) with whatever library FL_LIB_PATH names: the timing
leg of the -DFL_PG_ABLATE builds (tools/build_prog_ablate_lib.sh)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, program as P  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 256
un = M.ConditionalUNet().to(dev).eval()
x = torch.randn(B, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
c = torch.randint(0, 102, (B,), device=dev)
out = torch.empty(B, 256, device=dev)
with torch.no_grad():
    fp = P.FoldedUNetProgram(un, B, dev)
    fp(x, t, c, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fp(x, t, c, out=out)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    torch.cuda.synchronize()
    print("%s: folded forward %.4f ms" % (os.path.basename(os.environ.get("FL_LIB_PATH", "libflowers_hip.so")), (time.perf_counter() - t0) / 500 * 1e3))
