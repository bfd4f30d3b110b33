"""Frozen-encoder phase alone (stream E of the pipelined step): graph-captured encode() at B=256."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, steps

dev = torch.device("cuda:0")
torch.manual_seed(0)
ae = M.SimpleAutoencoder().to(dev).eval()
un = M.ConditionalUNet().to(dev).train()
tr = steps.DDPMTrainer(ae, un)
B = int(os.environ.get("B", "256"))
data = torch.empty(B, 3, 64, 64, device=dev); ops.philox_fill(data, 1, mode=2)
labels = torch.zeros(B, dtype=torch.int64, device=dev)
for _ in range(3):
    tr.encode(data)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    z = tr.encode(data)
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("encoder graph: %.3f ms" % timeit(g.replay))
if os.environ.get("ENC_ONLY"):
    sys.exit(0)
z0 = z.clone()
g2 = torch.cuda.CUDAGraph()
tr.fwd_bwd(z0, labels); tr.update(); torch.cuda.synchronize()
with torch.cuda.graph(g2):
    tr.fwd_bwd(z0, labels)
    tr.update()
print("denoiser fwd/bwd + clip/AdamW graph: %.3f ms" % timeit(g2.replay))
if tr.persistent:
    tp = list(tr._tprog.values())[0]
    t = tr.diffusion.rng.randint((B,), dev, 1000)
    eps = torch.randn(B, 256, device=dev)
    bm, hm = un.dropout_masks(B, dev)
    xt = torch.randn(B, 256, device=dev)
    print("  train program alone (%d phases, %d jobs): %.3f ms" % (tp.prog.n_phases, tp.prog.n_jobs,
          timeit(lambda: tp.run(xt, t, labels, eps, bm, hm))))
    print("  optimizer program alone: %.3f ms" % timeit(tr._oprog.run))
tr.capture(data, labels)
print("whole step, one stream (encoder + denoiser + optimizer graphs): %.3f ms" % timeit(tr.replay))
