import torch
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops, program as P
dev = torch.device("cuda:0")
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
B, D = 100, 256
x, dy = rnd(B, D, seed=1, scale=2.0), rnd(B, D, seed=2)
w, b = 1 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5)
mask = ((torch.rand(B, D, generator=torch.Generator().manual_seed(6)) > 0.3).float() / 0.7)
xx = x.double(); m = xx.mean(1, keepdim=True); r = 1 / torch.sqrt(xx.var(1, unbiased=False, keepdim=True) + 1e-5)
xh = (xx - m) * r
u = mask.double() * (xh * w.double() + b.double())
s = torch.sigmoid(u); ag = s * (1 + u * (1 - s))
gl_ref = dy.double() * ag * mask.double()
d = lambda t: t.to(dev)
dx, gl, st = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, 2, device=dev)
dw, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
ddy, dxx = d(dy), d(x)
pb = P.ProgramBuilder()
pb.lnbwd(B, D, dxx, D, ddy, D, d(w), d(b), 1e-5, dx, D, act=ops.ACT_SWISH, mask=d(mask), gl=gl, stats=st)
pa = pb.finalize(dev)
pa.run([]); torch.cuda.synchronize()
print("gl err", (gl.cpu().double() - gl_ref).abs().max().item())
E = (gl.cpu().double() - gl_ref).abs()
bad = (E > 1e-4).nonzero()
print("bad elements", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist())))
for rr, cc in bad[:12].tolist():
    print(rr, cc, "dev", gl[rr, cc].item(), "ref", gl_ref[rr, cc].item(), "mask", mask[rr, cc].item(), "dy", dy[rr, cc].item(), "u", u[rr, cc].item())
print("mean err", (st[:, 0].cpu().double() - m[:, 0]).abs().max().item(), "rstd err", (st[:, 1].cpu().double() - r[:, 0]).abs().max().item())
pb = P.ProgramBuilder()
pb.lncol(B, D, gl, D, dxx, D, st, dw, db)
pc = pb.finalize(dev)
pc.run([]); torch.cuda.synchronize()
ref = (gl_ref * xh).sum(0)
e = (dw.cpu().double() - ref).abs()
print("dw err", e.max().item(), "bad cols", (e > 1e-3).nonzero().flatten().tolist())
print("db err", (db.cpu().double() - gl_ref.sum(0)).abs().max().item())
# per row-group check: emulate
