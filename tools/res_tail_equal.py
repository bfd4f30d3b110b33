"""Checksums of res_tail_out at the three encoder stages (with and without the fused residual pre-op), for an A/B of FL_TAIL_STREAM=0 / 1
in two processes: the streaming kernel must be bit-identical to the general one."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
dev = torch.device("cuda:0")
h = hashlib.sha256()
outs = []
for B, H, C in ((256, 32, 128), (40, 16, 256), (24, 8, 512), (3, 8, 64)):
    g = torch.Generator().manual_seed(H)
    y, x = torch.randn(B, H, H, C, generator=g).to(dev), torch.randn(B, H, H, C, generator=g).to(dev)
    sc, sh = (torch.rand(B, C, generator=g) + 0.5).to(dev), (torch.randn(B, C, generator=g) * 0.1).to(dev)
    s, saw = torch.rand(C, generator=g).to(dev), (torch.randn(1, 2, 7, 7, generator=g) * 0.1).to(dev).contiguous()
    m = ops.res_tail_map(y, sc, sh, s)
    for pre in ((sc, sh), None):
        for fs in (True, False):
            out, att = ops.res_tail_out(y, sc, sh, s, m, saw, xres=x, final_swish=fs, save_att=True, xres_pre=pre)
            h.update(out.cpu().numpy().tobytes())
            h.update(att.cpu().numpy().tobytes())
            outs.append(out.cpu())
    out = ops.res_tail_out(y, None, None, None, m, saw, xres=x, final_swish=True)
    h.update(out.cpu().numpy().tobytes())
print(h.hexdigest())
if len(sys.argv) > 1:
    if os.path.exists(sys.argv[1]):
        ref = torch.load(sys.argv[1])
        for i, (a, b) in enumerate(zip(outs, ref)):
            print(i, "max abs diff %.3e  max |ref| %.3e  differing %d of %d" % ((a - b).abs().max().item(), b.abs().max().item(), int((a != b).sum()), a.numel()))
    else:
        torch.save(outs, sys.argv[1])

