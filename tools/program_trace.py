"""Per-phase timeline of the persistent U-Net forward (needs the -DFL_PROG_TRACE build of the library:
FL_LIB_PATH=.../libflowers_tr.so python tools/program_trace.py). Stamps (100 MHz wall clock, thread 0 of each block):
0 phase start, 1 last MFMA loop done (GEMM tiles), 2 tiles done, 3 barrier entered, 4 release fence done,
5 arrive atomic returned, 6 barrier left."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, program as P  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
un = M.ConditionalUNet().to(dev).eval()
x = torch.randn(B, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
c = torch.randint(0, 102, (B,), device=dev)
TRAIN = os.environ.get("TRAIN", "0") == "1"
if TRAIN:  # the two-launch training step's forward+backward program (program_train.py)
    from oxford_102_flower_gan_vae_latent_diffusion_amd import steps
    from oxford_102_flower_gan_vae_latent_diffusion_amd.program_train import DenoiserTrainProgram, XT, T as TT, CLS as TC, EPS, BM0, HM0
    un.train()
    fp = steps.FlatParams(un)
    up = DenoiserTrainProgram(un, fp.gviews, B, dev)
    eps_t = torch.randn(B, 256, device=dev)
    bm, hm = un.dropout_masks(B, dev)
elif os.environ.get("FOLDED", "0") == "1":  # the eval-mode forward with folded chains (what ConditionalUNet.forward runs)
    up = P.FoldedUNetProgram(un, B, dev)
else:
    up = P.UNetProgram(un, B, dev)
nph = up.prog.n_phases
G = torch.cuda.get_device_properties(0).multi_processor_count
trace = torch.zeros(G * nph * 16, dtype=torch.int64, device=dev)
out = torch.empty(B, 256, device=dev)


def run():
    io = [up._dummy] * 16
    if TRAIN:
        io[XT], io[TT], io[TC], io[EPS] = x, t, c, eps_t
        for i in range(4):
            io[BM0 + i], io[HM0 + i] = bm[i], hm[i]
    else:
        io[P.X], io[P.T], io[P.CLS], io[P.OUT] = x, t, c, out
    io[15] = trace
    up.prog.run(io)


for _ in range(5):
    run()
torch.cuda.synchronize()
trace.zero_()   # stamps a tile kind does not write stay 0 = "missing" below
run()
torch.cuda.synchronize()
up.prog.check()
tr = trace.view(G, nph, 16).cpu().double() * 0.01  # us
print("program: %d phases, %d of them XCD-local (FL_PROG_LOCAL=%s)" % (nph, up.prog.n_local_phases, os.environ.get("FL_PROG_LOCAL", "1")))
t0 = tr[:, 0, 0].min()
print("blocks %d phases %d; total %.1f us (first phase start -> last phase tiles done)" % (G, nph, tr[:, nph - 1, 2].max() - t0))
print("%-5s %8s %8s | %8s %8s %8s %8s %8s | %s" % ("phase", "start", "len", "tiles", "mfma", "fence", "atomic", "spin", "busy blocks"))
jobs_per_phase = up.prog.phases.cpu().view(torch.int32).view(-1, 4)
for p in range(nph):
    s = tr[:, p, 0]
    tiles_done = tr[:, p, 2]
    ntiles = int(jobs_per_phase[p, 2])
    act = (tr[:, p, 2] - tr[:, p, 0]) > 0.3   # blocks that ran a tile (local phases deal tiles by XCD slot, not by block id)
    if not act.any():
        act = tr[:, p, 0] > 0
    start = s.min() - t0
    end = (tr[:, p, 6].max() if p + 1 < nph else tiles_done.max()) - t0
    tile_t = (tiles_done[act] - s[act]).mean()
    m1 = act & (tr[:, p, 1] > 0)
    mf = (tr[m1, p, 1] - s[m1]).mean() if m1.any() else float("nan")
    if p + 1 < nph:
        fence = float("nan")
        atom = float("nan")
        spin = (tr[:, p, 6] - tr[:, p, 5]).mean()
        lastarr = tr[:, p, 5].max() - t0
    else:
        fence = atom = spin = float("nan")
    det = ""
    a = (tr[:, p, 7] > 0) & (tr[:, p, 8] > 0) & (tr[:, p, 1] > 0)   # blocks whose LAST tile of the phase was a stamped GEMM tile
    if a.any():
        det = " ptrs %.2f issue %.2f loop %.2f epi %.2f (%d blocks; last tile of each)" % (
            (tr[a, p, 7] - s[a]).mean(), (tr[a, p, 8] - tr[a, p, 7]).mean(), (tr[a, p, 1] - tr[a, p, 8]).mean(),
            (tr[a, p, 2] - tr[a, p, 1]).mean(), int(a.sum()))
        one = a & (tr[:, p, 3] > s) & (tr[:, p, 4] > 0)
        if one.any():  # (of the block's last tile: dealing + descriptor requests; unpacking; epilogue-operand requests)
            det += " {deal %.2f (window %.2f, next tile %.2f, look-ahead %.2f) unpack %.2f operands %.2f}" % (
                (tr[one, p, 3] - s[one]).mean(), (tr[one, p, 12] - s[one]).mean(), (tr[one, p, 13] - tr[one, p, 12]).mean(),
                (tr[one, p, 3] - tr[one, p, 13]).mean(), (tr[one, p, 4] - tr[one, p, 3]).mean(), (tr[one, p, 7] - tr[one, p, 4]).mean())
        b = a & (tr[:, p, 9] > 0) & (tr[:, p, 10] > 0)    # LDS-DMA tiles: first chunk landed, partial tiles merged
        if b.any():
            det += " [first chunk %.2f, chunks %.2f, merge %.2f, epilogue+prefetch %.2f]" % (
                (tr[b, p, 9] - tr[b, p, 8]).mean(), (tr[b, p, 1] - tr[b, p, 9]).mean(), (tr[b, p, 10] - tr[b, p, 1]).mean(),
                (tr[b, p, 2] - tr[b, p, 10]).mean())
    cyc = trace.view(G, nph, 16)[act, p, 11].double().cpu()
    mhz = (cyc / (tiles_done[act] - s[act])).mean()
    print("%-5d %8.2f %8.2f | %8.2f %8.2f %8.2f %8.2f %8.2f | %d tiles%s  clk %.0f MHz" % (p, start, end - start, tile_t, mf, fence, atom, spin, ntiles, det, mhz))

# per-block phase durations (phase start -> next phase start): meaningful in XCD-local mode too, where the XCDs drift
print("\nper-block phase duration (start -> next start), mean / max over blocks; tile time = mean over blocks that ran a tile")
tot = 0.0
for p in range(nph - 1):
    d = tr[:, p + 1, 0] - tr[:, p, 0]
    ran = tr[:, p, 2] - tr[:, p, 0]
    tile = ran[ran > 0.3].mean() if (ran > 0.3).any() else float("nan")
    tot += d.mean()
    print("%-3d %7.2f %7.2f   tiles %6.2f  (%d tiles)" % (p, d.mean(), d.max(), tile, int(jobs_per_phase[p, 2])))
print("sum of mean phase durations %.1f us; launch span %.1f us" % (tot, tr[:, nph - 1, 2].max() - tr[:, 0, 0].min()))

# where a phase's time beyond its slowest tile goes (over the blocks that ran a tile):
#   drain  = tiles done -> barrier entered (s_waitcnt vmcnt(0) on the write-through stores + __syncthreads)
#   detect = the LAST block's arrive -> the FIRST block leaving (one poll round trip after the last arrival)
#   skew   = first block leaving -> last block leaving;  start = last block leaving -> the next phase's first tile start
print("\nbarrier anatomy per phase (us): slowest tile | drain (mean / of the slowest block) | detect | skew | next start")
acc = [0.0] * 5
for p in range(nph - 1):
    ran = (tr[:, p, 2] - tr[:, p, 0]) > 0.3
    if not ran.any():
        continue
    tile_end = tr[:, p, 2]
    slow = int(torch.argmax(torch.where(ran, tile_end, torch.full_like(tile_end, -1e30))))
    drain_mean = (tr[ran, p, 5] - tr[ran, p, 2]).mean()
    drain_slow = tr[slow, p, 5] - tr[slow, p, 2]
    last_arrive = tr[:, p, 5].max()
    first_leave, last_leave = tr[:, p, 6].min(), tr[:, p, 6].max()
    nxt = tr[:, p + 1, 0].min()
    vals = [tile_end[slow] - tr[slow, p, 0], drain_slow, first_leave - last_arrive, last_leave - first_leave, nxt - last_leave]
    for i, v in enumerate(vals):
        acc[i] += float(v)
    print("%-3d %6.2f | %5.2f / %5.2f | %5.2f | %5.2f | %5.2f" % (p, vals[0], drain_mean, drain_slow, vals[2], vals[3], vals[4]))
print("sums: slowest tiles %.1f, drain %.1f, detect %.1f, skew %.1f, next start %.1f us" % tuple(acc))
