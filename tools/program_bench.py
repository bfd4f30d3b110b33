"""Micro-benchmarks of the persistent phase-program executor: barrier cost, one GEMM phase per shape, row-norm phase."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import ops, program as P  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=200, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def graph_time(prog, io, iters=300):
    prog.run(io)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        prog.run(io)
    us = timeit(g.replay, iters)
    prog.check()
    return us


def barriers(n, max_blocks=0):
    pb = P.ProgramBuilder()
    t = torch.zeros(4, dtype=torch.int64, device=dev)
    f = torch.ones(128, device=dev)
    outs = [torch.empty(1, 256, device=dev) for _ in range(n)]
    for i in range(n):
        pb.sinusoid(1, 256, t, 1, f, outs[i], 256)
        pb.barrier()
    prog = pb.finalize(dev)
    prog.max_blocks = max_blocks
    return graph_time(prog, [])


def gemm_phase(M, N, K, reps=8):
    """`reps` dependent-free repeats of one GEMM phase (separated by barriers) to isolate per-phase time."""
    pb = P.ProgramBuilder()
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.zeros(N, device=dev)
    Cs = [torch.empty(M, N, device=dev) for _ in range(reps)]
    for i in range(reps):
        pb.gemm(M, N, [(A, K, K, W, K)], Cs[i], N, [(b, 1.0)])
        pb.barrier()
    prog = pb.finalize(dev)
    return graph_time(prog, []) / reps


def rownorm_phase(M, D, reps=8):
    pb = P.ProgramBuilder()
    x = torch.randn(M, D, device=dev)
    w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    ys = [(torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)) for _ in range(reps)]
    for i in range(reps):
        pb.rownorm(M, D, x, D, w, b, 1e-5, ys[i][0], D, act=ops.ACT_SWISH, R=x, ldr=D, w2=w, b2=b, C2=ys[i][1])
        pb.barrier()
    prog = pb.finalize(dev)
    return graph_time(prog, []) / reps


if __name__ == "__main__":
    b1, b25 = barriers(1), barriers(26)
    print("launch+1 phase: %.1f us; 26 phases: %.1f us -> %.2f us per barrier+tiny phase" % (b1, b25, (b25 - b1) / 25))
    for mb in (64, 128):
        print("  with %d blocks: 26 phases %.1f us" % (mb, barriers(26, mb)))
    for (M, N, K) in [(256, 256, 256), (256, 256, 512), (256, 512, 512), (256, 512, 768), (256, 1024, 1024),
                      (256, 1024, 1280), (256, 512, 2304), (256, 256, 1024), (512, 1024, 1024)]:
        us = gemm_phase(M, N, K)
        print("GEMM phase M=%d N=%d K=%d: %.2f us  (%.1f TFLOP/s incl. barrier)" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
    for D in (256, 512, 1024):
        print("rownorm phase 256x%d: %.2f us" % (D, rownorm_phase(256, D)))
