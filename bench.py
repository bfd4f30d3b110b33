"""bench.py — images/sec of the latent-DDPM training step (BASELINE.json metric: "images/sec VAE+UNet train
step, 64x64 bs256"): frozen-encoder forward + reparameterize + q_sample + denoiser fwd/bwd + clip + AdamW
(+ one RCCL all-reduce of the flat gradient buffer per step when N > 1), v2:1254-1264.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as torch.distributed.run (one rank per GPU, RCCL). Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
SPLIT_PIECE_PRODUCTS = 6        # bf16 MFMAs issued per fp32 product in the default "split" GEMM math (flowers_hip.h)
PMC_TRAFFIC_HALO = 2.73e8         # HBM-side bytes per launch of conv3x3_halo_kernel (rocprofv3 PMC, profiles/r01_conv_kernel_pmc.md)
# algorithmic GFLOP per image (forward MACs x2): frozen encoder fwd + denoiser fwd+bwd (3x fwd)
GFLOP_PER_IMAGE_STEP = 2.748


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def conv_shapes(B):
    """The launches of the dominant kernel gemm_kernel<128,128,2,2,A_KC_CONV,B_KC_PLAIN> in one step:
    (name, B,H,W,Cin,Cout,k,stride,pad). Algorithmic FLOP per launch = 2*B*Ho*Wo*Cout*k*k*Cin."""
    s = []
    for i, (C, H) in enumerate(((128, 32), (256, 16), (512, 8))):
        s.append(("down%d 4x4s2" % (i + 1), B, 2 * H, 2 * H, C // 2, C, 4, 2, 1))
        s.append(("res%d.conv1 3x3" % (i + 1), B, H, H, C, C, 3, 1, 1))
        s.append(("res%d.conv2 3x3" % (i + 1), B, H, H, C, C, 3, 1, 1))
    return s


def measure_dominant_kernel(dev, B, reps=5, halo_mode=0):
    """HIP-event timing (on the stream the kernels are launched on = torch's current stream) of every conv launch of the
    frozen encoder in one step, grouped by the kernel that serves it -> average duration per launch and algorithmic
    TFLOP/s per kernel. In split mode the six 3x3 ResidualBlock convs run on conv3x3_halo_kernel (the dominant kernel
    by time), the three 4x4 stride-2 convs on conv4x4s2_halo_kernel; in fp32 mode all nine run on gemm_kernel."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
    groups = {}
    halo_on = os.environ.get("FL_CONV_HALO", "1") != "0"
    old_mode = ops.set_conv_halo_mode(halo_mode)  # the instantiation the timed step runs (pipelined step: mode 1)
    for name, b, H, W, Ci, Co, k, s, p in conv_shapes(B):
        x = torch.empty(b, H, W, Ci, device=dev)
        ops.philox_fill(x, seed=11, mode=0)
        w = torch.empty(Co, k, k, Ci, device=dev)
        ops.philox_fill(w, seed=12, mode=0)
        bias = torch.zeros(Co, device=dev)
        Ho = (H + 2 * p - k) // s + 1
        y = torch.empty(b, Ho, Ho, Co, device=dev)
        # the launch the step itself makes: in split mode the frozen encoder's weights are pre-split (B_FRAGS operand)
        presplit = ops.get_gemm_math() == "split" and ops.presplit_ok(Ci, Co, Ho * Ho)
        h2 = presplit and os.environ.get("FL_CONV_F16X2", "1") != "0" and ops.f16x2_ok(x, Co, k, k, s, p)
        if h2:  # f16x2 split math (csrc/conv_h2.hip): what engine.conv_ln_infer launches for the ResidualBlock convs
            wh2, geom = ops.pack_weight_f16x2(w), tuple(w.shape)
            amax = torch.full((1,), 8.0, device=dev)  # range information: philox_fill mode 0 (normal) draws are bounded by 8
            run = lambda: ops.conv2d_fwd_f16x2(x, wh2, geom, bias, s, p, out=y, x_amax=amax)
        elif presplit:
            wsp, geom = ops.split_pack_weight(w), tuple(w.shape)
            run = lambda: ops.conv2d_fwd_presplit(x, wsp, geom, bias, s, p, out=y)
        else:
            run = lambda: ops.conv2d_fwd(x, w, bias, s, p, out=y)
        kern = "gemm"
        if h2:
            kern = "h2"
        elif presplit and halo_on and k == 3 and s == 1 and H in (8, 16, 32):
            kern = "halo"
        elif presplit and halo_on and k == 4 and s == 2 and H in (16, 32, 64):
            kern = "halo_s2"
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * b * Ho * Ho * Co * k * k * Ci
        by = 4.0 * b * (H * H * Ci + Ho * Ho * Co) + (6.0 if presplit else 4.0) * Co * k * k * Ci  # x + y + weights, once
        gr_pp = 3 if h2 else (SPLIT_PIECE_PRODUCTS if ops.get_gemm_math() == "split" else 0)
        gr = groups.setdefault(kern, dict(t=0.0, f=0.0, by=0.0, n=0, per=[], pp=gr_pp))
        gr["t"] += ms
        gr["f"] += fl
        gr["by"] += by
        gr["n"] += 1
        gr["per"].append((name, ms, fl / ms / 1e9))
    ops.set_conv_halo_mode(old_mode)
    res = {}
    for kname, gr in groups.items():
        res[kname] = dict(avg_ms=gr["t"] / gr["n"], tflops=gr["f"] / gr["t"] / 1e9, flop_per_launch=gr["f"] / gr["n"],
                          bytes_per_launch=gr["by"] / gr["n"], launches=gr["n"], total_ms=gr["t"], per=gr["per"], pp=gr["pp"])
    return res


def split_vs_fp32_mfma(dev):
    """In-run evidence for the default GEMM math: the same convolution (res2.conv shape at batch 32, K = 2304) through
    the exact-bf16-split path and through the native fp32 MFMA -> max |difference| relative to max |output|."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
    old = ops.get_gemm_math()
    x = torch.empty(32, 16, 16, 256, device=dev)
    ops.philox_fill(x, seed=21, mode=0)
    w = torch.empty(256, 3, 3, 256, device=dev)
    ops.philox_fill(w, seed=22, mode=0)
    outs = {}
    for mode in ("fp32", "split"):
        ops.set_gemm_math(mode)
        outs[mode] = ops.conv2d_fwd(x, w, None, 1, 1)
    ops.set_gemm_math(old)
    return float((outs["split"] - outs["fp32"]).abs().max() / outs["fp32"].abs().max())


def cpu_baseline(threads, B=32, iters=6):
    """The CPU oracle (validated restatement of the reference's step, same torch-CPU ops) timed on this box's
    host cores on a bounded sample of the same workload. Checker code: imported only here."""
    from oracle import procedural_init as pinit
    from oracle import v2_oracle as O
    import json as _json
    shapes = _json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    torch.set_num_threads(threads)
    ae_sd = pinit.procedural_state_dict(shapes["autoencoder"], 42)
    un_sd = pinit.procedural_state_dict(shapes["unet"], 44)
    g = torch.Generator().manual_seed(0)
    data, labels = torch.rand(B, 3, 64, 64, generator=g), torch.randint(0, 102, (B,), generator=g)
    re, t, eps = torch.randn(B, 256, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randn(B, 256, generator=g)
    O.ddpm_step(ae_sd, un_sd, data, labels, re, t, eps)
    t0 = time.perf_counter()
    for _ in range(iters):
        O.ddpm_step(ae_sd, un_sd, data, labels, re, t, eps)
    dt = time.perf_counter() - t0
    return dict(value=B * iters / dt, unit="images/s", cores=threads, kind="port",
                sample="%d steps of the v2 latent-DDPM step at batch %d (oracle.ddpm_step, fp32, torch %s)" % (iters, B, torch.__version__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE metric: 256)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="do not pipeline encoder(k+1) under the denoiser update(k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-math", choices=("split", "fp32"), default=None,
                    help="split (default): fp32 operands split exactly into 3 bf16 pieces, 6 piece products on the bf16 "
                         "matrix pipe; fp32: native v_mfma_f32_32x32x2_f32")
    a = ap.parse_args()
    if os.environ.get("BENCH_WATCHDOG"):  # debugging aid: dump the python stack if a phase stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    # rehearsal knobs (1-GPU boxes): FL_DIST_BACKEND=gloo + FL_SHARE_DEVICE=1 run all ranks on cuda:0 over gloo
    backend = os.environ.get("FL_DIST_BACKEND", "nccl")
    if os.environ.get("FL_SHARE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
    from oxford_102_flower_gan_vae_latent_diffusion_amd import ops, steps

    if a.gemm_math is not None:
        ops.set_gemm_math(a.gemm_math)
    gemm_math = ops.get_gemm_math()
    gemm_math_note = ("split: fp32 operands split exactly into 3 bf16 pieces (x = h+m+l), 6 of 9 piece products on "
                      "v_mfma_f32_32x32x16_bf16, fp32 accumulation; dropped terms < 2^-21 per product "
                      "(tests/test_kernels_gpu.py::test_split_mfma_is_fp32_equivalent); --gemm-math fp32 = native fp32 MFMA"
                      if gemm_math == "split" else "fp32: native v_mfma_f32_32x32x2_f32")

    torch.manual_seed(1234)  # identical initial weights on every rank (random init of the v2 architecture)
    ae = M.SimpleAutoencoder().to(dev).eval()
    un = M.ConditionalUNet().to(dev).train()  # the reference trains the denoiser in train() mode (dropout active)
    un.rng.manual_seed(44 + rank)
    ae.rng.manual_seed(42 + rank)
    tr = steps.DDPMTrainer(ae, un, lr=1e-3, process_group=pg, world_size=world)
    tr.diffusion.rng.manual_seed(1234 + rank)

    B = a.batch
    nb = 4  # distinct synthetic Flowers-shaped batches resident in HBM (seed offset = rank)
    batches, labels = [], []
    for i in range(nb):
        d = torch.empty(B, 3, 64, 64, device=dev)
        ops.philox_fill(d, seed=2000 + 97 * rank + i, mode=2)
        l = torch.empty(B, dtype=torch.int64, device=dev)
        ops.philox_randint(l, seed=3000 + 97 * rank + i, hi=102)
        batches.append(d)
        labels.append(l)
    sdata, slabels = batches[0].clone(), labels[0].clone()

    log("models + synthetic batches resident in HBM (B=%d per GPU)" % B)
    use_graph = not a.no_graph
    overlap = use_graph and not a.no_overlap
    if overlap:
        tr.capture_pipelined(sdata, slabels)
        tr.prime_pipeline(batches[0], labels[0])
        log("hipGraphs captured (encoder || denoiser pipeline)")
    elif use_graph:
        tr.capture(sdata, slabels)
        log("hipGraph captured")

    def one_step(i):
        if overlap:  # step i trains on batch i (already encoded) while batch i+1 is being encoded
            return tr.replay_pipelined(batches[(i + 1) % nb], labels[(i + 1) % nb])
        if use_graph:
            sdata.copy_(batches[i % nb])
            slabels.copy_(labels[i % nb])
            return tr.replay()
        return tr.step(batches[i % nb], labels[i % nb])

    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize(dev)
    log("warm-up done")

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = one_step(a.warmup + i)
    t_enq = time.perf_counter() - t0  # host time to enqueue the timed steps (not part of any reported metric)
    if overlap:
        tr.sync_pipeline()
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss)
    assert final_loss == final_loss, "loss is NaN"
    log("timed region: %d steps in %.3f s (loss %.4f); host enqueue %.3f ms/step" % (a.steps, dt, final_loss, 1e3 * t_enq / a.steps))

    out = {
        "metric": "images/sec VAE+UNet train step, 64x64 bs256",
        "value": world * B * a.steps / dt,
        "unit": "images/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1000.0 * dt / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[2]/[3]: v2 latent-DDPM eps-prediction training step on frozen-VAE "
                        "latents (encoder fwd + U-Net fwd/bwd + clip + AdamW), 102-class cond, 64x64x3 synthetic "
                        "Flowers-shaped batches, batch %d per GPU, random-init weights" % B,
            "global_batch": world * B,
            "parallelism": "dp%d" % world,
            "hipgraph": use_graph,
            "encoder_denoiser_overlap": overlap,
            "loss": final_loss,
            "gemm_math": gemm_math_note,
        },
    }
    if rank == 0 and world == 1:
        if not a.no_roofline:
            pipe_mode = int(os.environ.get("FL_PIPE_HALO_MODE", "1")) if overlap else 0
            groups = measure_dominant_kernel(dev, B, halo_mode=pipe_mode)
            dom = max(groups, key=lambda kname: groups[kname]["total_ms"])
            r = groups[dom]
            log("dominant kernel (%s): %.1f TFLOP/s avg over %d launches" % (dom, r["tflops"], r["launches"]))
            split = gemm_math == "split"
            # the fraction is taken against the peak of the instruction the kernel actually issues: in split mode one
            # algorithmic fp32 flop costs SPLIT_PIECE_PRODUCTS bf16 flops, so the ceiling of the scheme is 2500/6 TFLOP/s
            def peak_of(gr):  # ceiling of the scheme the kernel issues: 16-bit dense MFMA peak / piece products per fp32 product
                return BF16_MFMA_PEAK_TFLOPS / gr["pp"] if gr["pp"] else FP32_MFMA_PEAK_TFLOPS
            peak = peak_of(r)
            math_note = ("6 x v_mfma_f32_32x32x16_bf16 per 16 k on exact bf16 pieces of the fp32 operands" if split
                         else "v_mfma_f32_32x32x2_f32")
            v3 = "1: 8 waves x 32x64 tiles, leaves VGPRs/LDS for the denoiser stream" if pipe_mode in (1, 2) else "2: 4 waves x 64x64 tiles"
            names = {"h2": "conv3x3_h2_kernel<W> (3x3 s1 p1 conv of the encoder's ResidualBlocks in f16x2 split math: operands scaled by "
                           "a power of two and split into 2 fp16 pieces, 3 x v_mfma_f32_32x32x16_f16 per 16 k, fp32 accumulation; halo "
                           "split once per 16-channel chunk at LDS staging, pre-split weights via LDS-DMA ring)",
                     "halo": "conv3x3_halo_kernel<W," + v3 + "> (3x3 s1 p1 conv of the encoder's ResidualBlocks: input halo "
                             "staged once per 32-channel chunk, weights pre-split, " + math_note + ")",
                     "halo_s2": "conv4x4s2_halo_kernel<Wo> (4x4 s2 p1 down-sampling convs: input halo staged once per 16-channel "
                                "chunk, weights pre-split, " + math_note + ")",
                     "gemm": ("gemm_kernel<128,128,2,2,A_KC_CONV,B_FRAGS> (implicit-GEMM conv, weights pre-split, " if split
                              else "gemm_kernel<128,128,2,2,A_KC_CONV,B_KC_PLAIN> (implicit-GEMM conv, ") + math_note + ")"}
            # HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, averaged over the kernel's
            # launches at B=256; profiles/r01_conv_kernel_pmc.md). Not re-measured live: PMC needs the profiler.
            traffic = {"h2": None, "halo": PMC_TRAFFIC_HALO, "halo_s2": None, "gemm": 4.4e8 if split else 4.08e8}[dom] if B == 256 else None
            out["roofline"] = {
                "bound": "mfma",
                "kernel": names[dom],
                "achieved": r["tflops"],
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": r["tflops"] / peak,
                "peak_basis": ("16-bit dense MFMA peak %.0f / %d piece products per fp32 product" % (BF16_MFMA_PEAK_TFLOPS, r["pp"]))
                              if r["pp"] else "fp32 dense MFMA peak",
                "achieved_over_fp32_mfma_peak": r["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                "split_vs_fp32_mfma_max_rel_diff": split_vs_fp32_mfma(dev),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": r["bytes_per_launch"],
                "avg_launch_ms": r["avg_ms"],
                "algorithmic_gflop_per_launch": r["flop_per_launch"] / 1e9,
                "launches_per_step": r["launches"],
                "other_conv_kernels": {names[kname]: {"achieved": g2["tflops"], "frac": g2["tflops"] / peak_of(g2),
                                                      "avg_launch_ms": g2["avg_ms"], "launches_per_step": g2["launches"]}
                                       for kname, g2 in groups.items() if kname != dom},
                "all_encoder_convs_tflops": sum(g2["flop_per_launch"] * g2["launches"] for g2 in groups.values())
                                            / sum(g2["total_ms"] for g2 in groups.values()) / 1e9,
                "step_frac_of_mfma_roofline": (GFLOP_PER_IMAGE_STEP * B * a.steps / dt / 1e3) / peak,
            }
        if not a.no_cpu_baseline:
            # one GPU's share of the host (the affinity mask of a shared 8-GPU host is much wider than this process should use)
            threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
            log("timing the CPU oracle on %d host threads ..." % threads)
            out["cpu_baseline"] = cpu_baseline(threads)
            log("cpu baseline done")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
