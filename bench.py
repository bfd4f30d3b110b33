"""bench.py — images/sec of the latent-DDPM training step (BASELINE.json metric: "images/sec VAE+UNet train
step, 64x64 bs256"): frozen-encoder forward + reparameterize + q_sample + denoiser fwd/bwd + clip + AdamW
(+ one RCCL all-reduce of the flat gradient buffer per step when N > 1), v2:1254-1264.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as torch.distributed.run (one rank per GPU, RCCL). Rank 0 prints ONE JSON line:
the headline metric, `roofline` (dominant kernel, timed live with HIP events), `cpu_baseline` (the CPU oracle on the
box's host cores) and - at N = 1 - `secondary`: the other BASELINE.json configs (VAE-GAN step, 1000-step sampling +
decode, the U-Net forward at batch 256 of the north-star target, the same headline step in native-fp32-MFMA math),
so that this file is the single source of every number quoted in DESIGN.md.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
MFMA16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md, dense 16-bit (bf16 / f16) MFMA peak (v_mfma_f32_32x32x16_*)
BF16X3_PRODUCTS = 6             # bf16 MFMAs per fp32 product in the "split" GEMM math of gemm.hip (flowers_hip.h)
F16X2_PRODUCTS = 3              # f16 MFMAs per fp32 product in the f16x2 conv kernels (conv_h2.hip)
# HBM-side bytes per launch of the f16x2 conv kernels at B=256, averaged over a kernel's launches in one step: rocprofv3
# PMC passes (FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes), profiles/r03_conv_h2_pmc.md (the round-3 kernels;
# r02_conv_h2_pmc.md: the round-2 ones). None = not measured.
PMC_TRAFFIC = {"h2_3x3": None, "h2_4x4": None}
PMC_SOURCE = None
for _tag in ("r03", "r02"):
    try:
        PMC_TRAFFIC.update(json.load(open(os.path.join(ROOT, "profiles", "%s_conv_h2_pmc.json" % _tag))))
        PMC_SOURCE = "profiles/%s_conv_h2_pmc.md" % _tag
        break
    except (OSError, ValueError):
        pass
# algorithmic GFLOP per image (forward MACs x 2)
GFLOP_PER_IMAGE_STEP = 2.748     # config 3/4: frozen encoder fwd + denoiser fwd+bwd (3 x fwd)
GFLOP_PER_IMAGE_VAEGAN = 22.75   # config 2
GFLOP_PER_IMAGE_SAMPLE = 15.65   # config 5
GFLOP_UNET_FWD_B256 = 4.161      # the north-star "U-Net forward at batch 256"

_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def conv_shapes(B):
    """The conv launches of the frozen encoder in one step: (name, B,H,W,Cin,Cout,k,stride,pad).
    Algorithmic FLOP per launch = 2*B*Ho*Wo*Cout*k*k*Cin."""
    s = []
    for i, (C, H) in enumerate(((128, 32), (256, 16), (512, 8))):
        s.append(("down%d 4x4s2" % (i + 1), B, 2 * H, 2 * H, C // 2, C, 4, 2, 1))
        s.append(("res%d.conv1 3x3" % (i + 1), B, H, H, C, C, 3, 1, 1))
        s.append(("res%d.conv2 3x3" % (i + 1), B, H, H, C, C, 3, 1, 1))
    return s


def _event_ms(run, reps):
    """average duration of `run` by HIP events on torch's current stream = the stream the kernels are launched on."""
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def measure_dominant_kernel(dev, B, reps=10):
    """HIP-event timing of every conv launch of the frozen encoder in one step, grouped by the kernel that serves it,
    launched exactly as engine.conv_ln_infer launches it (f16x2 kernels with the fused LayerNorm2d-apply + Swish pre-op
    where the step fuses it) -> average duration per launch and algorithmic TFLOP/s per kernel."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
    groups = {}
    split = ops.get_gemm_math() == "split"
    f16x2_on = os.environ.get("FL_CONV_F16X2", "1") != "0"
    for name, b, H, W, Ci, Co, k, s, p in conv_shapes(B):
        x = torch.empty(b, H, W, Ci, device=dev)
        ops.philox_fill(x, seed=11, mode=0)
        w = torch.empty(Co, k, k, Ci, device=dev)
        ops.philox_fill(w, seed=12, mode=0)
        bias = torch.zeros(Co, device=dev)
        Ho = (H + 2 * p - k) // s + 1
        y = torch.empty(b, Ho, Ho, Co, device=dev)
        presplit = split and ops.presplit_ok(Ci, Co, Ho * Ho)
        h2 = presplit and f16x2_on and ops.f16x2_ok(x, Co, k, k, s, p)
        pp = 0
        if h2:
            wh2, geom = ops.pack_weight_f16x2(w), tuple(w.shape)
            amax = torch.full((1,), 8.0, device=dev)  # range information: normal draws are bounded by 8 (swish only shrinks)
            # every encoder conv but down2 / down3 consumes a raw conv output through the fused pre-op (engine.encoder_infer)
            pre = None if name in ("down2 4x4s2", "down3 4x4s2") else (torch.ones(b, Ci, device=dev), torch.zeros(b, Ci, device=dev))
            run = lambda: ops.conv2d_fwd_f16x2(x, wh2, geom, bias, s, p, out=y, x_amax=amax, pre=pre)
            kern, pp = ("h2_3x3" if k == 3 else "h2_4x4"), F16X2_PRODUCTS
        elif presplit:
            wsp, geom = ops.split_pack_weight(w), tuple(w.shape)
            run = lambda: ops.conv2d_fwd_presplit(x, wsp, geom, bias, s, p, out=y)
            kern, pp = ("halo_3x3" if k == 3 else "halo_4x4"), BF16X3_PRODUCTS
        else:
            run = lambda: ops.conv2d_fwd(x, w, bias, s, p, out=y)
            kern, pp = "gemm", (BF16X3_PRODUCTS if split else 0)
        ms = _event_ms(run, reps)
        fl = 2.0 * b * Ho * Ho * Co * k * k * Ci
        wbytes = {F16X2_PRODUCTS: 4.0, BF16X3_PRODUCTS: 6.0}.get(pp, 4.0) if presplit else 4.0
        by = 4.0 * b * (H * H * Ci + Ho * Ho * Co) + wbytes * Co * k * k * Ci  # x + y + weights, each once
        gr = groups.setdefault(kern, dict(t=0.0, f=0.0, by=0.0, n=0, per=[], pp=pp))
        gr["t"] += ms
        gr["f"] += fl
        gr["by"] += by
        gr["n"] += 1
        gr["per"].append((name, ms, fl / ms / 1e9))
    res = {}
    for kname, gr in groups.items():
        res[kname] = dict(avg_ms=gr["t"] / gr["n"], tflops=gr["f"] / gr["t"] / 1e9, flop_per_launch=gr["f"] / gr["n"],
                          bytes_per_launch=gr["by"] / gr["n"], launches=gr["n"], total_ms=gr["t"], per=gr["per"], pp=gr["pp"])
    return res


KERNEL_NAMES = {
    "h2_3x3": "conv3x3_h2p_kernel<W,PRE,BM> (3x3 s1 p1 convs of the encoder's ResidualBlocks in f16x2 split math: operands scaled by a "
              "power of two and split into 2 fp16 pieces, 3 x v_mfma_f32_16x16x32_f16 per 32 k (two taps x 16 channels), fp32 "
              "accumulation; input halo split once per 16-channel chunk at LDS staging with the producer's LayerNorm2d-apply + Swish "
              "fused in, pre-split weights through an LDS-DMA ring; conv3x3_h2_kernel = the 32x32x16 variant for Cin % 64 != 0)",
    "h2_4x4": "conv4x4s2_h2p_kernel<Wo,PRE> (4x4 s2 p1 down-sampling convs, same math on v_mfma_f32_16x16x32_f16: a 32-wide k step = one "
              "kernel row x 8 channels; conv4x4s2_h2_kernel = the 32x32x16 variant where the 48 KB weight ring does not fit)",
    "halo_3x3": "conv3x3_halo_kernel<W,TM> (r01 kernel, bf16x3 split math: 6 x v_mfma_f32_32x32x16_bf16 per 16 k)",
    "halo_4x4": "conv4x4s2_halo_kernel<Wo,TM> (r01 kernel, bf16x3 split math)",
    "gemm": "gemm_kernel<128,128,2,2,A_KC_CONV,..> (implicit-GEMM conv)",
}


def peak_of(pp):
    """ceiling of the scheme a kernel issues: 16-bit dense MFMA peak / piece products per fp32 product, else the fp32 MFMA peak"""
    return MFMA16_PEAK_TFLOPS / pp if pp else FP32_MFMA_PEAK_TFLOPS


def f16x2_vs_fp32_mfma(dev):
    """In-run evidence for the conv math: the same convolution (res2.conv shape at batch 32, K = 2304) through the f16x2
    kernel and through the native fp32 MFMA -> max |difference| relative to max |output|."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import ops
    old = ops.get_gemm_math()
    x = torch.empty(32, 16, 16, 256, device=dev)
    ops.philox_fill(x, seed=21, mode=0)
    w = torch.empty(256, 3, 3, 256, device=dev)
    ops.philox_fill(w, seed=22, mode=0)
    ops.set_gemm_math("fp32")
    ref = ops.conv2d_fwd(x, w, None, 1, 1)
    ops.set_gemm_math(old)
    if not ops.f16x2_ok(x, 256, 3, 3, 1, 1):
        return None
    y = ops.conv2d_fwd_f16x2(x, ops.pack_weight_f16x2(w), tuple(w.shape), None, 1, 1, x_amax=torch.full((1,), 8.0, device=dev))
    return float((y - ref).abs().max() / ref.abs().max())


# ------------------------------------------------------------------------------------------------ secondary configs
def _wall(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def library_gemm_tflops(dev, n=8192, iters=20):
    """the vendor library's plain bf16 GEMM (torch.mm -> hipBLASLt), n^3: a reference point for the matrix pipe, never part
    of the product path"""
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        torch.mm(a, b)
    torch.cuda.synchronize(dev)
    return 2.0 * n ** 3 * iters / (time.perf_counter() - t0) / 1e12


def secondary_unet_fwd(dev):
    """north-star target: ConditionalUNet forward at batch 256 (4.161 GFLOP of record), as one persistent program launch
    (csrc/program.hip: v_mfma_f32_32x32x2_f32, exact fp32) and as the multi-kernel hipGraph."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import engine, modules as M, ops, program as P
    torch.manual_seed(0)
    un = M.ConditionalUNet().to(dev).eval()
    B = 256
    x = torch.empty(B, 256, device=dev)
    ops.philox_fill(x, 1)
    t = torch.empty(B, dtype=torch.int64, device=dev)
    ops.philox_randint(t, 2, 1000)
    c = torch.empty(B, dtype=torch.int64, device=dev)
    ops.philox_randint(c, 3, 102)
    with torch.no_grad():
        mk = lambda: engine.unet_forward(un, x, t, c)[0]
        mk()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = mk()
        graph = _wall(g.replay, 200)
        up = P.UNetProgram(un, B, dev)
        o2 = up(x, t, c)
        torch.cuda.synchronize()
        up.prog.check()
        err = float((o2 - out).abs().max())
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            up(x, t, c, out=o2)
        pers = _wall(g2.replay, 500)
        up.prog.check()
        fp = P.FoldedUNetProgram(un, B, dev)   # frozen weights: linear chains folded, 13 phases
        o3 = fp(x, t, c)
        torch.cuda.synchronize()
        fp.prog.check()
        err_f = float((o3 - out).abs().max())
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3):
            fp(x, t, c, out=o3)
        folded = _wall(g3.replay, 500)
        fp.prog.check()
    # `ms` = what ConditionalUNet.forward runs in eval mode (frozen weights: folded chains); the algorithmic GFLOP of record
    # stay those of the unfolded network (the folded contractions issue ~1.5x more)
    return {"workload": "ConditionalUNet forward, batch 256, eval (north-star >= 60 % of MFMA roofline target)",
            "ms": folded * 1e3, "ms_unfolded_program": pers * 1e3, "ms_multikernel_graph": graph * 1e3,
            "phases": fp.prog.n_phases, "phases_unfolded": up.prog.n_phases, "algorithmic_gflop": GFLOP_UNET_FWD_B256,
            "tflops": GFLOP_UNET_FWD_B256 / folded / 1e3, "peak": FP32_MFMA_PEAK_TFLOPS,
            "peak_basis": "fp32 dense MFMA peak (exact-fp32 MFMA issued)",
            "frac": GFLOP_UNET_FWD_B256 / folded / 1e3 / FP32_MFMA_PEAK_TFLOPS, "persistent_vs_multikernel_maxabs": err,
            "folded_vs_multikernel_maxabs": err_f}


def secondary_vaegan(dev, iters=20):
    """BASELINE config 2: VAE-GAN full-loss step (recon + KL + VGG + PatchGAN + centre), batch 64, epoch 200, one hipGraph."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, steps
    torch.manual_seed(0)
    ae = M.SimpleAutoencoder().to(dev).train()
    D = M.Discriminator64().to(dev).train()
    vgg = M.VGGPerceptualLoss(dev)  # seeded-synthetic weights (no network for the pretrained ones)
    tr = steps.VAEGANTrainer(ae, D, vgg, lr=1e-4)
    B = 64
    data = torch.empty(B, 3, 64, 64, device=dev)
    ops.philox_fill(data, 5, mode=2)
    labels = torch.empty(B, dtype=torch.int64, device=dev)
    ops.philox_randint(labels, 6, 102)
    tr.capture(data, labels, epoch=200, num_epochs=300)
    dt = _wall(tr.replay, iters, warm=3)
    loss = float(tr.out["total"])
    assert loss == loss, "VAE-GAN loss is NaN"
    tf = GFLOP_PER_IMAGE_VAEGAN * B / dt / 1e3
    return {"workload": "BASELINE.json configs[1]: v2 VAE-GAN full loss step, 64x64 synthetic batch 64, epoch 200 (all six terms), 1 GPU",
            "ms_per_step": dt * 1e3, "images_per_s": B / dt, "algorithmic_gflop_per_image": GFLOP_PER_IMAGE_VAEGAN, "tflops": tf,
            "peak": peak_of(BF16X3_PRODUCTS), "peak_basis": "16-bit dense MFMA peak 2500 / 6 piece products (bf16x3 split GEMMs dominate)",
            "frac": tf / peak_of(BF16X3_PRODUCTS), "loss": loss}


def secondary_sample(dev):
    """BASELINE config 5: 1000-step class-conditional sampling + VAE decode, batch 512 (one persistent launch per step)."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
    torch.manual_seed(0)
    ae = M.SimpleAutoencoder().to(dev).eval()
    un = M.ConditionalUNet().to(dev).eval()
    d = M.ConditionalDenoiseDiffusion(un, n_steps=1000, device=dev)
    B = 512
    c = (torch.arange(B, device=dev) % 102).long()
    with torch.no_grad():
        ae.decode(d.sample((B, 256), dev, c))  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = d.sample((B, 256), dev, c)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        img = ae.decode(z)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    assert img.shape == (B, 3, 64, 64) and bool(torch.isfinite(img).all())
    tot = t2 - t0
    return {"workload": "BASELINE.json configs[4]: v2 p_sample_loop 1000 steps, class-conditional, + VAE decode, batch 512, 1 GPU",
            "s_sampling": t1 - t0, "s_decode": t2 - t1, "images_per_s": B / tot, "ms_per_reverse_step": (t1 - t0),
            "algorithmic_gflop_per_image": GFLOP_PER_IMAGE_SAMPLE, "tflops": GFLOP_PER_IMAGE_SAMPLE * B / tot / 1e3,
            "peak": FP32_MFMA_PEAK_TFLOPS, "peak_basis": "fp32 dense MFMA peak (the sampler program issues exact-fp32 MFMAs)",
            "frac": GFLOP_PER_IMAGE_SAMPLE * B / tot / 1e3 / FP32_MFMA_PEAK_TFLOPS}


def secondary_fp32_step(dev, B, steps_n=20):
    """the headline step again with every GEMM / conv on the native fp32 MFMA (--gemm-math fp32)."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, steps
    old = ops.get_gemm_math()
    ops.set_gemm_math("fp32")
    try:
        torch.manual_seed(1234)
        ae = M.SimpleAutoencoder().to(dev).eval()
        un = M.ConditionalUNet().to(dev).train()
        tr = steps.DDPMTrainer(ae, un, lr=1e-3)
        d = torch.empty(B, 3, 64, 64, device=dev)
        ops.philox_fill(d, seed=2000, mode=2)
        l = torch.empty(B, dtype=torch.int64, device=dev)
        ops.philox_randint(l, seed=3000, hi=102)
        tr.capture_pipelined(d.clone(), l.clone())
        tr.prime_pipeline(d, l)
        for _ in range(3):
            tr.replay_pipelined(d, l)
        tr.sync_pipeline()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps_n):
            tr.replay_pipelined(d, l)
        tr.sync_pipeline()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps_n
    finally:
        ops.set_gemm_math(old)
    tf = GFLOP_PER_IMAGE_STEP * B / dt / 1e3
    return {"workload": "the headline step with --gemm-math fp32 (v_mfma_f32_32x32x2_f32 everywhere)", "ms_per_step": dt * 1e3,
            "images_per_s": B / dt, "tflops": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "peak_basis": "fp32 dense MFMA peak",
            "frac": tf / FP32_MFMA_PEAK_TFLOPS}


# ------------------------------------------------------------------------------------------------ CPU baselines
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_time(fn, iters, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)


def cpu_baseline(threads, quick=False):
    """The CPU oracle (validated restatement of the reference's code path, same torch-CPU ops) timed on
    this box's host cores on BOUNDED samples of configs 1, 2, 3 and 5 (median and min): config 3 - the headline workload and the
    top-level value - at the metric's own batch of 256 (6 timed + 1 warm-up step), the others on >= 3 warm-up and >= 10 timed
    iterations of smaller batches. Checker code: imported only here."""
    from oracle import procedural_init as pinit
    from oracle import v2_oracle as O
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    torch.set_num_threads(threads)
    ae_sd = pinit.procedural_state_dict(shapes["autoencoder"], 42)
    un_sd = pinit.procedural_state_dict(shapes["unet"], 44)
    g = torch.Generator().manual_seed(0)
    per = {}
    it, wm = (3, 1) if quick else (10, 3)
    # config 3: the latent-DDPM step (frozen encoder fwd + U-Net fwd/bwd + clip + AdamW) at the metric's own batch of 256
    # (~1.7 s per step on 16 threads: 6 timed + 1 warm-up step, ~12 s)
    B = 256
    it3, wm3 = (2, 1) if quick else (6, 1)
    data, labels = torch.rand(B, 3, 64, 64, generator=g), torch.randint(0, 102, (B,), generator=g)
    re, t, eps = torch.randn(B, 256, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randn(B, 256, generator=g)
    med, mn = _cpu_time(lambda: O.ddpm_step(ae_sd, un_sd, data, labels, re, t, eps), it3, wm3)
    per["config3_ddpm_step"] = {"value": B / med, "unit": "images/s", "batch": B, "iters": it3, "median_s": med, "min_s": mn}
    # config 1: VAE recon-only train step (AE fwd + bwd + clip + AdamW), batch 8
    B1 = 8
    d1, e1 = torch.rand(B1, 3, 64, 64, generator=g), torch.randn(B1, 256, generator=g)

    def c1():
        ae = O._params(ae_sd)
        names = [k for k, v in ae.items() if v.requires_grad]
        opt = torch.optim.AdamW([ae[k] for k in names], lr=1e-4, weight_decay=1e-5)
        recon = O.autoencoder_forward(ae, d1, e1)[0]
        O.euclidean_distance_loss(recon, d1).backward()
        torch.nn.utils.clip_grad_norm_([ae[k] for k in names if ae[k].grad is not None], 1.0)
        opt.step()

    med, mn = _cpu_time(c1, it, wm)
    per["config1_vae_recon_step"] = {"value": B1 / med, "unit": "images/s", "batch": B1, "iters": it, "median_s": med, "min_s": mn}
    # config 2: VAE-GAN full-loss step, batch 4, epoch 200
    B2 = 4
    d_sd = pinit.procedural_state_dict(shapes["discriminator"], 43)
    vgg_sd = {"feature_extractor." + k: v for k, v in pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7).items()}
    d2, l2, e2 = torch.rand(B2, 3, 64, 64, generator=g), torch.randint(0, 102, (B2,), generator=g), torch.randn(B2, 256, generator=g)
    med, mn = _cpu_time(lambda: O.vae_gan_step(ae_sd, d_sd, vgg_sd, d2, l2, e2, epoch=200, num_epochs=300), it, wm)
    per["config2_vaegan_step"] = {"value": B2 / med, "unit": "images/s", "batch": B2, "iters": it, "median_s": med, "min_s": mn}
    # config 5: reverse steps at batch 512 (scaled to 1000) + decode of 64 images (scaled to 512)
    B5 = 512
    sched = O.schedule()
    x5, c5, n5 = torch.randn(B5, 256, generator=g), torch.arange(B5) % 102, torch.randn(B5, 256, generator=g)
    with torch.no_grad():
        med_s, mn_s = _cpu_time(lambda: O.p_sample(un_sd, sched, x5, 500, c5, n5), max(it, 20), wm)
        z64 = torch.randn(64, 256, generator=g)
        med_d, mn_d = _cpu_time(lambda: O.decoder_forward(ae_sd, "decoder.", z64), max(3, it // 2), 1)
    tot = 1000 * med_s + (B5 / 64) * med_d
    per["config5_sample_decode"] = {"value": B5 / tot, "unit": "images/s", "batch": B5, "median_s_per_reverse_step": med_s,
                                    "min_s_per_reverse_step": mn_s, "median_s_decode_64": med_d,
                                    "extrapolated": "1000 x one reverse step + 8 x decode(64)"}
    top = per["config3_ddpm_step"]
    return dict(value=top["value"], unit="images/s", cores=threads, kind="port",
                sample="%d timed + %d warm-up steps of the v2 latent-DDPM step at batch %d (oracle.ddpm_step, fp32, torch %s); median"
                       % (it3, wm3, B, torch.__version__),
                cpu_model=_cpu_model(),
                cores_visible=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                cores_note="threads = min(16, visible): one GPU's share of the host, as the GPU box allots it",
                median_s=top["median_s"], min_s=top["min_s"], per_config=per)


def _self_launch(n):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same arguments> as a child process; returns
    its exit code. Rendezvous on 127.0.0.1 (the container hostname may not resolve) at a free port."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / tensor sharing across processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: starting %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE metric: 256)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="do not pipeline encoder(k+1) under the denoiser update(k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="3 timed iterations per config instead of 10")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs (N = 1 only)")
    ap.add_argument("--gemm-math", choices=("split", "fp32"), default=None,
                    help="split (default): f16x2 conv kernels + bf16x3 GEMMs (fp32-equivalent split math on the 16-bit matrix "
                         "pipe); fp32: native v_mfma_f32_32x32x2_f32 everywhere")
    a = ap.parse_args()
    if os.environ.get("BENCH_WATCHDOG"):  # debugging aid: dump the python stack if a phase stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the 1-GPU case (`python bench.py --gpus N`): start the N ranks ourselves. Decided BEFORE anything in
        # this process touches the GPU (a process that has initialised HIP must never re-exec): the ranks are children of a
        # parent that only waits; rank 0's JSON line is the parent's stdout.
        raise SystemExit(_self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or unset "
                         "WORLD_SIZE and let bench.py start its ranks)" % (a.gpus, world, a.gpus))
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
    gemm_math_note = ("split: fp32-equivalent split math on the 16-bit matrix pipe. Frozen-encoder convs: f16x2 (operands scaled by a "
                      "power of two, split by round-to-nearest into 2 fp16 pieces, 3 of 4 piece products on v_mfma_f32_16x16x32_f16 / 32x32x16_f16, fp32 "
                      "accumulation; error vs fp64 <= 3e-6 of the output max, 2-4x the native fp32 MFMA kernel's, tests/test_kernels_gpu.py::"
                      "test_f16x2_conv_is_fp32_equivalent). Other GEMMs: bf16x3 (3 round-to-nearest bf16 pieces, 6 of 9 piece products on "
                      "v_mfma_f32_32x32x16_bf16; ::test_split_mfma_is_fp32_equivalent). --gemm-math fp32 = native fp32 MFMA"
                      if gemm_math == "split" else "fp32: native v_mfma_f32_32x32x2_f32")

    torch.manual_seed(1234)  # identical initial weights on every rank (random init of the v2 architecture)
    ae = M.SimpleAutoencoder().to(dev).eval()
    un = M.ConditionalUNet().to(dev).train()  # the reference trains the denoiser in train() mode (dropout active)
    un.rng.manual_seed(44 + rank)
    ae.rng.manual_seed(42 + rank)
    # ranks sharing ONE device (FL_SHARE_DEVICE rehearsal) cannot each run programs that need every CU: multi-kernel denoiser there
    share = bool(os.environ.get("FL_SHARE_DEVICE")) and world > 1
    tr = steps.DDPMTrainer(ae, un, lr=1e-3, process_group=pg, world_size=world, persistent=False if share else None)
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
    # the denoiser as persistent phase programs (default) needs the whole device: encoder graph -> programs on one stream.
    # FL_TRAIN_PROGRAM=0 restores the multi-kernel denoiser, which runs as a second stream next to the encoder of the next batch
    overlap = use_graph and not a.no_overlap and not tr.persistent
    # all-reduce under the next batch's encoder (FL_FORCE_DP_OVERLAP=1: rehearse that schedule on one rank)
    dp_overlap = use_graph and tr.persistent and not a.no_overlap and (world > 1 or os.environ.get("FL_FORCE_DP_OVERLAP") == "1")
    if dp_overlap:
        tr.capture_overlapped(sdata, slabels)
        tr.prime_overlapped(batches[0], labels[0])
        log("hipGraphs captured (all-reduce of step k under the encoder of batch k+1)")
    elif overlap:
        tr.capture_pipelined(sdata, slabels)
        tr.prime_pipeline(batches[0], labels[0])
        log("hipGraphs captured (encoder || denoiser pipeline)")
    elif use_graph:
        tr.capture(sdata, slabels)
        log("hipGraph captured")

    def one_step(i):
        if dp_overlap:
            return tr.replay_overlapped(batches[(i + 1) % nb], labels[(i + 1) % nb])
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
    if tr.persistent:
        tr.check_programs()   # a timed-out device-wide barrier would have voided the timed steps: raise instead of reporting
    log("timed region: %d steps in %.3f s (loss %.4f); host enqueue %.3f ms/step" % (a.steps, dt, final_loss, 1e3 * t_enq / a.steps))
    comm = None
    if world > 1 or os.environ.get("FL_FORCE_DP_OVERLAP") == "1":
        # after the timed region: what the collective costs alone and how much of it the schedule leaves exposed
        # (HIP events on the compute stream around the pieces of ten more steps), so that a scaling run explains itself
        comm = tr.measure_comm(batches, labels, steps_n=10, overlapped=dp_overlap)
        log("all-reduce alone %.3f ms; exposed per step %.3f ms" % (comm["allreduce_ms"], comm["allreduce_exposed_ms"]))

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
            "denoiser_persistent_programs": bool(tr.persistent),
            "allreduce_under_next_encoder": dp_overlap,
            "loss": final_loss,
            "gemm_math": gemm_math_note,
        },
    }
    if comm is not None:
        out["comm"] = comm
    if rank == 0 and world == 1:
        if not a.no_roofline:
            groups = measure_dominant_kernel(dev, B)
            dom = max(groups, key=lambda kname: groups[kname]["total_ms"])
            r = groups[dom]
            log("dominant kernel (%s): %.1f TFLOP/s avg over %d launches" % (dom, r["tflops"], r["launches"]))
            peak = peak_of(r["pp"])
            out["roofline"] = {
                "bound": "mfma",
                "kernel": KERNEL_NAMES[dom],
                "achieved": r["tflops"],
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": r["tflops"] / peak,
                "peak_basis": ("16-bit dense MFMA peak %.0f / %d piece products per fp32 product" % (MFMA16_PEAK_TFLOPS, r["pp"]))
                              if r["pp"] else "fp32 dense MFMA peak",
                "achieved_over_fp32_mfma_peak": r["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                "f16x2_vs_fp32_mfma_max_rel_diff": f16x2_vs_fp32_mfma(dev),
                # context for `frac`: what the matrix pipe sustains on this box under a real load. The vendor library's plain
                # bf16 GEMM (8192^3, measured here) is the practical ceiling of any MFMA kernel that also moves data: the
                # shader clock sits at ~1.85-1.9 GHz under it and under the conv kernel (profiles/r02_conv_power_probe.txt),
                # not at the 2.4 GHz the data-sheet peak assumes
                "issued_16bit_mfma_tflops": r["tflops"] * r["pp"] if r["pp"] else None,
                "library_bf16_gemm_8192_tflops": library_gemm_tflops(dev),
                # context, not the denominator of `frac`: what a bare LDS-fed f16 MFMA loop sustains on RANDOM operands under DVFS
                # (tools/mfma_shape_probe.hip, profiles/r03_mfma_shape_probe.txt; 2149 / 1992 on zero operands)
                "lds_fed_mfma_loop_random_operands_tflops": {"v_mfma_f32_16x16x32_f16": 1600, "v_mfma_f32_32x32x16_f16": 1432},
                "traffic": PMC_TRAFFIC.get(dom) if B == 256 else None,
                "traffic_source": "rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), %s" % PMC_SOURCE if PMC_TRAFFIC.get(dom) else None,
                "algorithmic_bytes_per_launch": r["bytes_per_launch"],
                "avg_launch_ms": r["avg_ms"],
                "algorithmic_gflop_per_launch": r["flop_per_launch"] / 1e9,
                "launches_per_step": r["launches"],
                "per_launch": [{"conv": n, "ms": ms, "tflops": tf} for n, ms, tf in r["per"]],
                "other_conv_kernels": {KERNEL_NAMES[kname]: {"achieved": g2["tflops"], "peak": peak_of(g2["pp"]),
                                                             "frac": g2["tflops"] / peak_of(g2["pp"]), "avg_launch_ms": g2["avg_ms"],
                                                             "launches_per_step": g2["launches"],
                                                             "traffic": PMC_TRAFFIC.get(kname) if B == 256 else None}
                                       for kname, g2 in groups.items() if kname != dom},
                "all_encoder_convs_tflops": sum(g2["flop_per_launch"] * g2["launches"] for g2 in groups.values())
                                            / sum(g2["total_ms"] for g2 in groups.values()) / 1e9,
                "all_encoder_convs_ms": sum(g2["total_ms"] for g2 in groups.values()),
                "step_tflops": GFLOP_PER_IMAGE_STEP * B * a.steps / dt / 1e3,
                "step_frac_of_mfma_roofline": (GFLOP_PER_IMAGE_STEP * B * a.steps / dt / 1e3) / peak,
            }
        if not a.no_secondary:
            del tr  # free the headline trainer's graphs before the other configs build theirs
            torch.cuda.empty_cache()
            sec = {}
            for name, fn in (("unet_fwd_b256", lambda: secondary_unet_fwd(dev)), ("config2_vaegan_b64", lambda: secondary_vaegan(dev)),
                             ("config5_sample1000_decode_b512", lambda: secondary_sample(dev)),
                             ("config3_native_fp32_mfma", lambda: secondary_fp32_step(dev, B))):
                if gemm_math == "fp32" and name == "config3_native_fp32_mfma":
                    continue
                try:
                    sec[name] = fn()
                    log("secondary %s done" % name)
                except Exception as e:  # a secondary leg must never cost the headline line
                    sec[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                    log("secondary %s FAILED: %s" % (name, e))
                torch.cuda.empty_cache()
            out["secondary"] = sec
        if not a.no_cpu_baseline:
            # one GPU's share of the host (the affinity mask of a shared 8-GPU host is much wider than this process should use)
            threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
            log("timing the CPU oracle on %d host threads (configs 3, 1, 2, 5) ..." % threads)
            out["cpu_baseline"] = cpu_baseline(threads, quick=a.quick_cpu_baseline)
            log("cpu baseline done")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
