"""HIP path vs the round-2 reference fixtures (tests/golden/make_golden_r2.py): train-mode dropout semantics and the
GRADIENTS / pre-clip norms of the reference's own two training loops (not only post-Adam weights)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import procedural_init as pinit
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
from oxford_102_flower_gan_vae_latent_diffusion_amd import engine_bwd, schedules, steps

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
from tests.test_oracle_golden_r2 import AE_KEYS, D_GRAD_KEYS, D_NOISE_KEYS, UNET_KEYS  # noqa: E402


def load(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind != "U" else v) for k, v in np.load(os.path.join(G, name + ".npz")).items()}


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, "%s: %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item()
    ref = max(1e-6, b.abs().max().item())
    assert err <= tol * ref, "%s: max err %.3e (ref max %.3e)" % (what, err, ref)


def seeded(module, seed):
    module.load_state_dict(pinit.procedural_state_dict(module.state_dict(), seed))
    return module


def sl(t, n=64):
    return t.detach().reshape(-1)[:n]


def test_unet_train_mode_vs_reference(dev):
    """train(): the reference's own dropout masks (nn.Dropout + MultiheadAttention attention dropout) -> eps and gradients."""
    f = load("unet_train")
    un = seeded(M.ConditionalUNet(), 44).to(dev).train()
    bm = [f["block_mask%d" % i].to(dev) for i in range(4)]
    hm = [f["head_mask%d" % i].to(dev) for i in range(4)]
    out = un(f["x"].to(dev), f["t"].to(dev), f["c"].to(dev), block_masks=bm, head_masks=hm)
    close(out, f["eps_train"], 5e-5, "train-mode eps")
    assert float(((out.detach().cpu() - f["eps_train"]) ** 2).mean()) <= 1e-9
    (out * f["dout"].to(dev)).sum().backward()
    none = set(str(n) for n in f["none_grad_names"])
    tot = 0.0
    for n, p in un.named_parameters():
        if n in none:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            tot += float((p.grad.double() ** 2).sum())
    close(math.sqrt(tot), f["grad_norm"], 1e-4, "grad norm")
    named = dict(un.named_parameters())
    for k in UNET_KEYS:
        close(sl(named[k].grad), f["grad__" + k.replace(".", "__")], 2e-4, "grad " + k)


def test_classifier_train_mode_vs_reference(dev):
    f = load("classify_train")
    ae = seeded(M.SimpleAutoencoder(), 42).to(dev).train()
    with torch.no_grad():
        logits = engine_bwd.classify(ae, f["z"].to(dev), (f["mask1"].to(dev), f["mask2"].to(dev)))
    close(logits, f["logits"], 2e-5, "train-mode logits")


def test_ddpm_trainer_gradients_vs_reference_loop(dev):
    """gradient VALUES and pre-clip norm of both iterations of the reference's train_conditional_diffusion."""
    f, gr = load("ddpm_loop"), load("ddpm_loop_grads")
    ae = seeded(M.SimpleAutoencoder(), 42).to(dev).eval()
    un = seeded(M.ConditionalUNet(), 44).to(dev).eval()
    tr = steps.DDPMTrainer(ae, un, lr=1e-3)
    for i in range(2):
        tr.step(f["data%d" % i].to(dev), f["labels%d" % i].to(dev), f["reparam_eps%d" % i].to(dev), f["t%d" % i].to(dev),
                f["eps%d" % i].to(dev))
        close(tr.opt.grad_norm.reshape(()), gr["grad_norm%d" % i].reshape(()), 1e-4, "pre-clip grad norm, step %d" % i)
        for k in UNET_KEYS:
            close(sl(tr.fp.gviews[k]), gr["grad%d__%s" % (i, k.replace(".", "__"))], 2e-4, "step %d grad %s" % (i, k))


def vgg_module(dev):
    return M.VGGPerceptualLoss(dev, state_dict=pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7))


def test_vae_gan_trainer_gradients_vs_reference_loop(dev):
    """gradients of both networks, pre-clip norm and the lr / beta1 schedule of the reference's own train_autoencoder.

    Conditioning: on these random-init weights ANY fp32 evaluation of these gradients sits 1e-3..5e-3 (relative to each
    tensor's max) away from the fp64 evaluation of the same algorithm - the reference's own fp32 CPU run included (oracle
    in double: grad norm 35.021, the reference's fp32 run 34.988, this HIP path 35.023; tools/vaegan_grad_diag.py, DESIGN.md
    section 5): the gradients of this deep, normalisation-heavy net cancel heavily. The bar is therefore "as accurate as
    the reference's own fp32 run": per tensor, the HIP gradient's distance to the fp64 oracle is at most 4x the distance of
    the fp32 reference (step 0: the reference fixture itself; step 1: the fp32 oracle, which test_oracle_golden_r2 pins
    to the reference at 5e-4) plus 2e-4. Step 1 starts from the HIP path's own post-step-0 weights (Adam's first step is
    ~lr*sign(g): which elements flip is decided by that fp32 noise, so the runs legitimately hold different weights)."""
    from oracle import v2_oracle as O
    f, gr = load("vae_gan_loop"), load("vae_gan_loop_grads")
    ae = seeded(M.SimpleAutoencoder(), 42).to(dev).train()
    D = seeded(M.Discriminator64(), 43).to(dev).train()
    vgg_sd = pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7)
    tr = steps.VAEGANTrainer(ae, D, vgg_module(dev), lr=1e-4)
    sch = schedules.OneCycle(tr.opt, 1e-4, total_steps=2)  # v2:1031-1037 (num_epochs 1 x 2 batches)
    cpu = lambda sd, dt: {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu()) for k, v in sd.items()}
    vggp = lambda dt: {"feature_extractor." + k: v.to(dt) for k, v in vgg_sd.items()}

    def dist(a, b):
        a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
        return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())

    def ok(hip, r64, r32, gn):
        """within 4x of the fp32 reference's own error (+ 2e-4; measured: <= 1.3x with --gemm-math fp32, <= 3.5x on a few
        tensors in the default split math), or an absolute error below 1e-7 of the total gradient norm
        (tensors whose whole gradient is ~1e-5 of the norm, e.g. the CALayer convs: immaterial to the clipped update)"""
        e_hip, e_ref = dist(hip, r64), dist(r32, r64)
        abs_err = (torch.as_tensor(hip).detach().cpu().double() - torch.as_tensor(r64).detach().cpu().double()).abs().max().item()
        return (e_hip <= 4 * e_ref + 2e-4 or abs_err <= 1e-7 * gn), "HIP %.2e vs fp32 reference %.2e from fp64 (abs %.2e)" % (e_hip, e_ref, abs_err)

    for i in range(2):
        assert abs(float(tr.opt.lr) - float(gr["lr%d" % i])) <= 1e-6 * float(gr["lr%d" % i]) + 1e-12
        assert abs(tr.opt.betas[0] - float(gr["beta1_%d" % i])) <= 1e-9
        ae_w, d_w = ae.state_dict(), D.state_dict()
        ae64, d64, ae32, d32 = cpu(ae_w, torch.float64), cpu(d_w, torch.float64), cpu(ae_w, torch.float32), cpu(d_w, torch.float32)
        out = tr.step(f["data%d" % i].to(dev), f["labels%d" % i].to(dev), epoch=0, num_epochs=1,
                      reparam_eps=f["reparam_eps%d" % i].to(dev))
        kw = dict(epoch=0, num_epochs=1, lr=float(gr["lr%d" % i]), ae_beta1=float(gr["beta1_%d" % i]))
        r64 = O.vae_gan_step(ae64, d64, vggp(torch.float64), f["data%d" % i].double(), f["labels%d" % i], f["reparam_eps%d" % i].double(), **kw)
        if i == 0:  # the reference run itself
            ref32_norm = gr["grad_norm0"]
            ref32_ae = {k: gr["ae_grad0__" + k.replace(".", "__")] for k in AE_KEYS}
            ref32_d = {k: gr["d_grad0__" + k.replace(".", "__")] for k in D_GRAD_KEYS}
        else:       # the reference-pinned fp32 oracle from the same weights
            r32 = O.vae_gan_step(ae32, d32, vggp(torch.float32), f["data%d" % i], f["labels%d" % i], f["reparam_eps%d" % i], **kw)
            ref32_norm = r32["grad_norm"]
            ref32_ae = {k: sl(r32["ae_grads"][k]) for k in AE_KEYS}
            ref32_d = {k: sl(r32["d_grads"][k]) for k in D_GRAD_KEYS}
        if i == 0:
            # (step 1: the discriminator's Adam moments of step 0 shape its update BEFORE the generator's adversarial term,
            # v2:1097-1105, and the oracle call above starts with fresh moments - so only what does not depend on them is
            # compared there: the discriminator's own gradients and the loss terms)
            e_hip, e_ref = dist(out["grad_norm"].reshape(()), r64["grad_norm"].reshape(())), dist(ref32_norm.reshape(()), r64["grad_norm"].reshape(()))
            assert e_hip <= 2 * e_ref + 2e-4, "step %d grad norm: HIP %.2e vs fp32 reference %.2e from fp64" % (i, e_hip, e_ref)
            for k in AE_KEYS:
                if float(sl(r64["ae_grads"][k]).abs().max()) < 1e-6:
                    assert float(sl(tr.fp.gviews[k]).abs().max()) < 1e-5, k  # analytically zero (conv bias in front of LayerNorm2d)
                    continue
                good, msg = ok(sl(tr.fp.gviews[k]), sl(r64["ae_grads"][k]), ref32_ae[k], float(r64["grad_norm"]))
                assert good, "step %d AE grad %s: %s" % (i, k, msg)
        for k in ("recon", "perceptual", "d_loss"):
            close(out[k].reshape(()), r64["losses"][k].reshape(()), 2e-4, "step %d loss %s" % (i, k))
        for k in D_GRAD_KEYS:
            if k in D_NOISE_KEYS:
                assert float(sl(tr.dfp.gviews[k]).abs().max()) < 1e-5, k
                continue
            good, msg = ok(sl(tr.dfp.gviews[k]), sl(r64["d_grads"][k]), ref32_d[k], float(r64["grad_norm"]))
            # 1e-2 floor: the fake batch differs by ~1e-4 between any two fp32 evaluations of the decoder, which flips a few
            # dozen of the 262k LeakyReLU signs of the discriminator's first layer - a finite change of its weight gradient
            # (seen in every pairing: reference fp32 vs fp64 5e-3 at step 0, HIP vs fp64 1.4e-3 at step 1)
            assert good or dist(sl(tr.dfp.gviews[k]), sl(r64["d_grads"][k])) <= 1e-2, "step %d D grad %s: %s" % (i, k, msg)
        if i == 0:
            sch.step()


# FIXED tolerances of the HIP gradients against the REFERENCE's fp64 run (fixtures of make_golden_r3.py; measured numbers in
# gpurun_out/r3/vg_ref64_*.txt via tools/vaegan_ref64_diag.py, the same in split and native-fp32 math):
#   norm : pre-clip total norm, relative                      measured 3.7e-5 (the reference's own fp32 CPU run: 9.4e-4)
#   abs  : max |error| of a tensor's slice / total norm        measured <= 2.4e-5
#   rel  : max |error| of a tensor's slice / the slice's max   measured <= 7.6e-3 (procedural weights), <= 1.6e-2 (the reference's
#          own initialisation - NOT better conditioned than the procedural weights: the slices are small against the norm)
TOL_REF64 = {"vae_gan_grads64": dict(norm=1e-4, abs=5e-5, rel=1.2e-2), "vae_gan_init_grads64": dict(norm=1e-4, abs=5e-5, rel=2.5e-2)}


@pytest.mark.parametrize("fixture", ["vae_gan_grads64", "vae_gan_init_grads64"])
def test_vae_gan_gradients_vs_reference_fp64(dev, fixture):
    """Iteration 0 of the reference's train_autoencoder (v2:1060-1134) run IN DOUBLE by the reference's own code is the truth:
    pre-clip norm, autoencoder and discriminator gradient slices and the loss terms of the HIP step at a fixed tolerance - on the
    procedural weights of the other fixtures and on the reference's own initialisation under a seed (this package's modules build
    bit-identical weights: checked against the fixture's checksums)."""
    f = load(fixture)
    tol = TOL_REF64[fixture]
    if fixture == "vae_gan_grads64":
        ae = seeded(M.SimpleAutoencoder(), 42)
        D = seeded(M.Discriminator64(), 43)
    else:
        torch.manual_seed(int(f["ae_seed"]))
        ae = M.SimpleAutoencoder()
        torch.manual_seed(int(f["d_seed"]))
        D = M.Discriminator64()
        for k in AE_KEYS:
            assert abs(ae.state_dict()[k].double().abs().sum().item() - float(f["ae_sum__" + k.replace(".", "__")])) <= 1e-9 * float(f["ae_sum__" + k.replace(".", "__")]) + 1e-12, k
        for k in D_GRAD_KEYS:
            assert abs(D.state_dict()[k].double().abs().sum().item() - float(f["d_sum__" + k.replace(".", "__")])) <= 1e-9 * float(f["d_sum__" + k.replace(".", "__")]) + 1e-12, k
    ae, D = ae.to(dev).train(), D.to(dev).train()
    tr = steps.VAEGANTrainer(ae, D, vgg_module(dev), lr=1e-4)
    schedules.OneCycle(tr.opt, 1e-4, total_steps=1)
    out = tr.step(f["data"].float().to(dev), f["labels"].to(dev), epoch=0, num_epochs=1, reparam_eps=f["reparam_eps"].float().to(dev))

    def dist(a, b):
        a, b = torch.as_tensor(a).detach().cpu().double().reshape(-1), torch.as_tensor(b).detach().cpu().double().reshape(-1)
        return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())

    gn = float(f["grad_norm"])
    worst = {"ae": (0.0, ""), "d": (0.0, "")}
    assert dist(out["grad_norm"].reshape(()), f["grad_norm"].reshape(())) <= tol["norm"], "pre-clip norm %.6f vs %.6f" % (float(out["grad_norm"]), gn)
    for which, keys, views, pre in (("ae", AE_KEYS, tr.fp.gviews, "ae_grad__"), ("d", D_GRAD_KEYS, tr.dfp.gviews, "d_grad__")):
        for k in keys:
            ref = f[pre + k.replace(".", "__")].double()
            got = sl(views[k]).detach().cpu().double()
            abs_err = float((got - ref).abs().max())
            assert abs_err <= tol["abs"] * gn, "%s grad %s: |error| %.2e = %.1e of the total norm (tolerance %.0e)" % (which, k, abs_err, abs_err / gn, tol["abs"])
            if float(ref.abs().max()) < 1e-6 * gn:   # analytically zero / immaterial (conv bias in front of a norm layer, CALayer convs)
                continue
            e = dist(got, ref)
            worst[which] = max(worst[which], (e, k))
            assert e <= tol["rel"], "%s grad %s: %.2e of its own max from the reference's fp64 run (tolerance %.1e)" % (which, k, e, tol["rel"])
    hist = {"recon": "hist_recon", "perceptual": "hist_perceptual"}
    for k, hk in hist.items():
        if hk in f:
            close(out[k].reshape(()), f[hk].reshape(()).float(), 2e-4, "loss term " + k)
    print("%s: worst AE gradient %.2e (%s), worst D gradient %.2e (%s), norm %.6f vs %.6f" % (
        fixture, worst["ae"][0], worst["ae"][1], worst["d"][0], worst["d"][1], float(out["grad_norm"]), gn))


def test_vae_gan_capture_replay_equals_eager(dev):
    """the captured step must repack the discriminator weights INSIDE the graph. 2 warm-up + 3 replays
    == 5 eager steps: weights of both networks, BatchNorm buffers, losses."""
    B = 4
    g = torch.Generator().manual_seed(33)
    data, labels = torch.rand(B, 3, 64, 64, generator=g), torch.randint(0, 102, (B,), generator=g)

    def build():
        ae = seeded(M.SimpleAutoencoder(), 42).to(dev).train()
        D = seeded(M.Discriminator64(), 43).to(dev).train()
        ae.rng.manual_seed(7)
        return ae, D, steps.VAEGANTrainer(ae, D, vgg_module(dev), lr=1e-4)

    ae1, D1, t1 = build()
    d, l = data.to(dev), labels.to(dev)
    for _ in range(5):
        o1 = t1.step(d, l, epoch=200, num_epochs=300)
    ae2, D2, t2 = build()
    t2.capture(d.clone(), l.clone(), epoch=200, num_epochs=300)  # runs 2 eager warm-up steps
    for _ in range(3):
        o2 = t2.replay()
    torch.cuda.synchronize()
    for k in ("total", "d_loss", "adv", "kl"):
        close(o2[k].reshape(()), o1[k].reshape(()), 1e-6, "loss " + k)
    for (n, a), (_, b) in zip(D1.state_dict().items(), D2.state_dict().items()):
        close(b.float(), a.float(), 1e-6, "D " + n)
    for (n, a), (_, b) in zip(ae1.state_dict().items(), ae2.state_dict().items()):
        if a.numel() > 4_000_000:
            a, b = a.reshape(-1)[:1 << 20], b.reshape(-1)[:1 << 20]
        close(b.float(), a.float(), 1e-6, "AE " + n)
    # eager inference after replays sees the weights the last replay wrote (packed layouts re-derived, not the graph's)
    with torch.no_grad():
        ae1.eval(), ae2.eval()
        z0 = torch.zeros(B, ae1.latent_dim, device=dev)
        ra, rb = ae1(d, eps=z0)[0], ae2(d, eps=z0)[0]
    close(rb, ra, 2e-5, "eager reconstruction after replays")


def test_vae_gan_resume_across_schedule_boundary(dev, tmp_path):
    """f3: side-car with both optimizers, the OneCycle position (lr AND beta1) and the Philox counters: 4 straight steps
    == 2 steps + save / fresh process state / load + 2 steps, bit for bit."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import checkpoint as ck
    B = 4
    g = torch.Generator().manual_seed(34)
    batches = [(torch.rand(B, 3, 64, 64, generator=g).to(dev), torch.randint(0, 102, (B,), generator=g).to(dev)) for _ in range(4)]

    def build(seed_models=True):
        ae = seeded(M.SimpleAutoencoder(), 42).to(dev).train()
        D = seeded(M.Discriminator64(), 43).to(dev).train()
        ae.rng.manual_seed(11)
        tr = steps.VAEGANTrainer(ae, D, vgg_module(dev), lr=1e-4)
        return ae, D, tr, schedules.OneCycle(tr.opt, 1e-4, total_steps=4)

    ae1, D1, t1, s1 = build()
    for i in range(4):
        t1.step(*batches[i], epoch=200, num_epochs=300)
        if i < 3:
            s1.step()
    ae2, D2, t2, s2 = build()
    for i in range(2):
        t2.step(*batches[i], epoch=200, num_epochs=300)
        s2.step()
    p_model, p_state = os.path.join(tmp_path, "vae_gan_epoch_200.pt"), os.path.join(tmp_path, "vae_gan_epoch_200.trainstate.pt")
    ck.save_vae_gan(p_model, ae2, D2)
    ck.save_train_state(p_state, t2, scheduler=s2)
    ae3 = M.SimpleAutoencoder().to(dev).train()
    D3 = M.Discriminator64().to(dev).train()
    ck.load_vae_gan(p_model, ae3, D3)
    t3 = steps.VAEGANTrainer(ae3, D3, vgg_module(dev), lr=1e-4)
    s3 = schedules.OneCycle(t3.opt, 1e-4, total_steps=4)
    ck.load_train_state(p_state, t3, scheduler=s3)
    assert s3.step_num == 2 and float(t3.opt.lr) == float(t2.opt.lr) and t3.opt.betas == t2.opt.betas
    for i in range(2, 4):
        t3.step(*batches[i], epoch=200, num_epochs=300)
        if i < 3:
            s3.step()
    torch.cuda.synchronize()
    for (n, a), (_, b) in zip(D1.state_dict().items(), D3.state_dict().items()):
        assert torch.equal(a, b), "D " + n
    for (n, a), (_, b) in zip(ae1.state_dict().items(), ae3.state_dict().items()):
        assert torch.equal(a, b), "AE " + n


def test_persistent_program_abort_is_loud_and_recovers(dev):
    """after a barrier time-out (abort word set, counters stale) later launches must do nothing and the
    host must raise on its next call - not return garbage for up to 255 forwards."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import program as P
    un = seeded(M.ConditionalUNet(), 44).to(dev).eval()
    B = 32
    g = torch.Generator().manual_seed(5)
    x, t, c = torch.randn(B, 256, generator=g).to(dev), torch.randint(0, 1000, (B,), generator=g).to(dev), torch.randint(0, 102, (B,), generator=g).to(dev)
    up = P.UNetProgram(un, B, dev)
    ref = up(x, t, c).clone()
    torch.cuda.synchronize()
    up.prog.sync[P.SYNC_ABORT_WORD] = 1            # what a timed-out barrier leaves behind
    up.prog.sync[0] = 12345                        # ... together with stale arrival counters
    torch.cuda.synchronize()
    up(x, t, c)                                    # does nothing on the device; mirrors the abort word to the host
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="aborted"):
        up(x, t, c)
    out = up(x, t, c)                              # re-armed by the raise: works again
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
