"""Round-2 pins of the CPU oracle to the REFERENCE (tests/golden/make_golden_r2.py): train-mode dropout semantics,
gradients of the reference's own two training loops, and the epoch gating of train_autoencoder. CPU only."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import procedural_init as pinit
from oracle import v2_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UNET_KEYS = ["residual_weight", "time_emb.lin1.weight", "time_emb.lin2.bias", "class_emb.embedding.weight",
             "class_emb.lin2.weight", "latent_proj.weight", "time_projections.0.weight", "time_projections.2.bias",
             "attention_layers.1.in_proj_weight", "attention_layers.1.in_proj_bias", "attention_layers.3.out_proj.weight",
             "layers.0.0.0.weight", "layers.1.0.1.weight", "layers.2.1.bias", "layers.3.2.weight", "final_time_proj.weight",
             "final_class_proj.bias", "final_norm.weight", "final.weight", "final.bias"]


def load(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind != "U" else v) for k, v in np.load(os.path.join(G, name + ".npz")).items()}


def shapes(which):
    return json.load(open(os.path.join(G, "state_dict_shapes.json")))[which]


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, "%s: %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item()
    ref = max(1e-6, b.abs().max().item())
    assert err <= tol * ref, "%s: max err %.3e (ref max %.3e)" % (what, err, ref)


def gslice(t, n=64):
    return t.detach().reshape(-1)[:n]


@pytest.fixture(scope="module")
def ae_sd():
    return pinit.procedural_state_dict(shapes("autoencoder"), 42)


@pytest.fixture(scope="module")
def unet_sd():
    return pinit.procedural_state_dict(shapes("unet"), 44)


def test_unet_train_mode_dropout_semantics(unet_sd):
    """nn.Dropout(.3) of layers[i][0] and the attention dropout of nn.MultiheadAttention on the length-1 sequence
    (v2:514, 522, 550): masks captured from the reference's own dropout calls."""
    f = load("unet_train")
    sd = O._params(unet_sd)
    bm = [f["block_mask%d" % i] for i in range(4)]
    hm = [f["head_mask%d" % i] for i in range(4)]
    out = O.unet_forward(sd, f["x"], f["t"], f["c"], bm, hm)
    close(out, f["eps_train"], 2e-5, "train-mode eps")
    (out * f["dout"]).sum().backward()
    none = set(str(n) for n in f["none_grad_names"])
    tot = 0.0
    for k, v in sd.items():
        if not v.requires_grad:
            continue
        if k in none:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        assert v.grad is not None, k
        tot += float((v.grad.double() ** 2).sum())
    close(math.sqrt(tot), f["grad_norm"], 2e-5, "grad norm")
    for k in UNET_KEYS:
        close(gslice(sd[k].grad), f["grad__" + k.replace(".", "__")], 1e-4, "grad " + k)


def test_classifier_train_mode(ae_sd):
    f = load("classify_train")
    close(O.classify(ae_sd, f["z"], f["mask1"], f["mask2"]), f["logits"], 1e-5, "train-mode logits")


def test_ddpm_loop_gradients(ae_sd, unet_sd):
    """gradients and pre-clip norms of the reference's own train_conditional_diffusion, both iterations."""
    f, gr = load("ddpm_loop"), load("ddpm_loop_grads")
    sd, opt_state = unet_sd, None
    for i in range(2):
        r = O.ddpm_step(ae_sd, sd, f["data%d" % i], f["labels%d" % i], f["reparam_eps%d" % i], f["t%d" % i], f["eps%d" % i],
                        lr=1e-3, opt_state=opt_state)
        close(r["grad_norm"], gr["grad_norm%d" % i], 5e-5, "pre-clip grad norm, step %d" % i)
        for k in UNET_KEYS:
            g = r["grads"][k]
            ref = gr["grad%d__%s" % (i, k.replace(".", "__"))]
            close(gslice(g) if g is not None else torch.zeros_like(ref), ref, 2e-4, "step %d grad %s" % (i, k))
        sd, opt_state = r["new_unet_sd"], r["opt"].state_dict()


AE_KEYS = ["encoder.initial_conv.0.weight", "encoder.down2.0.bias", "encoder.res1.conv1.weight", "encoder.res2.ln2.weight",
           "encoder.res3.ca.conv_du.0.weight", "encoder.res1.sa.conv.weight", "encoder.fc_mu.0.weight",
           "encoder.fc_logvar.3.bias", "decoder.fc.3.weight", "decoder.fc.4.weight", "decoder.res3.conv2.weight",
           "decoder.up3.0.weight", "decoder.up2.1.weight", "decoder.up1.0.bias", "decoder.final_conv.0.weight",
           "decoder.final_conv.3.weight", "decoder.final_conv.3.bias"]
D_GRAD_KEYS = ["model.0.weight", "model.2.weight", "model.3.weight", "model.6.bias", "model.8.bias", "model.11.weight", "model.11.bias"]
# conv biases in front of a BatchNorm: the true gradient is zero, what is left is rounding noise of either implementation
D_NOISE_KEYS = ("model.8.bias",)


def test_vae_gan_loop_gradients(ae_sd):
    """gradients, pre-clip norm, lr and beta1 of the reference's own train_autoencoder (epoch 0), both iterations."""
    f, gr = load("vae_gan_loop"), load("vae_gan_loop_grads")
    d_sd = pinit.procedural_state_dict(shapes("discriminator"), 43)
    vgg_sd = {"feature_extractor." + k: v for k, v in pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7).items()}
    a_sd, aopt, dopt = ae_sd, None, None
    for i in range(2):
        r = O.vae_gan_step(a_sd, d_sd, vgg_sd, f["data%d" % i], f["labels%d" % i], f["reparam_eps%d" % i], epoch=0, num_epochs=1,
                           lr=float(gr["lr%d" % i]), ae_beta1=float(gr["beta1_%d" % i]), ae_opt_state=aopt, d_opt_state=dopt)
        close(r["grad_norm"], gr["grad_norm%d" % i], 2e-4, "pre-clip grad norm, step %d" % i)
        for k in AE_KEYS:
            close(gslice(r["ae_grads"][k]), gr["ae_grad%d__%s" % (i, k.replace(".", "__"))], 5e-4, "step %d AE grad %s" % (i, k))
        for k in D_GRAD_KEYS:
            ref = gr["d_grad%d__%s" % (i, k.replace(".", "__"))]
            if k in D_NOISE_KEYS:
                assert float(gslice(r["d_grads"][k]).abs().max()) < 1e-6 and float(ref.abs().max()) < 1e-6, k
            else:
                close(gslice(r["d_grads"][k]), ref, 5e-4, "step %d D grad %s" % (i, k))
        a_sd, d_sd = r["new_ae_sd"], r["new_d_sd"]
        aopt, dopt = r["opt"].state_dict(), r["d_opt"].state_dict()


def _balanced_total(e, E, recon, percep, kl, cls, cen, adv, loss_phase, kl_weight_at, lam=(0.1, 0.05, 0.4, 0.2)):
    """v2:1107-1125 from the logged terms."""
    l_cls, l_cen, l_vgg, l_gan = lam
    klf, clsf, cenf = loss_phase(e)
    klw = kl_weight_at(e, E)
    if recon > 1e-8:
        p_scale = min(1.0, recon / (percep + 1e-8))
        k_scale = min(1.0, recon / (kl + 1e-8)) if kl > 0 else 1.0
        g_scale = min(1.0, recon / (adv + 1e-8))
    else:
        p_scale = k_scale = g_scale = 1.0
    return recon + l_vgg * p_scale * percep + klw * k_scale * klf * kl + l_cls * clsf * cls + l_cen * cenf * cen + l_gan * g_scale * adv


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_epoch_gating_over_165_epochs(which):
    """loss_phase / kl_weight / balancing arithmetic and the centre-EMA gate against 165 epochs of the reference's own
    train_autoencoder (one batch per epoch): the logged total must equal the formula applied to the logged terms."""
    f = load("vae_gan_gating")
    E = int(f["num_epochs"])
    if which == "oracle":
        lp, kw = O.loss_phase, O.kl_weight_at
    else:
        from oxford_102_flower_gan_vae_latent_diffusion_amd import steps
        lp, kw = steps.loss_phase, steps.kl_weight_at
    h = {k[5:]: f[k].double() for k in f if k.startswith("hist_")}
    for e in range(E):
        klf, clsf, cenf = lp(e)
        # a term is logged as 0 exactly while its factor is 0 (v2:1092-1094)
        assert (float(h["kl"][e]) > 0) == (klf > 0), "kl gate at epoch %d" % e
        assert (float(h["class"][e]) > 0) == (clsf > 0), "class gate at epoch %d" % e
        assert (float(h["center"][e]) > 0) == (cenf > 0), "centre gate at epoch %d" % e
        tot = _balanced_total(e, E, *(float(h[k][e]) for k in ("recon", "perceptual", "kl", "class", "center", "gan")), lp, kw)
        assert abs(tot - float(h["total"][e])) <= 2e-6 * max(1.0, abs(tot)), "total at epoch %d: %.7f vs %.7f" % (e, tot, float(h["total"][e]))
    # class centres move only once `epoch >= 60 and center_factor > 0` (v2:1133): first change is visible before step 161
    cs = f["centers_abs_sum_before_step"].double()
    changed = (cs[1:] != cs[:-1]).nonzero().flatten()
    first = int(changed[0]) + 1
    want = next(e for e in range(E) if e >= 60 and lp(e)[2] > 0) + 1
    assert first == want, (first, want)


def test_onecycle_matches_the_reference_run():
    """lr and beta1 the reference's OneCycleLR had set at each of the 165 optimizer steps."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import schedules
    f = load("vae_gan_gating")

    class Opt:
        def set_lr(self, v):
            self.lr = v

        def set_beta1(self, v):
            self.b1 = v

    o = Opt()
    sch = schedules.OneCycle(o, 1e-4, total_steps=int(f["num_epochs"]))
    for e in range(int(f["num_epochs"])):
        assert abs(o.lr - float(f["lr"][e])) <= 1e-12 + 1e-9 * abs(o.lr), e
        assert abs(o.b1 - float(f["beta1"][e])) <= 1e-9, e
        sch.step()
