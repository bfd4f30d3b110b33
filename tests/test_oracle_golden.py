"""Pins the CPU oracle (oracle/v2_oracle.py) to the golden vectors generated from the reference
itself (tests/golden/make_golden.py). CPU only; fp32 on both sides, tolerance = fp32 re-association."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import procedural_init as pinit
from oracle import v2_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name + ".npz")).items()}


def shapes(which):
    return json.load(open(os.path.join(G, "state_dict_shapes.json")))[which]


def close(a, b, tol=2e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, "%s: max err %.3e (ref max %.3e)" % (what, err, ref)


@pytest.fixture(scope="module")
def ae_sd():
    return pinit.procedural_state_dict(shapes("autoencoder"), 42)


@pytest.fixture(scope="module")
def unet_sd():
    return pinit.procedural_state_dict(shapes("unet"), 44)


@pytest.fixture(scope="module")
def d_sd():
    return pinit.procedural_state_dict(shapes("discriminator"), 43)


def test_primitives(ae_sd):
    f = load("primitives")
    x, p = f["x"], "encoder.res1."
    close(O.swish(x), f["swish"], 1e-6, "swish")
    close(O.ca_layer(ae_sd, p + "ca.", x), f["ca"], 1e-5, "ca")
    close(O.spatial_attention(ae_sd, p + "sa.", x), f["sa"], 1e-5, "sa")
    close(O.layernorm2d(ae_sd, p + "ln1.", x), f["ln2d"], 1e-5, "ln2d")
    close(O.residual_block(ae_sd, p, x), f["resblock"], 2e-5, "resblock")


def test_autoencoder_pieces(ae_sd):
    f = load("autoencoder")
    mu, logvar, feats = O.encoder_forward(ae_sd, "encoder.", f["data"], return_features=True)
    close(mu, f["mu"], 5e-5, "mu")
    close(logvar, f["logvar"], 5e-5, "logvar")
    close(feats[3], f["feat3"], 5e-5, "feat3")
    close(torch.stack([x.abs().mean() for x in feats]), f["feat_absmean"], 1e-5, "feat absmean")
    close(O.clamp_logvar(logvar), f["logvar_clamped"], 5e-5, "clamped")
    close(O.reparameterize(f["mu"], f["logvar"], f["eps777"]), f["z777"], 1e-6, "reparameterize (reference RNG draw)")
    close(O.reparameterize(f["mu"], f["logvar"], f["eps"]), f["z"], 1e-6, "z")
    close(O.decoder_forward(ae_sd, "decoder.", f["z"]), f["recon"], 5e-5, "recon")
    close(O.kl_divergence(f["mu"] * 6, f["logvar"] * 3), f["kl"], 1e-5, "kl")
    close(O.classify(ae_sd, f["z"]), f["logits"], 2e-5, "classify")
    centers = ae_sd["class_centers"]
    close(O.center_loss(centers, f["z"], f["labels"]), f["center_loss"], 1e-5, "center loss")
    close(centers[f["labels"]], f["centers_before"], 0, "centers")
    close(O.update_centers(centers, f["z"], f["labels"])[f["labels"]], f["centers_after"], 1e-6, "update_centers")
    close(O.euclidean_distance_loss(f["recon"], f["data"]), f["recon_loss"], 1e-6, "euclid mean")
    close(O.euclidean_distance_loss(f["recon"], f["data"], "none"), f["recon_norms"], 1e-6, "euclid none")


def test_unet_and_diffusion(unet_sd):
    f = load("unet")
    close(O.unet_forward(unet_sd, f["x"], f["t"], f["c"]), f["eps_full"], 3e-5, "eps (per-sample t)")
    close(O.unet_forward(unet_sd, f["x"], torch.tensor([437]), f["c"]), f["eps_shared"], 3e-5, "eps (shared t row)")
    close(O.unet_forward(unet_sd, f["x"], f["t"], None), f["eps_nocls"], 3e-5, "eps (no class)")
    sched = O.schedule()
    close(sched[0], f["beta"], 0, "beta")
    close(sched[2], f["alpha_bar"], 0, "alpha_bar")
    close(O.q_sample(sched[2], f["x"], f["t"], f["noise_q"]), f["xt"], 1e-6, "q_sample")
    x = f["traj"][0]
    for i, tt in enumerate(reversed(range(20))):
        x = O.p_sample(unet_sd, sched, x, tt, f["traj_c"], f["traj_noise"][i])
        close(x, f["traj"][i + 1], 1e-4, "p_sample t=%d" % tt)
    loss, _ = O.ddpm_loss(unet_sd, sched, f["x"], f["c"], f["loss_t"], f["loss_eps"])
    close(loss, f["loss"], 2e-5, "ddpm loss (reference RNG draws)")
    # eps-MSE parity criterion of the north star (<= 1e-4) holds with a wide margin for the oracle itself
    mse = ((O.unet_forward(unet_sd, f["x"], f["t"], f["c"]) - f["eps_full"]) ** 2).mean().item()
    assert mse < 1e-9


def test_discriminator(d_sd):
    f = load("discriminator")
    out, st = O.discriminator_forward(d_sd, f["x"], training=True)
    close(out, f["d_train"], 2e-5, "D train")
    for k, v in st.items():
        close(v, f[k.replace(".", "__")], 1e-5, k)
    sd2 = dict(d_sd)
    sd2.update(st)
    out, _ = O.discriminator_forward(sd2, f["x"], training=False)
    close(out, f["d_eval"], 2e-5, "D eval")


def test_vgg():
    f = load("vgg")
    sd = pinit.procedural_state_dict(pinit.vgg16_features16_shapes("feature_extractor."), 7)
    sd0 = pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7)
    for k, v in sd0.items():  # key-prefix independence of the procedural init is NOT assumed
        sd["feature_extractor." + k] = v
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    close(O.vgg_features(sd, "feature_extractor.", (f["x"] - mean) / std), f["feat_x"], 3e-5, "vgg features")
    close(O.vgg_perceptual_loss(sd, f["x"], f["y"]), f["loss"], 3e-5, "vgg loss")


def test_ddpm_loop_matches_reference_loop(ae_sd, unet_sd):
    """Two iterations of the reference's own train_conditional_diffusion (v2:1254-1264)."""
    f = load("ddpm_loop")
    sd, opt_state, losses = unet_sd, None, []
    for i in range(2):
        r = O.ddpm_step(ae_sd, sd, f["data%d" % i], f["labels%d" % i], f["reparam_eps%d" % i], f["t%d" % i],
                        f["eps%d" % i], lr=1e-3, opt_state=opt_state)
        sd, opt_state = r["new_unet_sd"], r["opt"].state_dict()
        losses.append(r["loss"].item())
    close(np.mean(losses), f["epoch_loss"], 2e-5, "epoch mean loss")
    for k in [k for k in f if "__" in k or k == "residual_weight"]:
        key = k.replace("__", ".")
        close(sd[key].reshape(-1)[:64], f[k], 2e-4, "post-step " + key)


def test_vae_gan_loop_matches_reference_loop(ae_sd, d_sd):
    """Two iterations of the reference's own train_autoencoder (v2:1060-1134), epoch 0 phase, OneCycleLR."""
    f = load("vae_gan_loop")
    vgg_sd = {"feature_extractor." + k: v for k, v in pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7).items()}
    # OneCycleLR(max_lr=1e-4, total_steps=2, pct_start=.3, div 25, final_div 1000), stepped per batch (v2:1031-1037,1130)
    dummy = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = torch.optim.lr_scheduler.OneCycleLR(dummy, max_lr=1e-4, total_steps=2, pct_start=0.3, div_factor=25,
                                              final_div_factor=1000)
    ae, dd, ae_opt, d_opt = ae_sd, d_sd, None, None
    hist = {k: [] for k in ("total", "recon", "perceptual", "kl", "cls", "center", "adv")}
    for i in range(2):
        lr = dummy.param_groups[0]["lr"]
        if ae_opt is not None:
            for gph in ae_opt["param_groups"]:
                gph["lr"] = lr
        r = O.vae_gan_step(ae, dd, vgg_sd, f["data%d" % i], f["labels%d" % i], f["reparam_eps%d" % i], epoch=0,
                           num_epochs=1, lr=lr, ae_opt_state=ae_opt, d_opt_state=d_opt)
        ae, dd = r["new_ae_sd"], r["new_d_sd"]
        ae_opt, d_opt = r["opt"].state_dict(), r["d_opt"].state_dict()
        for k in hist:
            hist[k].append(r["losses"][k].item())
        dummy.step()
        if i == 0:
            sch.step()
    for k, hk in (("total", "total"), ("recon", "recon"), ("perceptual", "perceptual"), ("kl", "kl"), ("cls", "class"),
                  ("center", "center"), ("adv", "gan")):
        close(np.mean(hist[k]), f["hist_" + hk], 3e-5, "loss history " + hk)
    for k in [k for k in f if k.startswith("ae__")]:
        key = k[4:].replace("__", ".")
        close(ae[key].reshape(-1)[:64], f[k], 3e-4, "post-step AE " + key)
    for k in [k for k in f if k.startswith("d__")]:
        key = k[3:].replace("__", ".")
        close(dd[key].reshape(-1)[:64], f[k], 3e-4, "post-step D " + key)


def test_dead_conv_unet_blocks():
    """UNetAttentionBlock / UNetResidualBlock / SwitchSequential (v2:434-498) instantiated directly in the reference."""
    f = load("dead_blocks")
    sh = json.load(open(os.path.join(G, "state_dict_shapes.json")))["dead_blocks"]
    a64 = pinit.procedural_state_dict(sh["attn64"], 45)
    r1 = pinit.procedural_state_dict(sh["res64_128"], 46)
    r2 = pinit.procedural_state_dict(sh["res128"], 47)
    a128 = pinit.procedural_state_dict(sh["attn128"], 48)
    close(O.unet_attention_block(a64, "", f["x"]), f["attn"], 3e-5, "attention block")
    close(O.unet_residual_block(r1, "", f["x"], f["t"], f["c"]), f["res"], 3e-5, "residual block")
    close(O.unet_residual_block(r1, "", f["x"], f["t"], None), f["res_noc"], 3e-5, "residual block (no class)")
    h = O.unet_residual_block(r1, "", f["x"], f["t"], f["c"])
    h = O.unet_attention_block(a128, "", h)
    h = O.unet_residual_block(r2, "", h, f["t"], f["c"])
    close(h, f["seq"], 5e-5, "SwitchSequential")
