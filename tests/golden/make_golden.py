"""Generate the golden fixtures that pin the CPU oracle to the REFERENCE itself.

Runs ONLY in the build container (the reference tree does not exist on the GPU box):
    python tests/golden/make_golden.py
It imports /root/reference/v2/model_train_test.py on CPU (torchvision / imageio are not
installed here and are replaced by inert stand-ins, exactly as SURVEY.md §8c records; the
reference's own arithmetic is untouched), builds the reference modules, loads the seeded
procedural weights of oracle/procedural_init.py, and runs
  * every module of the v2 surface (Encoder/Decoder/SimpleAutoencoder pieces, ConditionalUNet,
    ConditionalDenoiseDiffusion, Discriminator64, VGGPerceptualLoss),
  * the reference's own training loops train_conditional_diffusion (v2:1241) and
    train_autoencoder (v2:1014) for two iterations on a two-batch list "loader".
Only inputs/outputs are stored (tests/golden/*.npz) — no reference source.
"""
import importlib.util
import json
import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import procedural_init as pinit  # noqa: E402

REF = "/root/reference/v2/model_train_test.py"


def load_reference():
    sys.dont_write_bytecode = True
    for m in ["torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.models", "imageio"]:
        sys.modules.setdefault(m, MagicMock())
    spec = importlib.util.spec_from_file_location("v2ref", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # NB: re-seeds torch to 42 at import (v2:17)
    return mod


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def g(seed):
    return torch.Generator().manual_seed(seed)


def slices(sd, keys, n=64):
    """first n values of selected tensors (post-step parameter / gradient slices)."""
    return {k.replace(".", "__"): sd[k].detach().reshape(-1)[:n].clone() for k in keys}


AE_KEYS = ["encoder.initial_conv.0.weight", "encoder.down2.0.bias", "encoder.res1.conv1.weight", "encoder.res2.ln2.weight",
           "encoder.res3.ca.conv_du.0.weight", "encoder.res1.sa.conv.weight", "encoder.fc_mu.0.weight",
           "encoder.fc_logvar.3.bias", "decoder.fc.3.weight", "decoder.fc.4.weight", "decoder.res3.conv2.weight",
           "decoder.up3.0.weight", "decoder.up2.1.weight", "decoder.up1.0.bias", "decoder.final_conv.0.weight",
           "decoder.final_conv.3.weight", "decoder.final_conv.3.bias"]
UNET_KEYS = ["residual_weight", "time_emb.lin1.weight", "time_emb.lin2.bias", "class_emb.embedding.weight",
             "class_emb.lin2.weight", "latent_proj.weight", "time_projections.0.weight", "time_projections.2.bias",
             "attention_layers.1.in_proj_weight", "attention_layers.1.in_proj_bias", "attention_layers.3.out_proj.weight",
             "layers.0.0.0.weight", "layers.1.0.1.weight", "layers.2.1.bias", "layers.3.2.weight", "final_time_proj.weight",
             "final_class_proj.bias", "final_norm.weight", "final.weight", "final.bias"]
D_KEYS = ["model.0.weight", "model.2.weight", "model.3.weight", "model.6.bias", "model.8.bias", "model.11.weight",
          "model.11.bias", "model.3.running_mean", "model.9.running_var"]


def main():
    mod = load_reference()
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ reference modules + procedural weights
    ae = mod.SimpleAutoencoder()
    unet = mod.ConditionalUNet()
    disc = mod.Discriminator64()
    shapes = {"autoencoder": {k: list(v.shape) for k, v in ae.state_dict().items()},
              "unet": {k: list(v.shape) for k, v in unet.state_dict().items()},
              "discriminator": {k: list(v.shape) for k, v in disc.state_dict().items()}}
    json.dump(shapes, open(os.path.join(HERE, "state_dict_shapes.json"), "w"), indent=0)
    ae.load_state_dict(pinit.procedural_state_dict(ae.state_dict(), 42))
    unet.load_state_dict(pinit.procedural_state_dict(unet.state_dict(), 44))
    disc.load_state_dict(pinit.procedural_state_dict(disc.state_dict(), 43))
    ae.eval()
    unet.eval()

    # ------------------------------------------------------------------ primitives (v2:48-178)
    x128 = torch.randn(2, 128, 8, 8, generator=g(1)) * 1.5 + 0.2
    rb = ae.encoder.res1
    with torch.no_grad():
        npz("primitives", x=x128, swish=mod.Swish()(x128), ca=rb.ca(x128), sa=rb.sa(x128), ln2d=rb.ln1(x128),
            resblock=rb(x128))

    # ------------------------------------------------------------------ autoencoder pieces (v2:181-393)
    data = torch.rand(2, 3, 64, 64, generator=g(1234))
    eps = torch.randn(2, 256, generator=g(1235))
    labels = torch.tensor([3, 77])
    with torch.no_grad():
        mu, logvar = ae.encoder(data)
        feats = ae.encoder.skip_features
        mu_c, logvar_c = ae.encode_with_params(data)
        torch.manual_seed(777)
        z_ref = ae.reparameterize(mu, logvar)          # draws randn_like -> replicate below
        torch.manual_seed(777)
        eps777 = torch.randn_like(mu)
        z = mu + eps * torch.exp(0.5 * logvar_c)
        recon = ae.decode(z)
        centers = ae.class_centers.clone()
        closs = ae.compute_center_loss(z, labels)
        ae.update_centers(z, labels, momentum=0.9)
        npz("autoencoder", data=data, eps=eps, labels=labels, mu=mu, logvar=logvar, logvar_clamped=logvar_c,
            eps777=eps777, z777=z_ref, z=z, recon=recon,
            feat_absmean=torch.stack([f.abs().mean() for f in feats]), feat_mean=torch.stack([f.mean() for f in feats]),
            feat3=feats[3], kl=ae.kl_divergence(mu * 6, logvar * 3), logits=ae.classify(z),
            center_loss=closs, centers_before=centers[labels],
            centers_after=ae.class_centers[labels], recon_loss=mod.euclidean_distance_loss(recon, data),
            recon_norms=mod.euclidean_distance_loss(recon, data, reduction="none"))
        ae.class_centers.copy_(centers)

    # ------------------------------------------------------------------ denoiser + diffusion (v2:401-431, 501-607)
    xl = torch.randn(8, 256, generator=g(20))
    t = torch.randint(0, 1000, (8,), generator=g(21))
    t[0], t[1] = 0, 999
    c = torch.randint(0, 102, (8,), generator=g(22))
    diff = mod.ConditionalDenoiseDiffusion(unet, n_steps=1000, device=torch.device("cpu"))
    with torch.no_grad():
        e_full = unet(xl, t, c)
        e_shared = unet(xl, torch.tensor([437]), c)        # sampling-style: one t row broadcast (v2:581-582)
        e_nocls = unet(xl, t, None)
        noise_q = torch.randn(8, 256, generator=g(23))
        xt = diff.q_sample(xl, t, noise_q)
        # p_sample trajectory: 4 latents, t = 19..0, explicit noise recovered by re-seeding
        x = torch.randn(4, 256, generator=g(24))
        cc = torch.tensor([0, 5, 50, 101])
        traj, noises = [x.clone()], []
        for tt in reversed(range(20)):
            torch.manual_seed(1000 + tt)
            nz = torch.randn_like(x)
            torch.manual_seed(1000 + tt)
            x = diff.p_sample(x, tt, cc)
            traj.append(x.clone())
            noises.append(nz)
        # loss with the reference's own RNG draws (randint then randn_like, v2:602-603)
        torch.manual_seed(31337)
        t_l = torch.randint(0, 1000, (8,), dtype=torch.long)
        eps_l = torch.randn_like(xl)
        torch.manual_seed(31337)
        loss_ref = diff.loss(xl, c)
    npz("unet", x=xl, t=t, c=c, eps_full=e_full, eps_shared=e_shared, eps_nocls=e_nocls, noise_q=noise_q, xt=xt,
        beta=diff.beta, alpha_bar=diff.alpha_bar, traj=torch.stack(traj), traj_noise=torch.stack(noises), traj_c=cc,
        loss_t=t_l, loss_eps=eps_l, loss=loss_ref)

    # ------------------------------------------------------------------ the reference's own latent-DDPM loop (v2:1241-1285)
    B = 8
    batches = []
    for i in range(2):
        batches.append((torch.rand(B, 3, 64, 64, generator=g(3000 + i)), torch.randint(0, 102, (B,), generator=g(3100 + i))))
    mod.class_names = []  # visualisation loop iterates min(len(class_names), 2) -> none
    unet_t = mod.ConditionalUNet()
    unet_t.load_state_dict(pinit.procedural_state_dict(unet_t.state_dict(), 44))
    unet_t.eval()  # dropout off; gradients still flow
    tmp = tempfile.mkdtemp()
    seed = 4242
    torch.manual_seed(seed)
    draws = []
    for i in range(2):  # same draw order as v2:1259 (randn_like), v2:602 (randint), v2:603 (randn_like)
        draws.append((torch.randn(B, 256), torch.randint(0, 1000, (B,), dtype=torch.long), torch.randn(B, 256)))
    torch.manual_seed(seed)
    unet_t, _, hist = mod.train_conditional_diffusion(ae, unet_t, batches, num_epochs=1, lr=1e-3, visualize_every=1000,
                                                      save_dir=tmp, device=torch.device("cpu"))
    sd_after = unet_t.state_dict()
    npz("ddpm_loop", epoch_loss=np.float32(hist[0]),
        **{"data%d" % i: b[0] for i, b in enumerate(batches)}, **{"labels%d" % i: b[1] for i, b in enumerate(batches)},
        **{"reparam_eps%d" % i: d[0] for i, d in enumerate(draws)}, **{"t%d" % i: d[1] for i, d in enumerate(draws)},
        **{"eps%d" % i: d[2] for i, d in enumerate(draws)}, **slices(sd_after, UNET_KEYS))

    # ------------------------------------------------------------------ conv-UNet blocks (dead code in v2, A18)
    attn = mod.UNetAttentionBlock(64, num_heads=4)
    attn.load_state_dict(pinit.procedural_state_dict(attn.state_dict(), 45))
    rbk = mod.UNetResidualBlock(64, 128, d_time=256, dropout_rate=0.2)
    rbk.load_state_dict(pinit.procedural_state_dict(rbk.state_dict(), 46))
    rbk2 = mod.UNetResidualBlock(128, 128, d_time=256)
    rbk2.load_state_dict(pinit.procedural_state_dict(rbk2.state_dict(), 47))
    attn2 = mod.UNetAttentionBlock(128, num_heads=4)
    attn2.load_state_dict(pinit.procedural_state_dict(attn2.state_dict(), 48))
    seq = mod.SwitchSequential(rbk, attn2, rbk2).eval()
    attn.eval()
    xa = torch.randn(2, 64, 8, 8, generator=g(70))
    tt, cc2 = torch.randn(2, 256, generator=g(71)), torch.randn(2, 256, generator=g(72))
    shapes["dead_blocks"] = {"attn64": {k: list(v.shape) for k, v in attn.state_dict().items()},
                             "res64_128": {k: list(v.shape) for k, v in rbk.state_dict().items()},
                             "res128": {k: list(v.shape) for k, v in rbk2.state_dict().items()},
                             "attn128": {k: list(v.shape) for k, v in attn2.state_dict().items()}}
    json.dump(shapes, open(os.path.join(HERE, "state_dict_shapes.json"), "w"), indent=0)
    with torch.no_grad():
        npz("dead_blocks", x=xa, t=tt, c=cc2, attn=attn(xa), res=rbk(xa, tt, cc2), res_noc=rbk(xa, tt, None),
            seq=seq(xa, tt, cc2))

    # ------------------------------------------------------------------ discriminator (v2:983-1007)
    xd = torch.rand(4, 3, 64, 64, generator=g(50))
    disc.train()
    with torch.no_grad():
        d_train = disc(xd)
    sd_d = {k: v.clone() for k, v in disc.state_dict().items()}
    disc.eval()
    with torch.no_grad():
        d_eval = disc(xd)
    npz("discriminator", x=xd, d_train=d_train, d_eval=d_eval,
        **{k.replace(".", "__"): sd_d[k] for k in sd_d if "running" in k or "num_batches" in k})

    # ------------------------------------------------------------------ VGG perceptual (v2:962-981) with synthetic weights
    vgg_sd = pinit.procedural_state_dict(pinit.vgg16_features16_shapes(""), 7)

    def fake_vgg16(pretrained=True):
        layers, cin = [], 3
        for item in [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]:
            if item == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, item, 3, padding=1), nn.ReLU(inplace=True)]
                cin = item
        feats = nn.Sequential(*layers)
        feats.load_state_dict({**feats.state_dict(), **vgg_sd})
        holder = MagicMock()
        holder.features = feats
        return holder

    mod.models.vgg16 = fake_vgg16
    vgg = mod.VGGPerceptualLoss(torch.device("cpu"))
    a, b = torch.rand(2, 3, 64, 64, generator=g(60)), torch.rand(2, 3, 64, 64, generator=g(61))
    with torch.no_grad():
        npz("vgg", x=a, y=b, loss=vgg(a, b), feat_x=vgg.feature_extractor((a - vgg.mean) / vgg.std))

    # ------------------------------------------------------------------ the reference's own VAE-GAN loop (v2:1014-1193)
    Bv = 4
    vb = [(torch.rand(Bv, 3, 64, 64, generator=g(5000 + i)), torch.randint(0, 102, (Bv,), generator=g(5100 + i)))
          for i in range(2)]
    veps = [torch.randn(Bv, 256, generator=g(5200 + i)) for i in range(2)]

    class SeededD(mod.Discriminator64):
        def __init__(self, in_channels=3):
            super().__init__(in_channels)
            self.load_state_dict(pinit.procedural_state_dict(self.state_dict(), 43))

    mod.Discriminator64 = SeededD
    mod.visualize_reconstructions = lambda *a_, **k_: None
    mod.visualize_latent_space = lambda *a_, **k_: None
    ae_t = mod.SimpleAutoencoder()
    ae_t.load_state_dict(pinit.procedural_state_dict(ae_t.state_dict(), 42))
    it = iter(veps)

    def reparam_explicit(mu_, logvar_):  # explicit-noise form of v2:339-343 (itself pinned above via z777)
        return mu_ + next(it) * torch.exp(0.5 * torch.clamp(logvar_, min=-2.0, max=10.0))

    ae_t.reparameterize = reparam_explicit
    ae_t, d_t, lh = mod.train_autoencoder(ae_t, vb, num_epochs=1, lr=1e-4, visualize_every=1000, save_dir=tmp)
    npz("vae_gan_loop", **{"data%d" % i: b_[0] for i, b_ in enumerate(vb)}, **{"labels%d" % i: b_[1] for i, b_ in enumerate(vb)},
        **{"reparam_eps%d" % i: e_ for i, e_ in enumerate(veps)},
        **{"hist_" + k: np.float32(v[0]) for k, v in lh.items()},
        **{"ae__" + k: v for k, v in slices(ae_t.state_dict(), AE_KEYS).items()},
        **{"d__" + k: v for k, v in slices(d_t.state_dict(), D_KEYS).items()})
    print("done")


if __name__ == "__main__":
    main()
