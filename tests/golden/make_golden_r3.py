"""Round-3 golden fixtures: the VAE-GAN step's gradients from the REFERENCE ITSELF IN FP64 (missing #4: the only
fp64 truth the config-2 tests had was the oracle's). Same loader and stand-ins as make_golden.py / make_golden_r2.py; runs only in
the build container:
    python tests/golden/make_golden_r3.py

  vae_gan_grads64.npz       iteration 0 of the reference's train_autoencoder (v2:1060-1134) on the inputs of vae_gan_loop.npz with
                            every module in double (procedural weights, seeds 42 / 43 / 7): pre-clip gradient norm, autoencoder and
                            discriminator gradient slices, the loss terms. The fp64 run of the reference's own code is the truth the
                            HIP gradients are held to at a FIXED tolerance.
  vae_gan_init_grads64.npz  the same with the reference's OWN initialisation (SimpleAutoencoder._init_weights v2:326-337, torch's default
                            Conv2d / BatchNorm init for Discriminator64) under torch.manual_seed(1234 / 1235), built in fp32 and
                            then cast to double - a better-conditioned point than the procedural weights. The weights are 277 MB and are
                            NOT stored: this package's modules construct bit-identical weights under the same seeds (checked here
                            against the reference classes; per-tensor checksums are stored so that the GPU test can check it again).
Only inputs / outputs are stored - no reference source."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import AE_KEYS, D_KEYS, g, load_reference, npz, pinit  # noqa: E402
from make_golden_r2 import fake_vgg16_factory, gslices  # noqa: E402

AE_SEED, D_SEED = 1234, 1235
D_GRAD_KEYS = [k for k in D_KEYS if "running" not in k and "num_batches" not in k]


def checksums(sd, keys):
    return {k.replace(".", "__"): np.float64(sd[k].double().abs().sum().item()) for k in keys}


def run_step0(mod, ae_sd32, d_sd32, data, labels, eps, tmp, dtype=torch.float64):
    """iteration 0 of the reference's train_autoencoder with everything in double; the weights are the given fp32 values."""
    # (the procedural VGG weights are DRAWN here, under the fp32 default - torch.randn draws different numbers in double - and
    # only stored in double modules below)
    mod.models.vgg16 = fake_vgg16_factory()
    torch.set_default_dtype(dtype)
    cast = lambda v: v.to(dtype) if v.is_floating_point() else v
    try:
        d_names = [n for n, _ in mod._RefD().named_parameters()]

        class GivenD(mod._RefD):
            def __init__(self, in_channels=3):
                super().__init__(in_channels)
                self.load_state_dict({k: cast(v) for k, v in d_sd32.items()})

        mod.Discriminator64 = GivenD
        mod.visualize_reconstructions = lambda *a_, **k_: None
        mod.visualize_latent_space = lambda *a_, **k_: None
        ae = mod.SimpleAutoencoder()
        ae.load_state_dict({k: cast(v) for k, v in ae_sd32.items()})
        assert next(ae.parameters()).dtype == dtype
        it = iter([eps.to(dtype)])
        ae.reparameterize = lambda mu_, lv_: mu_ + next(it) * torch.exp(0.5 * torch.clamp(lv_, min=-2.0, max=10.0))
        rec = {}
        real_clip = torch.nn.utils.clip_grad_norm_
        ae_ids = {id(p) for p in ae.parameters()}

        def clip_tap(params, max_norm, *a, **k):
            params = list(params)
            if "norm" not in rec:
                rec["norm"] = float(real_clip(params, 1e30))
                rec["ae"] = gslices(ae.named_parameters(), AE_KEYS)
            return real_clip(params, max_norm, *a, **k)

        def pre_hook(opt, args, kwargs):
            first = opt.param_groups[0]["params"][0]
            if id(first) not in ae_ids and "d" not in rec:
                named_d = list(zip(d_names, opt.param_groups[0]["params"]))
                rec["d"] = gslices(named_d, D_GRAD_KEYS)

        torch.nn.utils.clip_grad_norm_ = clip_tap
        real_save = torch.save
        torch.save = lambda *a_, **k_: None
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        h = register_optimizer_step_pre_hook(pre_hook)
        try:
            _, _, lh = mod.train_autoencoder(ae, [(data.to(dtype), labels)], num_epochs=1, lr=1e-4, visualize_every=100000, save_dir=tmp)
        finally:
            h.remove()
            torch.save = real_save
            torch.nn.utils.clip_grad_norm_ = real_clip
        return rec, lh
    finally:
        torch.set_default_dtype(torch.float32)


def main():
    mod = load_reference()
    mod._RefD = mod.Discriminator64
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp()
    Bv = 4
    data, labels = torch.rand(Bv, 3, 64, 64, generator=g(5000)), torch.randint(0, 102, (Bv,), generator=g(5100))
    eps = torch.randn(Bv, 256, generator=g(5200))

    def store(name, rec, lh, extra=None):
        f = {"grad_norm": np.float64(rec["norm"]), "data": data, "labels": labels, "reparam_eps": eps}
        f.update({"ae_grad__" + k: v.double() for k, v in rec["ae"].items()})
        f.update({"d_grad__" + k: v.double() for k, v in rec["d"].items()})
        f.update({"hist_" + k: np.float64(v[0]) for k, v in lh.items()})
        f.update(extra or {})
        npz(name, **f)

    # ---- procedural weights (the inputs of vae_gan_loop.npz, iteration 0)
    ae_sd = pinit.procedural_state_dict(mod.SimpleAutoencoder().state_dict(), 42)
    d_sd = pinit.procedural_state_dict(mod._RefD().state_dict(), 43)
    torch.manual_seed(42)
    rec, lh = run_step0(mod, ae_sd, d_sd, data, labels, eps, tmp)
    store("vae_gan_grads64", rec, lh)

    # ---- the reference's own initialisation under a seed (fp32 construction)
    torch.manual_seed(AE_SEED)
    ae_ref = mod.SimpleAutoencoder()
    torch.manual_seed(D_SEED)
    d_ref = mod._RefD()
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
    torch.manual_seed(AE_SEED)
    ae_own = M.SimpleAutoencoder()
    torch.manual_seed(D_SEED)
    d_own = M.Discriminator64()
    for a, b, what in ((ae_ref, ae_own, "autoencoder"), (d_ref, d_own, "discriminator")):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb), what
        for k in sa:
            assert torch.equal(sa[k], sb[k]), "%s: %s differs between the reference's and this package's construction" % (what, k)
    ae_sd, d_sd = ae_ref.state_dict(), d_ref.state_dict()
    torch.manual_seed(43)
    rec, lh = run_step0(mod, ae_sd, d_sd, data, labels, eps, tmp)
    extra = {"ae_seed": np.int64(AE_SEED), "d_seed": np.int64(D_SEED)}
    extra.update({"ae_sum__" + k: v for k, v in checksums(ae_sd, AE_KEYS).items()})
    extra.update({"d_sum__" + k: v for k, v in checksums(d_sd, D_GRAD_KEYS).items()})
    store("vae_gan_init_grads64", rec, lh, extra)
    print("done")


if __name__ == "__main__":
    main()
