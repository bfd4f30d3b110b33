"""Round-2 golden fixtures: train-mode (dropout) semantics, GRADIENTS of the reference's own training loops, and the
epoch gating of train_autoencoder - all taken from the REFERENCE itself (same loader and stand-ins as make_golden.py;
runs only in the build container):
    python tests/golden/make_golden_r2.py

  unet_train.npz          ConditionalUNet in train(): the masks of every dropout call (nn.Dropout v2:522 and the attention
                          dropout inside nn.MultiheadAttention v2:514/550) captured by wrapping torch.nn.functional.dropout,
                          the output, and parameter-gradient slices of sum(out * dout)
  classify_train.npz      SimpleAutoencoder.classify in train() (v2:311-321): captured masks + logits
  ddpm_loop_grads.npz     the two iterations of train_conditional_diffusion of ddpm_loop.npz again, with every
                          clip_grad_norm_ call (v2:1263) intercepted: pre-clip total norm + gradient slices per step
  vae_gan_loop_grads.npz  the two iterations of train_autoencoder of vae_gan_loop.npz again: pre-clip norm + autoencoder
                          gradient slices (v2:1127-1128), discriminator gradient slices at d_optimizer.step (v2:1102), and the
                          lr / beta1 OneCycleLR had set for each step
  vae_gan_gating.npz      165 epochs of train_autoencoder (one batch of 2 images per epoch, num_epochs=165): the seven
                          per-epoch loss histories, lr / beta1 per step and the class-centre magnitude per step - pins
                          loss_phase / kl_weight / balancing (v2:1055, v2:1072-1087, v2:1107-1125) and the centre-EMA gate
                          (v2:1133) across the 40 / 80 / 160 epoch boundaries
Only inputs / outputs are stored - no reference source."""
import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import AE_KEYS, D_KEYS, UNET_KEYS, g, load_reference, npz, pinit, slices  # noqa: E402


class DropoutTap:
    """records the scale tensor (0 or 1/(1-p)) of every torch.nn.functional.dropout call while active."""

    def __init__(self):
        self.masks = []
        self._orig = F.dropout

    def __enter__(self):
        orig = self._orig

        def tapped(input, p=0.5, training=True, inplace=False):
            out = orig(input, p, training, False)
            if training and p > 0:
                m = torch.where(out != 0, torch.full_like(out, 1.0 / (1.0 - p)), torch.zeros_like(out))
                # an input element that is exactly 0 hides its mask value; none of the taps below feeds zeros
                assert torch.allclose(input * m, out)
                self.masks.append(m.detach().clone())
            return out

        F.dropout = tapped
        torch.nn.functional.dropout = tapped
        return self

    def __exit__(self, *a):
        F.dropout = self._orig
        torch.nn.functional.dropout = self._orig


def fake_vgg16_factory():
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

    return fake_vgg16


def gslices(named, keys, n=64):
    d = dict(named)
    return {k.replace(".", "__"): (d[k].grad.detach().reshape(-1)[:n].clone() if d[k].grad is not None
                                   else torch.zeros(min(n, d[k].numel()))) for k in keys}


def main():
    mod = load_reference()
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp()

    # ------------------------------------------------------------------ ConditionalUNet, train mode (v2:514, 522, 550)
    unet = mod.ConditionalUNet()
    unet.load_state_dict(pinit.procedural_state_dict(unet.state_dict(), 44))
    unet.train()
    B = 8
    x = torch.randn(B, 256, generator=g(200))
    t = torch.randint(0, 1000, (B,), generator=g(201))
    c = torch.randint(0, 102, (B,), generator=g(202))
    dout = torch.randn(B, 256, generator=g(203))
    torch.manual_seed(9001)
    with DropoutTap() as tap:
        out = unet(x, t, c)
    assert len(tap.masks) == 8, [m.shape for m in tap.masks]  # per layer: block dropout (B,d) then attention dropout (B*8,1,1)
    (out * dout).sum().backward()
    named = list(unet.named_parameters())
    gn = torch.sqrt(sum((p.grad ** 2).sum() for _, p in named if p.grad is not None))
    none_grads = sorted(n for n, p in named if p.grad is None)
    fields = dict(x=x, t=t, c=c, dout=dout, eps_train=out, grad_norm=gn)
    for i in range(4):
        fields["block_mask%d" % i] = tap.masks[2 * i]
        fields["head_mask%d" % i] = tap.masks[2 * i + 1].reshape(B, 8)  # (B*heads,1,1): batch-major, head-minor
    fields.update({"grad__" + k: v for k, v in gslices(named, UNET_KEYS).items()})
    npz("unet_train", none_grad_names=np.array(none_grads), **fields)

    # ------------------------------------------------------------------ classifier head, train mode (v2:311-321)
    ae = mod.SimpleAutoencoder()
    ae.load_state_dict(pinit.procedural_state_dict(ae.state_dict(), 42))
    ae.train()
    z = torch.randn(4, 256, generator=g(210))
    torch.manual_seed(9002)
    with DropoutTap() as tap, torch.no_grad():
        logits = ae.classify(z)
    assert len(tap.masks) == 2
    npz("classify_train", z=z, mask1=tap.masks[0], mask2=tap.masks[1], logits=logits)
    ae.eval()

    # ------------------------------------------------------------------ gradients of the latent-DDPM loop (v2:1254-1264)
    Bd = 8
    batches = [(torch.rand(Bd, 3, 64, 64, generator=g(3000 + i)), torch.randint(0, 102, (Bd,), generator=g(3100 + i)))
               for i in range(2)]
    mod.class_names = []
    unet_t = mod.ConditionalUNet()
    unet_t.load_state_dict(pinit.procedural_state_dict(unet_t.state_dict(), 44))
    unet_t.eval()
    rec = []
    real_clip = torch.nn.utils.clip_grad_norm_

    def clip_tap(params, max_norm, *a, **k):
        params = list(params)
        total = real_clip(params, 1e30)  # measure only; then the real call
        rec.append((float(total), gslices(unet_t.named_parameters(), UNET_KEYS)))
        return real_clip(params, max_norm, *a, **k)

    torch.nn.utils.clip_grad_norm_ = clip_tap
    torch.manual_seed(4242)
    mod.train_conditional_diffusion(ae, unet_t, batches, num_epochs=1, lr=1e-3, visualize_every=1000, save_dir=tmp,
                                    device=torch.device("cpu"))
    torch.nn.utils.clip_grad_norm_ = real_clip
    assert len(rec) == 2
    fields = {}
    for i, (tn, gs) in enumerate(rec):
        fields["grad_norm%d" % i] = np.float32(tn)
        fields.update({"grad%d__%s" % (i, k): v for k, v in gs.items()})
    npz("ddpm_loop_grads", **fields)

    # ------------------------------------------------------------------ gradients of the VAE-GAN loop (v2:1060-1134)
    mod.models.vgg16 = fake_vgg16_factory()

    class SeededD(mod.Discriminator64):
        def __init__(self, in_channels=3):
            super().__init__(in_channels)
            self.load_state_dict(pinit.procedural_state_dict(self.state_dict(), 43))

    mod.Discriminator64 = SeededD
    D_PARAM_NAMES = [n for n, _ in SeededD().named_parameters()]
    mod.visualize_reconstructions = lambda *a_, **k_: None
    mod.visualize_latent_space = lambda *a_, **k_: None

    def run_loop(batches_, veps_, num_epochs, want_grads, dropout_log=None):
        ae_t = mod.SimpleAutoencoder()
        ae_t.load_state_dict(pinit.procedural_state_dict(ae_t.state_dict(), 42))
        it = iter(veps_)

        def reparam_explicit(mu_, logvar_):
            return mu_ + next(it) * torch.exp(0.5 * torch.clamp(logvar_, min=-2.0, max=10.0))

        ae_t.reparameterize = reparam_explicit
        steps_rec = []
        ae_ids = {id(p) for p in ae_t.parameters()}

        def pre_hook(opt, args, kwargs):
            first = opt.param_groups[0]["params"][0]
            if id(first) in ae_ids:  # the AdamW of the autoencoder: clip already applied; lr / beta1 as OneCycleLR left them
                steps_rec.append(dict(kind="ae", lr=opt.param_groups[0]["lr"], beta1=opt.param_groups[0]["betas"][0],
                                      centers=float(ae_t.class_centers.abs().sum())))
            else:
                named_d = [("model.%s" % n, p) for n, p in _disc_named(opt)]
                steps_rec.append(dict(kind="d", grads=gslices(named_d, [k for k in D_KEYS if "running" not in k]) if want_grads else None))

        def _disc_named(opt):
            # the discriminator is created inside train_autoencoder; recover names by parameter order (model.* layout)
            return [(n[len("model."):], p) for n, p in zip(D_PARAM_NAMES, opt.param_groups[0]["params"])]

        clips = []

        def clip_tap2(params, max_norm, *a, **k):
            params = list(params)
            total = real_clip(params, 1e30)
            clips.append((float(total), gslices(ae_t.named_parameters(), AE_KEYS) if want_grads else None))
            return real_clip(params, max_norm, *a, **k)

        torch.nn.utils.clip_grad_norm_ = clip_tap2
        real_save = torch.save
        torch.save = lambda *a_, **k_: None  # the loop checkpoints the 277 MB autoencoder whenever the loss improves
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        h = register_optimizer_step_pre_hook(pre_hook)
        tapctx = DropoutTap() if dropout_log is not None else None
        if tapctx is not None:
            tapctx.__enter__()
        try:
            ae_t, d_t, lh = mod.train_autoencoder(ae_t, batches_, num_epochs=num_epochs, lr=1e-4, visualize_every=100000,
                                                  save_dir=tmp)
        finally:
            h.remove()
            torch.save = real_save
            torch.nn.utils.clip_grad_norm_ = real_clip
            if tapctx is not None:
                tapctx.__exit__()
                dropout_log.extend(tapctx.masks)
        return ae_t, d_t, lh, steps_rec, clips

    Bv = 4
    vb = [(torch.rand(Bv, 3, 64, 64, generator=g(5000 + i)), torch.randint(0, 102, (Bv,), generator=g(5100 + i))) for i in range(2)]
    veps = [torch.randn(Bv, 256, generator=g(5200 + i)) for i in range(2)]
    torch.manual_seed(42)
    _, _, lh, srec, clips = run_loop(vb, veps, 1, True)
    ae_steps = [s for s in srec if s["kind"] == "ae"]
    d_steps = [s for s in srec if s["kind"] == "d"]
    assert len(ae_steps) == 2 and len(d_steps) == 2 and len(clips) == 2
    fields = {}
    for i in range(2):
        fields["grad_norm%d" % i] = np.float32(clips[i][0])
        fields["lr%d" % i] = np.float64(ae_steps[i]["lr"])
        fields["beta1_%d" % i] = np.float64(ae_steps[i]["beta1"])
        fields.update({"ae_grad%d__%s" % (i, k): v for k, v in clips[i][1].items()})
        fields.update({"d_grad%d__%s" % (i, k): v for k, v in d_steps[i]["grads"].items()})
    fields.update({"hist_" + k: np.float32(v[0]) for k, v in lh.items()})
    npz("vae_gan_loop_grads", **fields)

    # ------------------------------------------------------------------ epoch gating over 165 epochs (v2:1055, 1072-1087, 1133)
    E = 165
    gb = [(torch.rand(2, 3, 64, 64, generator=g(7000)), torch.tensor([5, 60]))]
    geps = [torch.randn(2, 256, generator=g(7100 + e)) for e in range(E)]
    dlog = []
    torch.manual_seed(43)
    _, _, lh, srec, clips = run_loop(gb, geps, E, False, dropout_log=dlog)
    ae_steps = [s for s in srec if s["kind"] == "ae"]
    assert len(ae_steps) == E
    npz("vae_gan_gating", num_epochs=np.int32(E),
        lr=np.array([s["lr"] for s in ae_steps]), beta1=np.array([s["beta1"] for s in ae_steps]),
        centers_abs_sum_before_step=np.array([s["centers"] for s in ae_steps], dtype=np.float64),
        grad_norm=np.array([cn for cn, _ in clips], dtype=np.float32), n_dropout_calls=np.int32(len(dlog)),
        **{"hist_" + k: np.array(v, dtype=np.float64) for k, v in lh.items()})
    print("done")


if __name__ == "__main__":
    main()
