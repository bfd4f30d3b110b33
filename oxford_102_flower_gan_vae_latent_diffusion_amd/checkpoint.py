"""Checkpoint interop + resume.

On-disk format = the reference's: `{'autoencoder': sd, 'discriminator': sd}` for the VAE-GAN (v2:1179-1191) and a bare
ConditionalUNet state-dict (v2:1282-1283); the epoch is parsed from `..._epoch_N.pt` (v2:1354-1356) and loading is
`strict=False` (v2:1358). What the reference does not save — optimizer moments/step, learning rate and the Philox
counters — goes into an optional side-car `*.trainstate.pt`, so a resumed run continues bit-exactly while a plain
reference checkpoint still loads.
"""
import os
import re

import torch


def save_vae_gan(path, autoencoder, discriminator):
    torch.save({"autoencoder": autoencoder.state_dict(), "discriminator": discriminator.state_dict()}, path)


def load_vae_gan(path, autoencoder, discriminator=None, map_location="cpu"):
    ckpt = torch.load(path, map_location=map_location)
    if "autoencoder" in ckpt:
        autoencoder.load_state_dict(ckpt["autoencoder"], strict=False)
        if discriminator is not None and "discriminator" in ckpt:
            discriminator.load_state_dict(ckpt["discriminator"], strict=False)
    else:  # v2:1326 saves the bare autoencoder state-dict under flowers_autoencoder.pt
        autoencoder.load_state_dict(ckpt, strict=False)
    return ckpt


def save_unet(path, unet):
    torch.save(unet.state_dict(), path)


def epoch_from_filename(path):
    """v2:1354-1356: '..._epoch_123.pt' -> 123 (0 when absent)."""
    m = re.search(r"epoch_(\d+)", os.path.basename(path))
    return int(m.group(1)) if m else 0


def load_unet(path, unet, map_location="cpu"):
    unet.load_state_dict(torch.load(path, map_location=map_location), strict=False)
    return epoch_from_filename(path)


def _rngs(*objs):
    out = {}
    for name, o in objs:
        r = getattr(o, "rng", None)
        if r is not None:
            out[name] = {"seed": r.seed, "state": {"%s:%s" % k: v.detach().cpu() for k, v in r._state.items()}}
    return out


def _load_rngs(saved, dev, *objs):
    """Philox counters are per device: restore each one onto the device it was saved for (a state saved for 'cuda:i' goes
    to `dev` when it is a cuda device - one process per GPU, the index may differ after a resume)."""
    dev = torch.device(dev)
    for name, o in objs:
        info = saved.get(name)
        r = getattr(o, "rng", None)
        if info is None or r is None:
            continue
        r.seed = info["seed"]
        for key, v in info["state"].items():
            kind = str(key).split(":")[0].strip("(' ")
            if kind == dev.type or ("cuda" in kind and dev.type == "cuda"):
                r.state(dev).copy_(v)
            elif kind == "cpu":
                r.state("cpu").copy_(v)


def save_train_state(path, trainer, scheduler=None, extra=None):
    """Side-car with everything the reference checkpoint omits, for a bit-exact resume: optimizer moments / step / lr
    (both optimizers of a VAEGANTrainer), the LR-schedule position and the Philox counters. Works for DDPMTrainer and
    VAEGANTrainer (the class centres and BatchNorm buffers travel in the state-dicts, checkpoint.save_vae_gan)."""
    st = {"opt": trainer.opt.state_dict(), "extra": extra}
    if hasattr(trainer, "d_opt"):
        st["kind"] = "vaegan"
        st["d_opt"] = trainer.d_opt.state_dict()
        st["rng"] = _rngs(("ae", trainer.ae), ("disc", trainer.disc), ("vgg", trainer.vgg))
    else:
        st["kind"] = "ddpm"
        st["rng"] = _rngs(("ae", trainer.ae), ("unet", trainer.unet), ("diffusion", trainer.diffusion))
    if scheduler is not None:
        st["scheduler"] = scheduler.state_dict()
    torch.save(st, path)


def load_train_state(path, trainer, scheduler=None):
    st = torch.load(path, map_location="cpu", weights_only=False)
    if "opt" not in st:  # round-1 side-car layout (DDPM only)
        st = {"kind": "ddpm", "opt": {k: st[k] for k in ("m", "v", "step", "lr")}, "rng": st.get("rng", {})}
    trainer.opt.load_state_dict(st["opt"])
    if st["kind"] == "vaegan":
        trainer.d_opt.load_state_dict(st["d_opt"])
        _load_rngs(st["rng"], trainer.dev, ("ae", trainer.ae), ("disc", trainer.disc), ("vgg", trainer.vgg))
    else:
        _load_rngs(st["rng"], trainer.dev, ("ae", trainer.ae), ("unet", trainer.unet), ("diffusion", trainer.diffusion))
    if scheduler is not None and "scheduler" in st:
        scheduler.load_state_dict(st["scheduler"])  # re-applies lr (and beta1) for the restored position
    return st
