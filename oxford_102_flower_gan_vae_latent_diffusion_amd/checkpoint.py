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
            out[name] = {"seed": r.seed, "state": {str(k): v.detach().cpu() for k, v in r._state.items()}}
    return out


def save_train_state(path, trainer):
    """Side-car with everything needed for a bit-exact resume of a DDPMTrainer."""
    opt = trainer.opt
    torch.save({"m": opt.m.detach().cpu(), "v": opt.v.detach().cpu(), "step": opt.step_count.detach().cpu(),
                "lr": opt.lr.detach().cpu(),
                "rng": _rngs(("ae", trainer.ae), ("unet", trainer.unet), ("diffusion", trainer.diffusion))}, path)


def load_train_state(path, trainer):
    st = torch.load(path, map_location="cpu")
    opt = trainer.opt
    opt.m.copy_(st["m"])
    opt.v.copy_(st["v"])
    opt.step_count.copy_(st["step"])
    opt.lr.copy_(st["lr"])
    for name, o in (("ae", trainer.ae), ("unet", trainer.unet), ("diffusion", trainer.diffusion)):
        info = st["rng"].get(name)
        if info is None:
            continue
        o.rng.seed = info["seed"]
        dev = next(iter(trainer.unet.parameters())).device
        for _, v in info["state"].items():
            o.rng.state(dev).copy_(v)
    return st
