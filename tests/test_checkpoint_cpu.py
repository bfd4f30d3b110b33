"""Checkpoint interop on CPU: reference-format containers round-trip, epoch-from-filename, flat-buffer
re-homing keeps state_dict semantics."""
import os

import torch

from oxford_102_flower_gan_vae_latent_diffusion_amd import checkpoint as ck
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
from oxford_102_flower_gan_vae_latent_diffusion_amd import steps


def test_reference_containers_round_trip(tmp_path):
    torch.manual_seed(0)
    un, un2 = M.ConditionalUNet(), M.ConditionalUNet()
    p = os.path.join(tmp_path, "conditional_diffusion_epoch_120.pt")
    ck.save_unet(p, un)
    assert set(torch.load(p).keys()) == set(un.state_dict().keys())       # bare state-dict, like v2:1282
    assert ck.load_unet(p, un2) == 120                                     # epoch parsed from the name, v2:1354-1356
    for k, v in un.state_dict().items():
        assert torch.equal(v, un2.state_dict()[k]), k
    assert ck.epoch_from_filename("conditional_diffusion_final.pt") == 0
    d, d2 = M.Discriminator64(), M.Discriminator64()
    # only the (small) discriminator + a stub AE dict is written here; the container layout is what matters
    p2 = os.path.join(tmp_path, "vae_gan_best.pt")
    torch.save({"autoencoder": {}, "discriminator": d.state_dict()}, p2)
    ck.load_vae_gan(p2, torch.nn.Module(), d2)
    for k, v in d.state_dict().items():
        assert torch.equal(v, d2.state_dict()[k]), k


def test_flat_params_keep_state_dict_semantics():
    torch.manual_seed(1)
    un = M.ConditionalUNet()
    sd = {k: v.clone() for k, v in un.state_dict().items()}
    fp = steps.FlatParams(un, ("time_projections.4.", "attention_layers.4."))
    assert fp.n_active < fp.n_total
    for k, v in un.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # loading a checkpoint writes through the views into the flat buffer
    new = {k: (v + 1 if v.is_floating_point() else v) for k, v in sd.items()}
    un.load_state_dict(new)
    o, n, shp = fp.slots["final.weight"]
    assert torch.equal(fp.flat[o:o + n].view(shp), new["final.weight"])
    assert un.final.weight.grad.data_ptr() == fp.gviews["final.weight"].data_ptr()
