"""Training-side compositions for the VAE-GAN half of the surface (classifier, VGG perceptual features,
Discriminator64) and the autograd Functions that give the drop-in modules a backward.

Forward compositions are complete; the convolutional backward chains (A5-A7, A14, A15 of SURVEY.md §8a) are
wired kernel-by-kernel through ops.conv2d_dgrad / conv2d_wgrad / *norm_bwd.
"""
import torch

from . import engine, ops

SW = ops.ACT_SWISH
_d = engine._d


def _f32c(x):
    return x if x.is_contiguous() else x.contiguous()


# ----------------------------------------------------------------------------- small helpers
def clamp_bwd(d, x, lo, hi):
    """gradient of clamp(x, lo, hi): passes where lo <= x <= hi (torch semantics)."""
    # act_bwd with RELU on (x-lo) and (hi-x) would need two passes; use the dedicated kernel via kl-style mask:
    # dx = d * [lo <= x <= hi]  == d * relu'(x - lo) * relu'(hi - x) up to the boundary convention.
    m1 = ops.act_bwd(d, ops.axpby(1.0, x, 0.0, None, None) if False else _shift(x, -lo), ops.ACT_RELU)
    return ops.act_bwd(m1, _shift_neg(x, hi), ops.ACT_RELU)


def _shift(x, a):
    """x + a via the affine kernel's scalar form."""
    one = torch.ones(1, device=x.device)
    return ops.axpby(1.0, x, a, one.expand_as(x).contiguous())


def _shift_neg(x, a):
    one = torch.ones(1, device=x.device)
    return ops.axpby(-1.0, x, a, one.expand_as(x).contiguous())


def reparam_dlogvar(dz, logvar, eps):
    """z = mu + eps*exp(0.5*clamp(lv)); dz/dlv = 0.5*eps*exp(0.5*lv_c) inside the clamp window, else 0."""
    raise NotImplementedError("reparameterize backward lands with the VAE-GAN step")


# ----------------------------------------------------------------------------- classifier (v2:311-321)
def classify(ae, z, masks=(None, None)):
    if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in ae.classifier.parameters())):
        raise NotImplementedError("classifier backward lands with the VAE-GAN step")
    c = ae.classifier
    m1, m2 = masks
    if ae.training and m1 is None:
        m1 = ae.rng.dropout((z.shape[0], 512), z.device, c[3].p)
        m2 = ae.rng.dropout((z.shape[0], 256), z.device, c[7].p)
    h = ops.linear(_f32c(z), _d(c[0].weight), _d(c[0].bias))
    h, _, _ = ops.layernorm_fwd(h, _d(c[1].weight), _d(c[1].bias), eps=c[1].eps, act=SW, save=False)
    if m1 is not None:  # Dropout AFTER Swish here (v2:314-315), unlike the denoiser block
        h = _mul(h, m1)
    h = ops.linear(h, _d(c[4].weight), _d(c[4].bias))
    h, _, _ = ops.layernorm_fwd(h, _d(c[5].weight), _d(c[5].bias), eps=c[5].eps, act=SW, save=False)
    if m2 is not None:
        h = _mul(h, m2)
    return ops.linear(h, _d(c[8].weight), _d(c[8].bias))


def _mul(x, m):
    """elementwise product via scale_heads with one 'head' per element."""
    B, D = x.shape
    return ops.scale_heads(x, m, D)


# ----------------------------------------------------------------------------- VGG16[:16] features (v2:965)
def vgg_features_nhwc(vgg, x4):
    """x4: normalised NHWC4 image -> (B,16,16,256) features; conv3x3+bias+ReLU fused, maxpool2 as a kernel."""
    raise NotImplementedError


def vgg_loss(vgg, x, y):
    raise NotImplementedError("VGG perceptual path lands with the VAE-GAN step")


# ----------------------------------------------------------------------------- Discriminator64 (v2:983-1007)
def discriminator(dm, x):
    raise NotImplementedError("Discriminator64 path lands with the VAE-GAN step")
