"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A functional, state-dict-driven restatement (torch CPU ops, fp32 or fp64, NCHW like the
reference) of the v2 hot path of ynyeh0221/Oxford-102-Flower-GAN-VAE-latent-diffusion.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package, and only as the checker / the timed CPU baseline. The product
(`oxford_102_flower_gan_vae_latent_diffusion_amd`) never imports it.

Pinning: the reference holds no tests or golden vectors, so this oracle is
pinned by fixtures generated from the reference itself, imported on CPU in the build
container by `tests/golden/make_golden.py` (committed under tests/golden/*.npz);
`tests/test_oracle_golden.py` checks every function below against them.

Every function takes `sd`, a dict {reference state-dict key: tensor}, and a key prefix `p`.
Citations `v2:N` = /root/reference/v2/model_train_test.py:N.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- primitives
def swish(x):
    """v2:48-50."""
    return x * torch.sigmoid(x)


def ca_layer(sd, p, x):
    """CALayer v2:53-67: global average pool -> 1x1 (C->C/8) -> Swish -> 1x1 (C/8->C) -> sigmoid -> scale."""
    y = x.mean(dim=(2, 3), keepdim=True)
    y = swish(F.conv2d(y, sd[p + "conv_du.0.weight"]))
    y = torch.sigmoid(F.conv2d(y, sd[p + "conv_du.2.weight"]))
    return x * y


def spatial_attention(sd, p, x):
    """SpatialAttention v2:69-81: channel mean & max -> 7x7 conv (pad 3, no bias) -> sigmoid -> scale."""
    a = torch.cat([x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True)[0]], dim=1)
    a = torch.sigmoid(F.conv2d(a, sd[p + "conv.weight"], padding=3))
    return x * a


def layernorm2d(sd, p, x, eps=1e-5):
    """LayerNorm2d v2:144-156: per-(n,c) statistics over (H,W), biased variance, per-channel affine."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), keepdim=True, unbiased=False)
    xn = (x - mean) / torch.sqrt(var + eps)
    return xn * sd[p + "weight"].view(1, -1, 1, 1) + sd[p + "bias"].view(1, -1, 1, 1)


def residual_block(sd, p, x):
    """ResidualBlock v2:159-178."""
    out = F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    out = swish(layernorm2d(sd, p + "ln1.", out))
    out = F.conv2d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    out = layernorm2d(sd, p + "ln2.", out)
    out = ca_layer(sd, p + "ca.", out)
    out = spatial_attention(sd, p + "sa.", out)
    return swish(out + x)


def _head(sd, p, x):
    """Linear -> LayerNorm -> Swish -> Linear (fc_mu / fc_logvar, v2:210-221)."""
    h = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    h = swish(F.layer_norm(h, (h.shape[-1],), sd[p + "1.weight"], sd[p + "1.bias"]))
    return F.linear(h, sd[p + "3.weight"], sd[p + "3.bias"])


def encoder_forward(sd, p, x, return_features=False):
    """Encoder.forward v2:223-239 -> (mu, logvar)."""
    h = F.conv2d(x, sd[p + "initial_conv.0.weight"], sd[p + "initial_conv.0.bias"], padding=1)
    h = swish(layernorm2d(sd, p + "initial_conv.1.", h))
    feats = [h]
    for i in (1, 2, 3):
        h = F.conv2d(h, sd[p + "down%d.0.weight" % i], sd[p + "down%d.0.bias" % i], stride=2, padding=1)
        h = swish(layernorm2d(sd, p + "down%d.1." % i, h))
        h = residual_block(sd, p + "res%d." % i, h)
        feats.append(h)
    flat = h.reshape(h.shape[0], -1)
    mu, logvar = _head(sd, p + "fc_mu.", flat), _head(sd, p + "fc_logvar.", flat)
    return (mu, logvar, feats) if return_features else (mu, logvar)


def decoder_forward(sd, p, z):
    """Decoder.forward v2:280-290."""
    h = F.linear(z, sd[p + "fc.0.weight"], sd[p + "fc.0.bias"])
    h = swish(F.layer_norm(h, (512,), sd[p + "fc.1.weight"], sd[p + "fc.1.bias"]))
    h = F.linear(h, sd[p + "fc.3.weight"], sd[p + "fc.3.bias"])
    h = swish(F.layer_norm(h, (h.shape[-1],), sd[p + "fc.4.weight"], sd[p + "fc.4.bias"]))
    h = h.view(-1, 512, 8, 8)
    for i, groups in ((3, 32), (2, 16), (1, 8)):
        h = residual_block(sd, p + "res%d." % i, h)
        h = F.conv_transpose2d(h, sd[p + "up%d.0.weight" % i], sd[p + "up%d.0.bias" % i], stride=2, padding=1)
        h = swish(F.group_norm(h, groups, sd[p + "up%d.1.weight" % i], sd[p + "up%d.1.bias" % i]))
    h = F.conv2d(h, sd[p + "final_conv.0.weight"], sd[p + "final_conv.0.bias"], padding=1)
    h = swish(F.group_norm(h, 8, sd[p + "final_conv.1.weight"], sd[p + "final_conv.1.bias"]))
    h = F.conv2d(h, sd[p + "final_conv.3.weight"], sd[p + "final_conv.3.bias"], padding=1)
    return torch.sigmoid(h)


def clamp_logvar(logvar):
    return torch.clamp(logvar, min=-2.0, max=10.0)


def reparameterize(mu, logvar, eps):
    """v2:339-343 with the noise supplied explicitly."""
    return mu + eps * torch.exp(0.5 * clamp_logvar(logvar))


def autoencoder_forward(sd, x, eps):
    """SimpleAutoencoder.forward v2:388-393 -> (x_recon, mu, logvar, z)."""
    mu, logvar = encoder_forward(sd, "encoder.", x)
    z = reparameterize(mu, logvar, eps)
    return decoder_forward(sd, "decoder.", z), mu, logvar, z


def classify(sd, z, mask1=None, mask2=None):
    """classifier v2:311-321; dropout masks (scale tensors) explicit, None = eval."""
    p = "classifier."
    h = F.linear(z, sd[p + "0.weight"], sd[p + "0.bias"])
    h = swish(F.layer_norm(h, (512,), sd[p + "1.weight"], sd[p + "1.bias"]))
    if mask1 is not None:
        h = h * mask1
    h = F.linear(h, sd[p + "4.weight"], sd[p + "4.bias"])
    h = swish(F.layer_norm(h, (256,), sd[p + "5.weight"], sd[p + "5.bias"]))
    if mask2 is not None:
        h = h * mask2
    return F.linear(h, sd[p + "8.weight"], sd[p + "8.bias"])


def euclidean_distance_loss(x, y, reduction="mean"):
    """v2:293-302: per-sample L2 norm (not MSE)."""
    d = torch.sqrt(((x - y) ** 2).reshape(x.shape[0], -1).sum(dim=1) + 1e-8)
    if reduction == "mean":
        return d.mean()
    if reduction == "sum":
        return d.sum()
    return d


def kl_divergence(mu, logvar):
    """v2:380-386."""
    mu = torch.clamp(mu, min=-10.0, max=10.0)
    logvar = clamp_logvar(logvar)
    kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1)
    return torch.clamp(kl, min=0.0, max=100.0).mean() + 1e-4 * torch.sum(mu.pow(2))


def center_loss(centers, z, labels):
    """compute_center_loss v2:362-367."""
    return torch.sqrt(((z - centers[labels]) ** 2).sum(dim=1) + 1e-8).mean()


def update_centers(centers, z, labels, momentum=0.9):
    """update_centers v2:369-378 (returns the new centres)."""
    centers = centers.clone()
    for c in torch.unique(labels):
        centers[c] = momentum * centers[c] + (1 - momentum) * z[labels == c].mean(0)
    return centers


# ----------------------------------------------------------------------------- denoiser
def time_features(t, n_channels=256):
    """Sinusoidal part of TimeEmbedding.forward v2:410-414."""
    half = n_channels // 2
    k = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, device=t.device) * -k)
    e = t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=1)


def time_embedding(sd, p, t):
    """TimeEmbedding v2:401-418."""
    e = time_features(t).to(sd[p + "lin1.weight"].dtype)  # features are fp32 (v2:412-414); fp64 only in gradient checks
    e = swish(F.linear(e, sd[p + "lin1.weight"], sd[p + "lin1.bias"]))
    return F.linear(e, sd[p + "lin2.weight"], sd[p + "lin2.bias"])


def class_embedding(sd, p, c):
    """ClassEmbedding v2:421-431."""
    e = sd[p + "embedding.weight"][c]
    e = swish(F.linear(e, sd[p + "lin1.weight"], sd[p + "lin1.bias"]))
    return F.linear(e, sd[p + "lin2.weight"], sd[p + "lin2.bias"])


def unet_forward(sd, x, t, c, block_masks=None, head_masks=None, heads=8):
    """ConditionalUNet.forward v2:535-561.

    nn.MultiheadAttention is fed a length-1 sequence (v2:550), so softmax over one key == 1 and
    attention == out_proj(v_proj(x)) (pinned by the golden fixture generated with
    the real nn.MultiheadAttention). Train-mode dropout is explicit: block_masks[i] (B,d) scales
    the Linear->LN output (v2:522), head_masks[i] (B,heads) scales each head's value slice
    (attention dropout on the 1x1 weight, v2:514). None = eval.
    """
    te = time_embedding(sd, "time_emb.", t)
    ce = class_embedding(sd, "class_emb.", c) if c is not None else None
    h = F.linear(x, sd["latent_proj.weight"], sd["latent_proj.bias"])
    n_layers = len([k for k in sd if k.startswith("layers.") and k.endswith(".2.weight")])
    for i in range(n_layers):
        tw, tb = sd["time_projections.%d.weight" % i], sd["time_projections.%d.bias" % i]
        h = h + F.linear(te, tw, tb)
        if ce is not None:
            h = h + F.linear(ce, tw, tb)  # class embedding goes through the *time* projection (v2:544)
        d = h.shape[-1]
        g = F.linear(h, sd["layers.%d.0.0.weight" % i], sd["layers.%d.0.0.bias" % i])
        g = F.layer_norm(g, (d,), sd["layers.%d.0.1.weight" % i], sd["layers.%d.0.1.bias" % i])
        if block_masks is not None:
            g = g * block_masks[i]
        h = swish(g) + h
        hn = F.layer_norm(h, (d,), sd["layers.%d.1.weight" % i], sd["layers.%d.1.bias" % i])
        ipw, ipb = sd["attention_layers.%d.in_proj_weight" % i], sd["attention_layers.%d.in_proj_bias" % i]
        v = F.linear(hn, ipw[2 * d:], ipb[2 * d:])
        if head_masks is not None:
            v = (v.view(-1, heads, d // heads) * head_masks[i].unsqueeze(-1)).reshape(-1, d)
        h = h + F.linear(v, sd["attention_layers.%d.out_proj.weight" % i], sd["attention_layers.%d.out_proj.bias" % i])
        h = F.linear(h, sd["layers.%d.2.weight" % i], sd["layers.%d.2.bias" % i])
    h = h + F.linear(te, sd["final_time_proj.weight"], sd["final_time_proj.bias"])
    if ce is not None:
        h = h + F.linear(ce, sd["final_class_proj.weight"], sd["final_class_proj.bias"])
    h = F.layer_norm(h, (h.shape[-1],), sd["final_norm.weight"], sd["final_norm.bias"])
    out = F.linear(h, sd["final.weight"], sd["final.bias"])
    return out + torch.sigmoid(sd["residual_weight"]) * F.linear(x, sd["final.weight"], sd["final.bias"])


# ----------------------------------------------------------------------------- conv-UNet blocks (defined, never instantiated by v2)
def unet_attention_block(sd, p, x, num_heads=4):
    """UNetAttentionBlock v2:434-459: GroupNorm(1,C) -> 1x1 qkv -> per-head softmax(q k^T * d^-1/2) v over h*w tokens
    -> (quirk, v2:456-457) channels re-ordered d-major -> 1x1 proj + residual."""
    b, c, h, w = x.shape
    d = c // num_heads
    xn = F.group_norm(x, 1, sd[p + "norm.weight"], sd[p + "norm.bias"])
    qkv = F.conv2d(xn, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(b, 3, num_heads, d, h * w)
    q, k, v = qkv[:, 0].permute(0, 1, 3, 2), qkv[:, 1], qkv[:, 2].permute(0, 1, 3, 2)
    attn = F.softmax(torch.matmul(q, k) * (d ** -0.5), dim=-1)
    out = torch.matmul(attn, v).permute(0, 3, 1, 2).reshape(b, c, h, w)
    return F.conv2d(out, sd[p + "proj.weight"], sd[p + "proj.bias"]) + x


def unet_residual_block(sd, p, x, t, c=None, mask=None):
    """UNetResidualBlock v2:462-486 (dropout mask explicit, None = eval)."""
    h = swish(layernorm2d(sd, p + "norm1.", x))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = h + swish(F.linear(t, sd[p + "time_emb.weight"], sd[p + "time_emb.bias"]))[:, :, None, None]
    if c is not None:
        h = h + swish(F.linear(c, sd[p + "class_emb.weight"], sd[p + "class_emb.bias"]))[:, :, None, None]
    h = swish(layernorm2d(sd, p + "norm2.", h))
    if mask is not None:
        h = h * mask
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    res = F.conv2d(x, sd[p + "residual.weight"], sd[p + "residual.bias"]) if (p + "residual.weight") in sd else x
    return h + res


# ----------------------------------------------------------------------------- diffusion process
def schedule(n_steps=1000, dtype=torch.float32):
    """v2:569-571."""
    beta = torch.linspace(0.0001, 0.02, n_steps, dtype=dtype)
    alpha = 1 - beta
    return beta, alpha, torch.cumprod(alpha, dim=0)


def q_sample(alpha_bar, x0, t, eps):
    """v2:574-578."""
    ab = alpha_bar[t].reshape(-1, 1)
    return torch.sqrt(ab) * x0 + torch.sqrt(1 - ab) * eps


def p_sample(sd, sched, xt, t, c, noise):
    """v2:580-592 with explicit noise; t is a python int (shared by the batch)."""
    beta, alpha, alpha_bar = sched
    tt = torch.tensor([t])
    eps_theta = unet_forward(sd, xt, tt, c)
    a, ab = alpha[tt].reshape(-1, 1), alpha_bar[tt].reshape(-1, 1)
    mean = (xt - ((1 - a) / torch.sqrt(1 - ab)) * eps_theta) / torch.sqrt(a)
    if t > 0:
        return mean + torch.sqrt(beta[tt].reshape(-1, 1)) * noise
    return mean


def ddpm_loss(sd, sched, x0, labels, t, eps, block_masks=None, head_masks=None):
    """ConditionalDenoiseDiffusion.loss v2:600-607 with explicit (t, eps)."""
    xt = q_sample(sched[2], x0, t, eps)
    eps_theta = unet_forward(sd, xt, t, labels, block_masks, head_masks)
    return euclidean_distance_loss(eps, eps_theta), eps_theta


# ----------------------------------------------------------------------------- perceptual / GAN
VGG16_CFG_16 = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256)]
VGG16_CONV_IDX = [0, 2, 5, 7, 10, 12, 14]  # positions of the convs inside torchvision vgg16().features[:16]


def vgg_features(sd, p, x):
    """torchvision VGG16 features[:16] (v2:965): conv3x3+ReLU x2, pool, x2, pool, x3 — public layout;
    weights are supplied (synthetic or a user's local copy), parity at this boundary is unpinned by the
    reference (torchvision is not vendored)."""
    h = x
    ci = 0
    for item in VGG16_CFG_16:
        if item == "M":
            h = F.max_pool2d(h, 2)
        else:
            k = VGG16_CONV_IDX[ci]
            h = F.relu(F.conv2d(h, sd[p + "%d.weight" % k], sd[p + "%d.bias" % k], padding=1))
            ci += 1
    return h


def vgg_perceptual_loss(sd, x, y, p="feature_extractor."):
    """VGGPerceptualLoss.forward v2:973-981."""
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=x.dtype).view(1, 3, 1, 1)
    return euclidean_distance_loss(vgg_features(sd, p, (x - mean) / std), vgg_features(sd, p, (y - mean) / std))


def discriminator_forward(sd, x, training=True, p="model.", momentum=0.1):
    """Discriminator64 v2:983-1007. Returns (probabilities (B,), dict of updated running stats)."""
    new_stats = {}
    h = F.leaky_relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=2, padding=1), 0.2)
    for conv_i, bn_i in ((2, 3), (5, 6), (8, 9)):
        h = F.conv2d(h, sd[p + "%d.weight" % conv_i], sd[p + "%d.bias" % conv_i], stride=2, padding=1)
        rm, rv = sd[p + "%d.running_mean" % bn_i].clone(), sd[p + "%d.running_var" % bn_i].clone()
        h = F.batch_norm(h, rm, rv, sd[p + "%d.weight" % bn_i], sd[p + "%d.bias" % bn_i], training, momentum, 1e-5)
        new_stats[p + "%d.running_mean" % bn_i], new_stats[p + "%d.running_var" % bn_i] = rm, rv
        h = F.leaky_relu(h, 0.2)
    h = torch.sigmoid(F.conv2d(h, sd[p + "11.weight"], sd[p + "11.bias"]))
    return h.view(-1), new_stats


# ----------------------------------------------------------------------------- training steps (callers)
def loss_phase(epoch):
    """Loss gating v2:1072-1087 -> (kl_factor, cls_factor, center_factor)."""
    if epoch < 40:
        return 0.0, 0.0, 0.0
    if epoch < 80:
        return min(1.0, (epoch - 20) / 20), 0.0, 0.0
    if epoch < 160:
        return 1.0, min(0.2, (epoch - 40) / 20), 0.0
    return 1.0, 1.0, min(1.0, (epoch - 60) / 20)


def kl_weight_at(epoch, num_epochs, start=0.001, end=0.05):
    """v2:1055."""
    return min(end, start + (epoch / (num_epochs * 0.6)) * (end - start))


def _params(sd, keys=None):
    out = {}
    for k, v in sd.items():
        if v.is_floating_point() and (keys is None or k in keys) and "running_" not in k and "class_centers" not in k \
                and "center_counts" not in k:
            out[k] = v.clone().requires_grad_(True)
        else:
            out[k] = v.clone()
    return out


def ddpm_step(ae_sd, unet_sd, data, labels, reparam_eps, t, eps, lr=1e-3, block_masks=None, head_masks=None,
              opt_state=None):
    """One iteration of train_conditional_diffusion's loop body v2:1254-1264 (explicit noise).
    Returns dict(loss, eps_theta, grad_norm, z, new_unet_sd, grads, opt)."""
    with torch.no_grad():
        mu, logvar = encoder_forward(ae_sd, "encoder.", data)
        z = reparameterize(mu, clamp_logvar(logvar), reparam_eps)  # v2:1258-1259 (clamp is idempotent)
    sd = _params(unet_sd)
    names = [k for k, v in sd.items() if v.requires_grad]
    params = [sd[k] for k in names]
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=1e-5, betas=(0.9, 0.999))
    if opt_state is not None:
        opt.load_state_dict(opt_state)
    sched = schedule()
    loss, eps_theta = ddpm_loss(sd, sched, z, labels, t, eps, block_masks, head_masks)
    opt.zero_grad()
    loss.backward()
    grads = {k: (sd[k].grad.clone() if sd[k].grad is not None else None) for k in names}
    gn = torch.nn.utils.clip_grad_norm_([p_ for p_ in params], max_norm=1.0)
    opt.step()
    return dict(loss=loss.detach(), eps_theta=eps_theta.detach(), grad_norm=gn.detach(), z=z,
                new_unet_sd={k: v.detach() for k, v in sd.items()}, grads=grads, opt=opt)


def vae_gan_step(ae_sd, d_sd, vgg_sd, data, labels, reparam_eps, epoch=200, num_epochs=300, lr=1e-4,
                 lambda_cls=0.1, lambda_center=0.05, lambda_vgg=0.4, lambda_gan=0.2, cls_masks=(None, None),
                 ae_opt_state=None, d_opt_state=None, ae_beta1=0.9):
    """One iteration of train_autoencoder's loop body v2:1060-1134 (explicit reparam noise; classifier
    dropout masks explicit). Order of operations, the D-update-before-G-adversarial ordering (v2:1097-1105),
    the detached loss-balancing scales (v2:1107-1115) and the optimizers (v2:1026-1028) follow the reference.
    lr / ae_beta1: what OneCycleLR (v2:1031-1037; it cycles AdamW's beta1 too) has set for this step."""
    ae = _params(ae_sd)
    dd = _params(d_sd)
    ae_names = [k for k, v in ae.items() if v.requires_grad]
    d_names = [k for k, v in dd.items() if v.requires_grad]
    opt = torch.optim.AdamW([ae[k] for k in ae_names], lr=lr, weight_decay=1e-5, betas=(ae_beta1, 0.999))
    d_opt = torch.optim.Adam([dd[k] for k in d_names], lr=1e-4, betas=(0.5, 0.999))
    if ae_opt_state is not None:
        opt.load_state_dict(ae_opt_state)
        for gp in opt.param_groups:
            gp["lr"], gp["betas"] = lr, (ae_beta1, 0.999)
    if d_opt_state is not None:
        d_opt.load_state_dict(d_opt_state)
    kl_weight = kl_weight_at(epoch, num_epochs)
    kl_factor, cls_factor, center_factor = loss_phase(epoch)
    B = data.shape[0]
    valid, fake = torch.ones(B, dtype=data.dtype), torch.zeros(B, dtype=data.dtype)

    recon_x, mu, logvar, z = autoencoder_forward(ae, data, reparam_eps)
    zero = torch.tensor(0.0, dtype=data.dtype)
    recon_loss = euclidean_distance_loss(recon_x, data)
    perceptual = vgg_perceptual_loss(vgg_sd, recon_x, data)
    kl = kl_divergence(mu, logvar) if kl_factor > 0 else zero
    cls = F.cross_entropy(classify(ae, z, *cls_masks), labels) if cls_factor > 0 else zero
    cen = center_loss(ae["class_centers"], z, labels) if center_factor > 0 else zero

    d_real, st1 = discriminator_forward(dd, data, True)
    dd.update(st1)
    d_fake, st2 = discriminator_forward(dd, recon_x.detach(), True)
    dd.update(st2)
    d_loss = (F.binary_cross_entropy(d_real, valid) + F.binary_cross_entropy(d_fake, fake)) / 2
    d_opt.zero_grad()
    d_loss.backward()
    d_grads = {k: dd[k].grad.clone() for k in d_names}
    d_opt.step()
    d_adv, st3 = discriminator_forward(dd, recon_x, True)
    dd.update(st3)
    adv = F.binary_cross_entropy(d_adv, valid)

    if recon_loss.item() > 1e-8:
        p_scale = min(1.0, recon_loss.item() / (perceptual.item() + 1e-8))
        k_scale = min(1.0, recon_loss.item() / (kl.item() + 1e-8)) if kl.item() > 0 else 1.0
        g_scale = min(1.0, recon_loss.item() / (adv.item() + 1e-8))
    else:
        p_scale = k_scale = g_scale = 1.0
    total = (1.0 * recon_loss + lambda_vgg * p_scale * perceptual + kl_weight * k_scale * kl_factor * kl
             + lambda_cls * cls_factor * cls + lambda_center * center_factor * cen + lambda_gan * g_scale * adv)
    opt.zero_grad()
    for k in d_names:
        dd[k].grad = None
    total.backward()
    ae_grads = {k: (ae[k].grad.clone() if ae[k].grad is not None else None) for k in ae_names}
    gn = torch.nn.utils.clip_grad_norm_([ae[k] for k in ae_names], max_norm=1.0)
    opt.step()
    centers = ae["class_centers"]
    if epoch >= 60 and center_factor > 0:
        centers = update_centers(centers, z.detach(), labels, 0.9)
    return dict(losses=dict(total=total.detach(), recon=recon_loss.detach(), perceptual=perceptual.detach(),
                            kl=kl.detach(), cls=cls.detach(), center=cen.detach(), adv=adv.detach(),
                            d_loss=d_loss.detach()),
                recon_x=recon_x.detach(), mu=mu.detach(), logvar=logvar.detach(), z=z.detach(), grad_norm=gn.detach(),
                new_ae_sd={k: v.detach() for k, v in ae.items()}, new_d_sd={k: v.detach() for k, v in dd.items()},
                ae_grads=ae_grads, d_grads=d_grads, centers=centers.detach(), opt=opt, d_opt=d_opt)
