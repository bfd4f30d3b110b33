"""Training-side compositions of the VAE-GAN half of the surface: forward-with-saved-activations and manual
backward chains for ResidualBlock / Encoder / Decoder / classifier / VGG16[:16] / Discriminator64 (SURVEY §8a
A2-A9, A14, A15), plus the torch.autograd.Function wrappers that give the drop-in modules a backward.

All arithmetic is libflowers_hip.so kernels via ops.py (NHWC fp32). Gradients are returned as dicts keyed by the
module-relative parameter names of the reference's state-dict.
"""
import os

import torch

from . import engine, ops

SW, RELU, LRELU, SIG, NONE = ops.ACT_SWISH, ops.ACT_RELU, ops.ACT_LRELU, ops.ACT_SIGMOID, ops.ACT_NONE
_d = engine._d


def _f32c(x):
    return x if x.is_contiguous() else x.contiguous()


def _add(a, b):
    return ops.axpby(1.0, a, 1.0, b)


def _c2(x):
    """(B,H,W,C) -> (B*H*W, C) view."""
    return x.view(-1, x.shape[-1])


# f16x2 split math (csrc/conv_h2.hip) for the 3x3 / stride-1 convs of the TRAINING paths too (forward, and the input
# gradient as the mirror convolution): the weights are re-packed once per weight version (= per optimizer step for the
# trainable ones, once for the frozen VGG), the input's range is its measured max (fl_absmax). FL_TRAIN_F16X2=0: bf16x3.
USE_F16X2_TRAIN = os.environ.get("FL_TRAIN_F16X2", "1") != "0"


def _h2_ok(x, Cout, k=3, stride=1):
    """the f16x2 kernel's tiles are 256 (or 128) pixels x 128 channels, one per block: below ~3/4 of the CUs busy (batch 64
    on the 8-wide stage: 128 tiles) the generic split-K kernel is the faster one"""
    B, H, W, _ = x.shape
    return (USE_F16X2_TRAIN and ops.get_gemm_math() == "split" and B > 0 and (B * H * W // 128) * (Cout // 128) >= 192
            and ops.f16x2_ok(x, Cout, k, k, stride, 1))


def conv3_fwd_ln(x, conv, ln, pk):
    """3x3 s1 p1 conv + fused LayerNorm2d statistics -> (y, (scale, shift, mean, rstd))"""
    if _h2_ok(x, conv.weight.shape[0]):
        wh, geom = pk.conv_h2(conv)
        return ops.conv2d_fwd_f16x2(x, wh, geom, _d(conv.bias), 1, 1, ln_w=_d(ln.weight), ln_b=_d(ln.bias), eps=ln.eps,
                                    x_amax=ops.absmax(x))
    return ops.conv2d_fwd_instnorm(x, pk.conv(conv), _d(conv.bias), 1, 1, _d(ln.weight), _d(ln.bias), ln.eps)


def conv3_fwd_act(x, conv, pk, act):
    if _h2_ok(x, conv.weight.shape[0]):
        wh, geom = pk.conv_h2(conv)
        return ops.conv2d_fwd_f16x2(x, wh, geom, _d(conv.bias), 1, 1, act=act, x_amax=ops.absmax(x))
    return ops.conv2d_fwd(x, pk.conv(conv), _d(conv.bias), 1, 1, act=act)


def conv3_dgrad(dy, conv, hw, pk):
    """input gradient of a 3x3 s1 p1 conv: the mirror convolution's forward on dy when the f16x2 kernel covers the shape"""
    if _h2_ok(dy, conv.weight.shape[1]):
        wh, geom = pk.conv_h2_dgrad(conv)
        return ops.conv2d_fwd_f16x2(dy, wh, geom, None, 1, 1, x_amax=ops.absmax(dy))
    return ops.conv2d_dgrad(dy, pk.conv(conv), hw, 1, 1)


def reparam_dlogvar(dz, logvar, eps):
    return ops.reparam_bwd(dz, logvar, eps)


def clamp_bwd(d, x, lo, hi):
    return ops.clamp_bwd(d, x, lo, hi)


# ============================================================================= ResidualBlock (v2:159-178)
class RBSaved:
    pass


def _ca_mlp_fwd(ca, pooled):
    w1 = _d(ca.conv_du[0].weight).view(ca.conv_du[0].weight.shape[0], -1)
    w2 = _d(ca.conv_du[2].weight).view(ca.conv_du[2].weight.shape[0], -1)
    hid_pre = ops.linear(pooled, w1)
    hid = ops.act_fwd(hid_pre, SW)
    s_pre = ops.linear(hid, w2)
    s = ops.act_fwd(s_pre, SIG)
    return s, (pooled, w1, w2, hid_pre, hid, s_pre)


def _ca_mlp_bwd(saved, ds):
    """-> dW1 (C/8,C,1,1), dW2 (C,C/8,1,1), dpooled"""
    pooled, w1, w2, hid_pre, hid, s_pre = saved
    ds_pre = ops.act_bwd(ds, s_pre, SIG)
    dw2 = ops.linear_wgrad(ds_pre, hid)
    dhid_pre = ops.act_bwd(ops.linear_dgrad(ds_pre, w2), hid_pre, SW)
    dw1 = ops.linear_wgrad(dhid_pre, pooled)
    dpooled = ops.linear_dgrad(dhid_pre, w1)
    return dw1.view(dw1.shape[0], dw1.shape[1], 1, 1), dw2.view(dw2.shape[0], dw2.shape[1], 1, 1), dpooled


def resblock_fwd(x, rb, pk):
    S = RBSaved()
    S.x = x
    S.y1, S.n1 = conv3_fwd_ln(x, rb.conv1, rb.ln1, pk)
    S.a1 = ops.affine_act(S.y1, S.n1[0], S.n1[1], act=SW)
    S.y2, S.n2 = conv3_fwd_ln(S.a1, rb.conv2, rb.ln2, pk)
    s, S.ca = _ca_mlp_fwd(rb.ca, _d(rb.ln2.bias).view(1, -1))  # avg_pool(LN2d(.)) == ln2.bias (see engine.resblock_infer)
    S.s = s.view(-1)
    S.saw = _d(rb.sa.conv.weight).contiguous()
    S.map = ops.res_tail_map(S.y2, S.n2[0], S.n2[1], S.s)
    out, S.att = ops.res_tail_out(S.y2, S.n2[0], S.n2[1], S.s, S.map, S.saw, xres=x, final_swish=True, save_att=True)
    return out, S


def resblock_bwd(S, dout, rb, pk, need_dx=True):
    G = {}
    sc2, sh2, m2, r2 = S.n2
    dw, da = ops.res_tail_bwd_a(dout, S.y2, sc2, sh2, S.s, S.att, S.x, final_swish=True)
    dmap, G["sa.conv.weight"] = ops.sa_conv_bwd(da, S.map, S.saw)
    dn = ops.res_tail_bwd_c(dw, S.y2, sc2, sh2, S.s, S.att, dmap)
    dy2, dln2w, dln2b = ops.instnorm_bwd(dn, S.y2, sc2, sh2, _d(rb.ln2.weight), m2, r2)
    ds = ops.ca_gate_grad(_d(rb.ln2.weight), _d(rb.ln2.bias), dln2w, dln2b, S.s)
    G["ca.conv_du.0.weight"], G["ca.conv_du.2.weight"], dpooled = _ca_mlp_bwd(S.ca, ds.view(1, -1))
    G["ln2.weight"], G["ln2.bias"] = dln2w, _add(dln2b, dpooled.view(-1))
    C = S.y2.shape[-1]
    hw = (S.y2.shape[1], S.y2.shape[2])
    G["conv2.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy2, S.a1, 3, 3, 1, 1), C, C, 3, 3)
    G["conv2.bias"] = ops.colsum(_c2(dy2))
    da1 = conv3_dgrad(dy2, rb.conv2, hw, pk)
    sc1, sh1, m1, r1 = S.n1
    dy1, G["ln1.weight"], G["ln1.bias"] = ops.instnorm_bwd(da1, S.y1, sc1, sh1, _d(rb.ln1.weight), m1, r1, act=SW)
    G["conv1.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy1, S.x, 3, 3, 1, 1), C, C, 3, 3)
    G["conv1.bias"] = ops.colsum(_c2(dy1))
    dx = None
    if need_dx:
        dx = _add(conv3_dgrad(dy1, rb.conv1, hw, pk), dw)
    return dx, G


def _pref(G, prefix, out):
    for k, v in G.items():
        out[prefix + k] = v


# ============================================================================= Encoder (v2:181-239)
def _head_fwd(fc, flat):
    g = ops.linear(flat, _d(fc[0].weight), _d(fc[0].bias))
    g2, mean, rstd = ops.layernorm_fwd(g, _d(fc[1].weight), _d(fc[1].bias), eps=fc[1].eps, act=SW)
    out = ops.linear(g2, _d(fc[3].weight), _d(fc[3].bias))
    return out, (flat, g, g2, mean, rstd)


def _head_bwd(fc, saved, dout, prefix, G, R=None):
    flat, g, g2, mean, rstd = saved
    G[prefix + "3.weight"] = ops.linear_wgrad(dout, g2)
    G[prefix + "3.bias"] = ops.colsum(dout)
    dg2 = ops.linear_dgrad(dout, _d(fc[3].weight))
    dg, G[prefix + "1.weight"], G[prefix + "1.bias"] = ops.layernorm_bwd(dg2, g, _d(fc[1].weight), _d(fc[1].bias), None,
                                                                         mean, rstd, act=SW)
    G[prefix + "0.weight"] = ops.linear_wgrad(dg, flat)
    G[prefix + "0.bias"] = ops.colsum(dg)
    K = fc[0].weight.shape[1]
    return ops.gemm(dg, _d(fc[0].weight), dg.shape[0], K, dg.shape[1], a_layout=ops.KC, b_layout=ops.KS,
                    lda=dg.shape[1], ldb=K, R=R)


def encoder_fwd(enc, x):
    pk = enc._pk
    S = RBSaved()
    B = x.shape[0]
    S.x4 = ops.nchw_to_nhwc(x, 4)
    c0, ln0 = enc.initial_conv[0], enc.initial_conv[1]
    S.y0, S.n0 = ops.conv2d_fwd_instnorm(S.x4, pk.conv(c0), _d(c0.bias), 1, 1, _d(ln0.weight), _d(ln0.bias), ln0.eps)
    h = ops.affine_act(S.y0, S.n0[0], S.n0[1], act=SW)
    S.stages = []
    feats = [h]
    for i in (1, 2, 3):
        dn, rb = getattr(enc, "down%d" % i), getattr(enc, "res%d" % i)
        hin = h
        yd, nd = ops.conv2d_fwd_instnorm(hin, pk.conv(dn[0]), _d(dn[0].bias), 2, 1, _d(dn[1].weight), _d(dn[1].bias),
                                         dn[1].eps)
        ad = ops.affine_act(yd, nd[0], nd[1], act=SW)
        h, Srb = resblock_fwd(ad, rb, pk)
        S.stages.append((hin, yd, nd, Srb))
        feats.append(h)
    flat = ops.nhwc_to_nchw(h).view(B, -1)
    mu, S.hmu = _head_fwd(enc.fc_mu, flat)
    logvar, S.hlv = _head_fwd(enc.fc_logvar, flat)
    return mu, logvar, feats, S


def encoder_bwd(enc, S, dmu, dlogvar, need_dx=False):
    pk = enc._pk
    G = {}
    dflat = _head_bwd(enc.fc_mu, S.hmu, dmu, "fc_mu.", G)
    dflat = _head_bwd(enc.fc_logvar, S.hlv, dlogvar, "fc_logvar.", G, R=dflat)
    B = dflat.shape[0]
    dh = ops.nchw_to_nhwc(dflat.view(B, 512, 8, 8))
    for i in (3, 2, 1):
        dn, rb = getattr(enc, "down%d" % i), getattr(enc, "res%d" % i)
        hin, yd, nd, Srb = S.stages[i - 1]
        dad, Grb = resblock_bwd(Srb, dh, rb, pk)
        _pref(Grb, "res%d." % i, G)
        dyd, G["down%d.1.weight" % i], G["down%d.1.bias" % i] = ops.instnorm_bwd(dad, yd, nd[0], nd[1], _d(dn[1].weight),
                                                                                 nd[2], nd[3], act=SW)
        Co, Ci = dn[0].weight.shape[0], dn[0].weight.shape[1]
        G["down%d.0.weight" % i] = ops.unpack_conv_weight(ops.conv2d_wgrad(dyd, hin, 4, 4, 2, 1), Co, Ci, 4, 4)
        G["down%d.0.bias" % i] = ops.colsum(_c2(dyd))
        dh = ops.conv2d_dgrad(dyd, pk.conv(dn[0]), (hin.shape[1], hin.shape[2]), 2, 1)
    c0, ln0 = enc.initial_conv[0], enc.initial_conv[1]
    dy0, G["initial_conv.1.weight"], G["initial_conv.1.bias"] = ops.instnorm_bwd(dh, S.y0, S.n0[0], S.n0[1], _d(ln0.weight),
                                                                                S.n0[2], S.n0[3], act=SW)
    G["initial_conv.0.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy0, S.x4, 3, 3, 1, 1), 64, 3, 3, 3)
    G["initial_conv.0.bias"] = ops.colsum(_c2(dy0))
    dx = None
    if need_dx:
        dx = ops.nhwc_to_nchw(ops.conv2d_dgrad(dy0, pk.conv(c0), (64, 64), 1, 1), 3)
    return G, dx


# ============================================================================= Decoder (v2:242-290)
def _final_pack(dec):
    """final conv 32->3 computed with Cout padded to 4 (zero row) so its dgrad gathers 16-byte vectors."""
    f3 = dec.final_conv[3]

    def mk():
        wp = ops.pack_conv_weight(f3.weight.detach(), None, 4)
        b4 = torch.zeros(4, dtype=torch.float32, device=f3.weight.device)
        ops.axpby(1.0, f3.bias.detach(), out=b4[:3])
        return wp, b4

    return dec._pk.get(("final4", id(f3)), [f3.weight, f3.bias], mk)


def decoder_fwd(dec, z):
    pk = dec._pk
    S = RBSaved()
    B = z.shape[0]
    fc = dec.fc
    S.z = z
    S.g0 = ops.linear(z, _d(fc[0].weight), _d(fc[0].bias))
    S.h0, S.m0, S.r0 = ops.layernorm_fwd(S.g0, _d(fc[1].weight), _d(fc[1].bias), eps=fc[1].eps, act=SW)
    S.g3 = ops.linear(S.h0, _d(fc[3].weight), _d(fc[3].bias))
    h3, S.m3, S.r3 = ops.layernorm_fwd(S.g3, _d(fc[4].weight), _d(fc[4].bias), eps=fc[4].eps, act=SW)
    h = ops.nchw_to_nhwc(h3.view(B, 512, 8, 8))
    S.stages = {}
    for i in (3, 2, 1):
        rb, up = getattr(dec, "res%d" % i), getattr(dec, "up%d" % i)
        hr, Srb = resblock_fwd(h, rb, pk)
        yu = ops.convT2d_fwd(hr, pk.conv(up[0]), _d(up[0].bias))
        nu = ops.groupnorm_stats(yu, _d(up[1].weight), _d(up[1].bias), up[1].num_groups, up[1].eps)
        h = ops.affine_act(yu, nu[0], nu[1], act=SW)
        S.stages[i] = (Srb, hr, yu, nu)
    f = dec.final_conv
    S.au = h
    S.yf = ops.conv2d_fwd(h, pk.conv(f[0]), _d(f[0].bias), 1, 1)
    S.nf = ops.groupnorm_stats(S.yf, _d(f[1].weight), _d(f[1].bias), f[1].num_groups, f[1].eps)
    S.af = ops.affine_act(S.yf, S.nf[0], S.nf[1], act=SW)
    wp4, b4 = _final_pack(dec)
    S.y4 = ops.conv2d_fwd(S.af, wp4, b4, 1, 1)
    o4 = ops.act_fwd(S.y4, SIG)
    return ops.nhwc_to_nchw(o4, 3), S


def decoder_bwd(dec, S, dout, need_dz=True):
    pk = dec._pk
    G = {}
    f = dec.final_conv
    wp4, _ = _final_pack(dec)
    dy4 = ops.act_bwd(ops.nchw_to_nhwc(dout, 4), S.y4, SIG)
    G["final_conv.3.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy4, S.af, 3, 3, 1, 1), 3, 32, 3, 3)
    G["final_conv.3.bias"] = ops.colsum(_c2(dy4))[:3]
    daf = ops.conv2d_dgrad(dy4, wp4, (64, 64), 1, 1)
    dyf, G["final_conv.1.weight"], G["final_conv.1.bias"] = ops.groupnorm_bwd(daf, S.yf, S.nf[0], S.nf[1], _d(f[1].weight),
                                                                              S.nf[2], S.nf[3], f[1].num_groups, act=SW)
    G["final_conv.0.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dyf, S.au, 3, 3, 1, 1), 32, 64, 3, 3)
    G["final_conv.0.bias"] = ops.colsum(_c2(dyf))
    dh = ops.conv2d_dgrad(dyf, pk.conv(f[0]), (64, 64), 1, 1)
    for i in (1, 2, 3):
        rb, up = getattr(dec, "res%d" % i), getattr(dec, "up%d" % i)
        Srb, hr, yu, nu = S.stages[i]
        dyu, G["up%d.1.weight" % i], G["up%d.1.bias" % i] = ops.groupnorm_bwd(dh, yu, nu[0], nu[1], _d(up[1].weight), nu[2],
                                                                              nu[3], up[1].num_groups, act=SW)
        Ci, Co = up[0].weight.shape[0], up[0].weight.shape[1]
        # ConvTranspose backward = the mirror convolution (4x4 s2 p1) forward / wgrad with the roles swapped
        G["up%d.0.weight" % i] = ops.unpack_conv_weight(ops.conv2d_wgrad(hr, dyu, 4, 4, 2, 1), Ci, Co, 4, 4)
        G["up%d.0.bias" % i] = ops.colsum(_c2(dyu))
        dhr = ops.conv2d_fwd(dyu, pk.conv(up[0]), None, 2, 1)
        dh, Grb = resblock_bwd(Srb, dhr, rb, pk)
        _pref(Grb, "res%d." % i, G)
    B = dh.shape[0]
    fc = dec.fc
    dh3 = ops.nhwc_to_nchw(dh).view(B, -1)
    dg3, G["fc.4.weight"], G["fc.4.bias"] = ops.layernorm_bwd(dh3, S.g3, _d(fc[4].weight), _d(fc[4].bias), None, S.m3, S.r3,
                                                              act=SW)
    G["fc.3.weight"] = ops.linear_wgrad(dg3, S.h0)
    G["fc.3.bias"] = ops.colsum(dg3)
    dh0 = ops.linear_dgrad(dg3, _d(fc[3].weight))
    dg0, G["fc.1.weight"], G["fc.1.bias"] = ops.layernorm_bwd(dh0, S.g0, _d(fc[1].weight), _d(fc[1].bias), None, S.m0, S.r0,
                                                              act=SW)
    G["fc.0.weight"] = ops.linear_wgrad(dg0, S.z)
    G["fc.0.bias"] = ops.colsum(dg0)
    dz = ops.linear_dgrad(dg0, _d(fc[0].weight)) if need_dz else None
    return G, dz


# ============================================================================= classifier (v2:311-321)
def classifier_fwd(ae, z, m1=None, m2=None, save=True):
    c = ae.classifier
    S = RBSaved()
    S.z = z
    S.g0 = ops.linear(z, _d(c[0].weight), _d(c[0].bias))
    S.h0, S.mean0, S.rstd0 = ops.layernorm_fwd(S.g0, _d(c[1].weight), _d(c[1].bias), eps=c[1].eps, act=SW, save=save)
    S.m1, S.m2 = m1, m2
    S.h0m = ops.mul(S.h0, m1) if m1 is not None else S.h0  # Dropout AFTER Swish (v2:314-315)
    S.g4 = ops.linear(S.h0m, _d(c[4].weight), _d(c[4].bias))
    S.h4, S.mean4, S.rstd4 = ops.layernorm_fwd(S.g4, _d(c[5].weight), _d(c[5].bias), eps=c[5].eps, act=SW, save=save)
    S.h4m = ops.mul(S.h4, m2) if m2 is not None else S.h4
    return ops.linear(S.h4m, _d(c[8].weight), _d(c[8].bias)), S


def classifier_bwd(ae, S, dl, need_dz=True):
    c = ae.classifier
    G = {}
    G["8.weight"], G["8.bias"] = ops.linear_wgrad(dl, S.h4m), ops.colsum(dl)
    dh4 = ops.linear_dgrad(dl, _d(c[8].weight))
    if S.m2 is not None:
        dh4 = ops.mul(dh4, S.m2)
    dg4, G["5.weight"], G["5.bias"] = ops.layernorm_bwd(dh4, S.g4, _d(c[5].weight), _d(c[5].bias), None, S.mean4, S.rstd4, act=SW)
    G["4.weight"], G["4.bias"] = ops.linear_wgrad(dg4, S.h0m), ops.colsum(dg4)
    dh0 = ops.linear_dgrad(dg4, _d(c[4].weight))
    if S.m1 is not None:
        dh0 = ops.mul(dh0, S.m1)
    dg0, G["1.weight"], G["1.bias"] = ops.layernorm_bwd(dh0, S.g0, _d(c[1].weight), _d(c[1].bias), None, S.mean0, S.rstd0, act=SW)
    G["0.weight"], G["0.bias"] = ops.linear_wgrad(dg0, S.z), ops.colsum(dg0)
    dz = ops.linear_dgrad(dg0, _d(c[0].weight)) if need_dz else None
    return G, dz


def classify(ae, z, masks=(None, None)):
    m1, m2 = masks
    z = _f32c(z)
    if ae.training and m1 is None:
        c = ae.classifier
        m1 = ae.rng.dropout((z.shape[0], 512), z.device, c[3].p)
        m2 = ae.rng.dropout((z.shape[0], 256), z.device, c[7].p)
    if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in ae.classifier.parameters())):
        names = [n for n, _ in ae.classifier.named_parameters()]
        return ClassifierFn.apply(ae, names, m1, m2, z, *[p for _, p in ae.classifier.named_parameters()])
    out, _ = classifier_fwd(ae, z, m1, m2, save=False)
    return out


# ============================================================================= VGG16[:16] (v2:962-981)
def _vgg_consts(vgg):
    def mk():
        dev = vgg.mean.device
        sc = torch.zeros(4, dtype=torch.float32, device=dev)
        sh = torch.zeros(4, dtype=torch.float32, device=dev)
        inv = (1.0 / vgg.std.detach().double().view(-1)).float()       # host-side constants (3 numbers)
        sc[:3] = inv
        sh[:3] = (-vgg.mean.detach().double().view(-1) / vgg.std.detach().double().view(-1)).float()
        return sc, sh, torch.zeros(4, dtype=torch.float32, device=dev)

    return vgg._pk.get(("norm",), [], mk)


def vgg_features(vgg, x, save):
    """x NCHW (B,3,H,W) in [0,1] -> NHWC features; conv3x3 + bias + ReLU fused in the GEMM epilogue."""
    pk = vgg._pk
    sc, sh, _ = _vgg_consts(vgg)
    x4 = ops.nchw_to_nhwc(x, 4)
    h = ops.affine_act(x4, sc, sh)  # (x - mean)/std, pad channel stays 0
    tape = []
    for layer in vgg.feature_extractor:
        if isinstance(layer, torch.nn.Conv2d):
            hin = h
            h = conv3_fwd_act(hin, layer, pk, RELU)
            if save:
                tape.append(("conv", layer, hin, h))
        elif isinstance(layer, torch.nn.MaxPool2d):
            hin = h
            h = ops.maxpool2_fwd(hin)
            if save:
                tape.append(("pool", None, hin, h))
    return h, tape


def vgg_features_pair(vgg, x, y):
    """features of x (with the tape its input gradient needs) and of y (none: the target of the perceptual loss, v2:976-980) from
    ONE batch-stacked pass over [x | y] - the VGG is frozen and has no batch statistics, so stacking changes nothing but the
    launch count and the tile fill (and, in f16x2 math, the shared power-of-two operand scale: the larger of the two ranges).
    -> fx, fy, tape (views of the first half)."""
    B = x.shape[0]
    f, tape2 = vgg_features(vgg, torch.cat((x, y), 0), True)
    tape = [(kind, layer, hin[:B], hout[:B]) for kind, layer, hin, hout in tape2]
    return f[:B], f[B:], tape


def vgg_features_bwd(vgg, tape, dfeat):
    """input gradient only (the VGG is frozen, v2:966-967)."""
    pk = vgg._pk
    d = dfeat
    for kind, layer, hin, hout in reversed(tape):
        if kind == "pool":
            d = ops.maxpool2_bwd(d, hin)
        else:
            dpre = ops.act_bwd(d, hout, RELU)  # relu' from the output sign
            d = conv3_dgrad(dpre, layer, (hin.shape[1], hin.shape[2]), pk)
    sc, _, zero = _vgg_consts(vgg)
    return ops.nhwc_to_nchw(ops.affine_act(d, sc, zero), 3)


def vgg_loss(vgg, x, y):
    x, y = _f32c(x), _f32c(y)
    if torch.is_grad_enabled() and x.requires_grad:
        return VGGLossFn.apply(vgg, x, y)
    fx, _ = vgg_features(vgg, x, False)
    fy, _ = vgg_features(vgg, y, False)
    loss, _ = ops.euclid_loss_fwd(fx, fy, 1)
    return loss.view(())


class VGGLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vgg, x, y):
        fx, tape = vgg_features(vgg, x, True)
        fy, _ = vgg_features(vgg, y, False)
        loss, norms = ops.euclid_loss_fwd(fx, fy, 1)
        ctx.vgg, ctx.tape, ctx.fx, ctx.fy, ctx.norms = vgg, tape, fx, fy, norms
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        dfx = ops.euclid_loss_bwd(ctx.fx, ctx.fy, ctx.norms, 1.0, _f32c(g).view(1), 1)
        dx = vgg_features_bwd(ctx.vgg, ctx.tape, dfx)
        ctx.tape = None
        return None, dx, None


# ============================================================================= Discriminator64 (v2:983-1007)
def disc_fwd(dm, x, save=True, sync=None, world=1):
    """x NCHW (B,3,64,64) -> probabilities (B,). BatchNorm uses batch statistics when dm.training (and updates the
    running buffers exactly like nn.BatchNorm2d), running statistics otherwise. sync/world: the data-parallel step's
    all-reduce hook (ops.batchnorm_stats) -> statistics of the global batch; disc_bwd reuses it from the tape."""
    pk = dm._pk
    m = dm.model
    S = RBSaved()
    S.x4 = ops.nchw_to_nhwc(x, 4)
    S.y0 = ops.conv2d_fwd(S.x4, pk.conv(m[0]), _d(m[0].bias), 2, 1, act=LRELU, act_param=0.2)
    h = S.y0
    S.blocks = []
    S.sync, S.world = (sync if dm.training else None), world
    for ci, bi in ((2, 3), (5, 6), (8, 9)):
        conv, bn = m[ci], m[bi]
        hin = h
        yc = ops.conv2d_fwd(hin, pk.conv(conv), _d(conv.bias), 2, 1)
        nb = ops.batchnorm_stats(yc, _d(bn.weight), _d(bn.bias), bn.running_mean, bn.running_var, dm.training,
                                 bn.momentum, bn.eps, sync=sync, world=world)
        if dm.training:
            bn.num_batches_tracked += 1
        h = ops.affine_act(yc, nb[0], nb[1], act=LRELU, act_param=0.2)
        S.blocks.append((hin, yc, nb))
    B = x.shape[0]
    S.hf = h.view(B, -1)  # (B, 4*4*512) NHWC-flat == the packed weight's [kh][kw][Cin] order
    S.wf = pk.conv(m[11]).view(1, -1)
    S.logit = ops.linear(S.hf, S.wf, _d(m[11].bias))
    p = ops.act_fwd(S.logit, SIG).view(-1)
    return p, (S if save else None)


def disc_bwd(dm, S, dp, need_wgrad=True, need_dx=True):
    pk = dm._pk
    m = dm.model
    G = {}
    dlogit = ops.act_bwd(_f32c(dp).view(-1, 1), S.logit, SIG)
    if need_wgrad:
        G["model.11.weight"] = ops.unpack_conv_weight(ops.linear_wgrad(dlogit, S.hf).view(1, 4, 4, 512), 1, 512, 4, 4)
        G["model.11.bias"] = ops.colsum(dlogit)
    dh = ops.linear_dgrad(dlogit, S.wf).view(-1, 4, 4, 512)
    for (ci, bi), (hin, yc, nb) in zip(((8, 9), (5, 6), (2, 3)), reversed(S.blocks)):
        conv, bn = m[ci], m[bi]
        dyc, dw, db = ops.batchnorm_bwd(dh, yc, nb[0], nb[1], _d(bn.weight), nb[2], nb[3], act=LRELU, act_param=0.2,
                                        sync=S.sync, world=S.world)
        if need_wgrad:
            Co, Cin = conv.weight.shape[0], conv.weight.shape[1]
            G["model.%d.weight" % bi], G["model.%d.bias" % bi] = dw, db
            G["model.%d.weight" % ci] = ops.unpack_conv_weight(ops.conv2d_wgrad(dyc, hin, 4, 4, 2, 1), Co, Cin, 4, 4)
            G["model.%d.bias" % ci] = ops.colsum(_c2(dyc))
        dh = ops.conv2d_dgrad(dyc, pk.conv(conv), (hin.shape[1], hin.shape[2]), 2, 1)
    dy0 = ops.act_bwd(dh, S.y0, LRELU, 0.2)  # leaky-relu' from the output sign
    if need_wgrad:
        G["model.0.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy0, S.x4, 4, 4, 2, 1), 64, 3, 4, 4)
        G["model.0.bias"] = ops.colsum(_c2(dy0))
    dx = ops.nhwc_to_nchw(ops.conv2d_dgrad(dy0, pk.conv(m[0]), (64, 64), 2, 1), 3) if need_dx else None
    return G, dx


def disc_fwd_pair(dm, xa, xb, sync=None, world=1):
    """The discriminator update's two passes (v2:1097-1098: D(real), D(fake.detach())) as ONE batch-stacked pass: every conv, the
    final linear and the activations run once on [xa | xb]; BatchNorm keeps per-pass statistics (ops.batchnorm_stats_pair: each half
    normalised by its own batch statistics, running buffers updated real-then-fake as two passes do), and under data parallelism
    the two passes' SyncBatchNorm sums share one all-reduce per layer. Training mode only. -> (p_a, p_b), tape for disc_bwd_pair."""
    assert dm.training, "disc_fwd_pair is the training-mode discriminator update"
    pk = dm._pk
    m = dm.model
    B = xa.shape[0]
    S = RBSaved()
    S.B = B
    S.x4 = ops.nchw_to_nhwc(torch.cat((xa, xb), 0), 4)
    S.y0 = ops.conv2d_fwd(S.x4, pk.conv(m[0]), _d(m[0].bias), 2, 1, act=LRELU, act_param=0.2)
    h = S.y0
    S.blocks = []
    S.sync, S.world = sync, world
    for ci, bi in ((2, 3), (5, 6), (8, 9)):
        conv, bn = m[ci], m[bi]
        hin = h
        yc = ops.conv2d_fwd(hin, pk.conv(conv), _d(conv.bias), 2, 1)
        nba, nbb = ops.batchnorm_stats_pair(yc, _d(bn.weight), _d(bn.bias), bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                            sync=sync, world=world)
        bn.num_batches_tracked += 2
        h = torch.empty_like(yc)
        ops.affine_act(yc[:B], nba[0], nba[1], act=LRELU, act_param=0.2, out=h[:B])
        ops.affine_act(yc[B:], nbb[0], nbb[1], act=LRELU, act_param=0.2, out=h[B:])
        S.blocks.append((hin, yc, nba, nbb))
    S.hf = h.view(2 * B, -1)
    S.wf = pk.conv(m[11]).view(1, -1)
    S.logit = ops.linear(S.hf, S.wf, _d(m[11].bias))
    p = ops.act_fwd(S.logit, SIG).view(-1)
    return (p[:B], p[B:]), S


def disc_bwd_pair(dm, S, dpa, dpb):
    """Weight gradients of both passes of disc_fwd_pair, SUMMED (what set_grads(real) + add_grads(fake) accumulated): the weight-
    gradient GEMMs contract over the stacked batch, the BatchNorm backward works per half (one exchange for both under `sync`)."""
    pk = dm._pk
    m = dm.model
    B = S.B
    G = {}
    dp = torch.cat((_f32c(dpa).view(-1), _f32c(dpb).view(-1)), 0).view(-1, 1)
    dlogit = ops.act_bwd(dp, S.logit, SIG)
    G["model.11.weight"] = ops.unpack_conv_weight(ops.linear_wgrad(dlogit, S.hf).view(1, 4, 4, 512), 1, 512, 4, 4)
    G["model.11.bias"] = ops.colsum(dlogit)
    dh = ops.linear_dgrad(dlogit, S.wf).view(-1, 4, 4, 512)
    for (ci, bi), (hin, yc, nba, nbb) in zip(((8, 9), (5, 6), (2, 3)), reversed(S.blocks)):
        conv, bn = m[ci], m[bi]
        dyc, dw, db = ops.batchnorm_bwd_pair(dh, yc, nba, nbb, _d(bn.weight), act=LRELU, act_param=0.2, sync=S.sync, world=S.world)
        Co, Cin = conv.weight.shape[0], conv.weight.shape[1]
        G["model.%d.weight" % bi], G["model.%d.bias" % bi] = dw, db
        G["model.%d.weight" % ci] = ops.unpack_conv_weight(ops.conv2d_wgrad(dyc, hin, 4, 4, 2, 1), Co, Cin, 4, 4)
        G["model.%d.bias" % ci] = ops.colsum(_c2(dyc))
        dh = ops.conv2d_dgrad(dyc, pk.conv(conv), (hin.shape[1], hin.shape[2]), 2, 1)
    dy0 = ops.act_bwd(dh, S.y0, LRELU, 0.2)
    G["model.0.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy0, S.x4, 4, 4, 2, 1), 64, 3, 4, 4)
    G["model.0.bias"] = ops.colsum(_c2(dy0))
    return G


def discriminator(dm, x):
    x = _f32c(x)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in dm.parameters())):
        names = [n for n, _ in dm.named_parameters()]
        return DiscFn.apply(dm, names, x, *[p for _, p in dm.named_parameters()])
    p, _ = disc_fwd(dm, x, save=False)
    return p


# ============================================================================= autograd wrappers
def _grads(G, names):
    return tuple(G.get(n) for n in names)


class DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dm, names, x, *params):
        p, S = disc_fwd(dm, x, save=True)
        ctx.dm, ctx.S, ctx.names = dm, S, names
        return p

    @staticmethod
    def backward(ctx, dp):
        need_w = any(ctx.needs_input_grad[3:])
        G, dx = disc_bwd(ctx.dm, ctx.S, dp, need_wgrad=need_w, need_dx=ctx.needs_input_grad[2])
        ctx.S = None
        return (None, None, dx) + _grads(G, ctx.names)


class ClassifierFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ae, names, m1, m2, z, *params):
        out, S = classifier_fwd(ae, z, m1, m2)
        ctx.ae, ctx.S, ctx.names = ae, S, names
        return out

    @staticmethod
    def backward(ctx, dl):
        G, dz = classifier_bwd(ctx.ae, ctx.S, _f32c(dl), need_dz=ctx.needs_input_grad[4])
        ctx.S = None
        return (None, None, None, None, dz) + _grads(G, ctx.names)


def resblock_params(rb):
    return [p for _, p in rb.named_parameters()]


class ResidualBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rb, x, *params):
        out, S = resblock_fwd(ops.nchw_to_nhwc(_f32c(x)), rb, rb._pk)
        ctx.rb, ctx.S = rb, S
        return ops.nhwc_to_nchw(out)

    @staticmethod
    def backward(ctx, dout):
        dx, G = resblock_bwd(ctx.S, ops.nchw_to_nhwc(_f32c(dout)), ctx.rb, ctx.rb._pk, need_dx=True)
        ctx.S = None
        names = [n for n, _ in ctx.rb.named_parameters()]
        return (None, ops.nhwc_to_nchw(dx)) + _grads(G, names)


def encoder_params(enc):
    return [p for _, p in enc.named_parameters()]


class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, x, *params):
        mu, logvar, feats, S = encoder_fwd(enc, _f32c(x))
        enc.skip_features = [f.permute(0, 3, 1, 2) for f in feats]
        ctx.enc, ctx.S = enc, S
        return mu, logvar

    @staticmethod
    def backward(ctx, dmu, dlogvar):
        S = ctx.S
        B = S.x4.shape[0]
        if dmu is None:
            dmu = torch.zeros(B, ctx.enc.latent_dim, device=S.x4.device)
        if dlogvar is None:
            dlogvar = torch.zeros(B, ctx.enc.latent_dim, device=S.x4.device)
        G, dx = encoder_bwd(ctx.enc, S, _f32c(dmu), _f32c(dlogvar), need_dx=ctx.needs_input_grad[1])
        ctx.S = None
        names = [n for n, _ in ctx.enc.named_parameters()]
        return (None, dx) + _grads(G, names)


def decoder_params(dec):
    return [p for _, p in dec.named_parameters()]


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, z, *params):
        out, S = decoder_fwd(dec, _f32c(z))
        ctx.dec, ctx.S = dec, S
        return out

    @staticmethod
    def backward(ctx, dout):
        G, dz = decoder_bwd(ctx.dec, ctx.S, _f32c(dout), need_dz=ctx.needs_input_grad[1])
        ctx.S = None
        names = [n for n, _ in ctx.dec.named_parameters()]
        return (None, dz) + _grads(G, names)


# ---- stand-alone CALayer / SpatialAttention (never used alone by v2, kept for surface completeness) -------------
class SpatialAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sa, x, w):
        xh = ops.nchw_to_nhwc(_f32c(x))
        saw = w.detach().contiguous()
        m = ops.res_tail_map(xh)
        out, att = ops.res_tail_out(xh, None, None, None, m, saw, save_att=True)
        ctx.saved = (xh, m, att, saw)
        return ops.nhwc_to_nchw(out)

    @staticmethod
    def backward(ctx, dout):
        xh, m, att, saw = ctx.saved
        dw, da = ops.res_tail_bwd_a(ops.nchw_to_nhwc(_f32c(dout)), xh, None, None, None, att, None, final_swish=False)
        dmap, dsaw = ops.sa_conv_bwd(da, m, saw)
        dx = ops.res_tail_bwd_c(dw, xh, None, None, None, att, dmap)
        return None, ops.nhwc_to_nchw(dx), dsaw


class CALayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ca, x, w1, w2):
        xh = ops.nchw_to_nhwc(_f32c(x))
        _, _, mean, _ = ops.instnorm_stats(xh, None, None)
        s, saved = _ca_mlp_fwd(ca, mean)
        ctx.saved = (xh, s, saved)
        return ops.nhwc_to_nchw(ops.res_tail_out(xh, None, None, s))

    @staticmethod
    def backward(ctx, dout):
        xh, s, saved = ctx.saved
        B, H, W, C = xh.shape
        dh = ops.nchw_to_nhwc(_f32c(dout))
        dxs = ops.res_tail_bwd_c(dh, xh, None, None, s, None, None)  # dout * s
        # ds[b,c] = sum_pix dout*x : per-image channel sums through the norm-backward statistics pass
        ones = torch.ones(B, C, device=xh.device)
        zeros = torch.zeros(B, C, device=xh.device)
        ds = ops.chan_dot(dh, xh)
        dw1, dw2, dpooled = _ca_mlp_bwd(saved, ds)
        dx = ops.affine_act(dxs, ones, ops.axpby(1.0 / (H * W), dpooled, 0.0, zeros))
        return None, ops.nhwc_to_nchw(dx), dw1, dw2
