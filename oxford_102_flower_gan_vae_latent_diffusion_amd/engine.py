"""Functional forward/backward compositions of the v2 module surface over the C-ABI ops.

Everything here works on NHWC fp32 device tensors and on the *parameter containers* of
modules.py (real nn.Conv2d / nn.Linear / ... objects whose .weight/.bias carry the reference's
state-dict keys); no torch arithmetic is executed — only libflowers_hip.so kernels.
"""
import os

import torch

from . import ops

SW = ops.ACT_SWISH
# f16x2 split math for the 3x3 ResidualBlock convs of the inference paths (FL_CONV_F16X2=0: the r01 bf16x3 kernels)
USE_F16X2 = os.environ.get("FL_CONV_F16X2", "1") != "0"
# FL_FUSED_HEADS=0: the encoder heads' tail and the DDPM step's draws as separate launches (the round-2 chain; same numbers)
FUSED_HEADS = os.environ.get("FL_FUSED_HEADS", "1") != "0"


class PackCache:
    """Derived weight layouts (packed conv weights), keyed by parameter version; never saved."""

    def __init__(self):
        self._c = {}
        self._keep = []  # entries replaced during a graph capture (the graph may hold their pointers)
        self.epoch = 0  # bumped by optimizers that update parameters through raw pointers

    def get(self, key, params, fn):
        ver = tuple((p._version, p.data_ptr()) for p in params) + (self.epoch,)
        ent = self._c.get(key)
        if ent is None or ent[0] != ver:
            if ent is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                # a hipGraph being captured may already read the entry this one replaces: keep it alive with the cache
                self._keep.append(ent)
            with torch.no_grad():
                ent = (ver, fn())
            self._c[key] = ent
        return ent[1]

    def conv(self, conv):
        """nn.Conv2d (O,I,kh,kw) -> [O][kh][kw][Ipad]; nn.ConvTranspose2d (I,O,4,4) -> [I][4][4][O]."""
        w = conv.weight
        ipad = (w.shape[1] + 3) // 4 * 4
        return self.get(("conv", id(conv)), [w], lambda: ops.pack_conv_weight(w.detach(), ipad))


    def conv_split(self, conv):
        """nn.Conv2d weight split once into its three bf16 pieces in MFMA-fragment order (ops.split_pack_weight) for the
        inference convs of the split GEMM math -> (pieces, (O, kh, kw, Ipad))."""
        w = conv.weight

        def make():
            wp = self.conv(conv)
            return ops.split_pack_weight(wp), tuple(wp.shape)

        return self.get(("convsplit", id(conv)), [w], make)


    def conv_h2(self, conv):
        """nn.Conv2d weight scaled, split into two fp16 pieces and laid out in MFMA-fragment order once
        (ops.pack_weight_f16x2) for the f16x2 inference convs -> (pack, (O, kh, kw, Ipad))."""
        w = conv.weight

        def make():
            wp = self.conv(conv)
            return ops.pack_weight_f16x2(wp), tuple(wp.shape)

        return self.get(("convh2", id(conv)), [w], make)

    def conv_h2_dgrad(self, conv):
        """f16x2 pack of the MIRROR convolution of a 3x3 / stride 1 / pad 1 nn.Conv2d: its input gradient is the same
        conv forward on dy with the taps flipped and the channel roles swapped -> (pack, (I, 3, 3, Opad))."""
        w = conv.weight

        def make():
            wt = w.detach().flip(2, 3).transpose(0, 1).contiguous()      # (I, O, 3, 3): a re-layout, no arithmetic
            wp = ops.pack_conv_weight(wt, (wt.shape[1] + 3) // 4 * 4)
            return ops.pack_weight_f16x2(wp), tuple(wp.shape)

        return self.get(("convh2d", id(conv)), [w], make)

    def ln_bound(self, ln, n, prev=None):
        """(1,) device float bounding |ln(.)| (and what Swish / (0,1) gates make of it) for a norm over n elements, plus
        the bound of a residual input given as prev = (its norm module, its n): the range information of the f16x2 convs.
        Keyed by the producing MODULES and versioned by every weight of the chain (keying on id() of the cached
        `prev` tensor grew the cache by one entry per weight version and could hit a stale entry through a recycled id)."""
        params = [ln.weight, ln.bias]
        key = ("lnbound", id(ln), n)
        if prev is not None:
            pln, pn = prev
            params += [pln.weight, pln.bias]
            key += (id(pln), pn)

        def make():
            pb = self.ln_bound(prev[0], prev[1]) if prev is not None else None
            return ops.affine_bound(ln.weight.detach(), ln.bias.detach(), float(n) ** 0.5, pb)

        return self.get(key, params, make)


def _d(p):
    return p.detach() if p is not None else None


def _f16x2_usable(x, conv, stride, pad, x_bound):
    Cout, _, kh, kw = conv.weight.shape
    return (USE_F16X2 and x_bound is not None and ops.get_gemm_math() == "split"
            and ops.f16x2_ok(x, Cout, kh, kw, stride, pad))


def conv_ln_infer(x, conv, ln, stride, pad, pk, x_bound=None, pre=None):
    """inference conv + fused LayerNorm2d statistics -> (y, scale, shift); weights pre-split when the layer qualifies.
    x_bound: (1,) device float >= max|input| (PackCache.ln_bound) - enables the f16x2 kernels.
    pre = (scale, shift): the conv's input is swish(x * scale[b,c] + shift[b,c]) - fused into the f16x2 kernels' halo
    staging, materialised by fl_affine_act for the others."""
    B, H, W, Cin = x.shape
    Cout, _, kh, kw = conv.weight.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if _f16x2_usable(x, conv, stride, pad, x_bound):
        wh, geom = pk.conv_h2(conv)
        y, (sc, sh, _, _) = ops.conv2d_fwd_f16x2(x, wh, geom, _d(conv.bias), stride, pad, ln_w=_d(ln.weight), ln_b=_d(ln.bias),
                                                 eps=ln.eps, x_amax=x_bound, pre=pre)
        return y, sc, sh
    if pre is not None:
        x = ops.affine_act(x, pre[0], pre[1], act=SW)
    if ops.get_gemm_math() == "split" and ops.presplit_ok(Cin, Cout, Ho * Wo):
        ws, geom = pk.conv_split(conv)
        y, (sc, sh, _, _) = ops.conv2d_fwd_presplit(x, ws, geom, _d(conv.bias), stride, pad, ln_w=_d(ln.weight),
                                                    ln_b=_d(ln.bias), eps=ln.eps)
    else:
        y, (sc, sh, _, _) = ops.conv2d_fwd_instnorm(x, pk.conv(conv), _d(conv.bias), stride, pad, _d(ln.weight), _d(ln.bias),
                                                    ln.eps)
    return y, sc, sh


# ----------------------------------------------------------------------------- VAE blocks (inference)
def ca_gate(ca, pooled):
    """CALayer MLP (v2:57-62) on pooled (R,C) -> sigmoid gate (R,C)."""
    w1 = _d(ca.conv_du[0].weight).view(ca.conv_du[0].weight.shape[0], -1)
    w2 = _d(ca.conv_du[2].weight).view(ca.conv_du[2].weight.shape[0], -1)
    h = ops.linear(pooled, w1, None, act=SW)
    return ops.linear(h, w2, None, act=ops.ACT_SIGMOID)


def resblock_infer(x, rb, pk, x_bound=None, x_pre=None):
    """ResidualBlock.forward (v2:170-178), NHWC, no saved state.
    x_bound: (1,) device float >= max|block input| or None (then the range-agnostic bf16x3 conv kernels run).
    x_pre = (scale, shift): the block input is swish(x * scale[b,c] + shift[b,c]) and is never materialised - conv1
    applies it while staging, the residual add re-applies it (engine.encoder_infer hands the down conv's raw output)."""
    HW = x.shape[1] * x.shape[2]
    y1, sc, sh = conv_ln_infer(x, rb.conv1, rb.ln1, 1, 1, pk, x_bound, x_pre)
    b1 = pk.ln_bound(rb.ln1, HW) if x_bound is not None else None
    if _f16x2_usable(y1, rb.conv2, 1, 1, b1):  # LayerNorm2d-apply + Swish of v2:163-164 fused into conv2's staging
        y2, sc2, sh2 = conv_ln_infer(y1, rb.conv2, rb.ln2, 1, 1, pk, b1, (sc, sh))
    else:
        a1 = ops.affine_act(y1, sc, sh, act=SW, out=y1)
        y2, sc2, sh2 = conv_ln_infer(a1, rb.conv2, rb.ln2, 1, 1, pk, b1)
    # avg_pool(LayerNorm2d(y)) == ln2.bias exactly (normalised maps have zero spatial mean), so the
    # channel-attention gate (v2:65-66) is one (C,) vector shared by the batch: no pooling pass.
    # ... and a function of the weights only: computed once per weight version, not per batch.
    s = pk.get(("cagate", id(rb)), [rb.ca.conv_du[0].weight, rb.ca.conv_du[2].weight, rb.ln2.bias],
               lambda: ca_gate(rb.ca, _d(rb.ln2.bias).view(1, -1)).view(-1))
    m = ops.res_tail_map(y2, sc2, sh2, s)
    return ops.res_tail_out(y2, sc2, sh2, s, m, _d(rb.sa.conv.weight).contiguous(), xres=x, final_swish=True, xres_pre=x_pre)


def _head(fc, flat, w0=None):
    """Linear -> LayerNorm -> Swish -> Linear (v2:210-221). w0: fc[0].weight re-laid out for an NHWC-flattened input."""
    g = ops.linear(flat, w0 if w0 is not None else _d(fc[0].weight), _d(fc[0].bias))
    g, _, _ = ops.layernorm_fwd(g, _d(fc[1].weight), _d(fc[1].bias), eps=fc[1].eps, act=SW, save=False)
    return ops.linear(g, _d(fc[3].weight), _d(fc[3].bias))


def encoder_infer(enc, x, want_feats=True):
    """Encoder.forward (v2:223-239). x (B,3,64,64) NCHW -> mu, logvar (B,latent), [NHWC skip features] (v2:224-235;
    want_feats=False: the list holds None where an activation was fused away - the decoder ignores it, v2:280)."""
    return _encoder_infer(enc, x, want_feats)[:3]


def encoder_latent(enc, x, rng):
    """z = reparameterize(*Encoder.forward(x)) (v2:1257-1259 / v2:339-343) with the N(0,1) draws taken from `rng` (rng.Philox):
    where the heads run through the fused tail kernel the draws are made inside it, otherwise by rng.normal - the same numbers."""
    mu, logvar, _, z = _encoder_infer(enc, x, False, rng)
    if z is None:
        z, _ = ops.reparameterize(mu, logvar, rng.normal(mu.shape, mu.device))
    return z


def _encoder_infer(enc, x, want_feats=True, reparam_rng=None):
    pk = enc._pk
    B = x.shape[0]
    h = ops.nchw_to_nhwc(x, 4)
    c0, ln0 = enc.initial_conv[0], enc.initial_conv[1]
    y, (sc, sh, _, _) = ops.conv2d_fwd_instnorm(h, pk.conv(c0), _d(c0.bias), 1, 1, _d(ln0.weight), _d(ln0.bias), ln0.eps)
    # range information of the f16x2 convs: |swish(ln(.))| <= the LayerNorm2d bound; a ResidualBlock adds its input's bound
    bound = pk.ln_bound(ln0, y.shape[1] * y.shape[2])
    # the LayerNorm2d-apply + Swish passes (v2:187-188, v2:193-194, ...) are fused into the consuming conv's halo staging:
    # `h` stays the raw conv output and `pre` its (scale, shift) until something needs the activation itself
    pre = (sc, sh)
    h = y
    feats = [ops.affine_act(y, sc, sh, act=SW) if want_feats else None]
    for i in (1, 2, 3):
        dn, rb = getattr(enc, "down%d" % i), getattr(enc, "res%d" % i)
        y, sc, sh = conv_ln_infer(h, dn[0], dn[1], 2, 1, pk, bound, pre)
        HW = y.shape[1] * y.shape[2]
        bd = pk.ln_bound(dn[1], HW)
        h = resblock_infer(y, rb, pk, bd, (sc, sh))
        pre = None
        bound = pk.ln_bound(rb.ln2, HW, (dn[1], HW))
        feats.append(h)
    # x.view(B,-1) of the NCHW tensor (v2:236): instead of transposing 33 MB of activations per batch, the columns of the
    # two 512 x 32768 head weights are permuted to NHWC order once per weight version
    Hh, Wh, Ch = h.shape[1], h.shape[2], h.shape[3]
    flat = h.view(B, -1)

    def perm(fc):
        w = fc[0].weight
        return pk.get(("fc_nhwc", id(fc)), [w], lambda: ops.nchw_to_nhwc(w.detach().view(-1, Ch, Hh, Wh)).view(w.shape[0], -1))

    N1 = enc.fc_mu[0].weight.shape[0]
    if (USE_F16X2 and bound is not None and ops.get_gemm_math() == "split" and enc.fc_logvar[0].weight.shape[0] == N1
            and ops.linear_f16x2_ok(B, 2 * N1, flat.shape[1])):
        # both heads' first Linear (2 x 512 x 32768, the longest reduction of the step) as ONE f16x2 split-K GEMM on the
        # concatenated, NHWC-permuted, pre-packed weight; x range = the bound of the last ResidualBlock's output
        def make():
            w = torch.cat([perm(enc.fc_mu), perm(enc.fc_logvar)], 0).contiguous()
            b = torch.cat([enc.fc_mu[0].bias.detach(), enc.fc_logvar[0].bias.detach()]).contiguous()
            return ops.pack_weight_f16x2(w), b

        wh, bcat = pk.get(("fc_heads_h2",), [enc.fc_mu[0].weight, enc.fc_logvar[0].weight, enc.fc_mu[0].bias,
                                              enc.fc_logvar[0].bias], make)
        g2 = ops.linear_fwd_f16x2(flat, wh, 2 * N1, bcat, n_block=N1, x_amax=bound)
        fm, fv = enc.fc_mu, enc.fc_logvar
        if (FUSED_HEADS and ops.encoder_heads_ok(B, N1, fm[3].weight.shape[0]) and fm[3].weight.shape == fv[3].weight.shape
                and fm[1].eps == fv[1].eps):
            # LayerNorm -> Swish -> Linear of both heads (and the reparameterisation draws, when asked) in ONE launch
            mu, lv, z = ops.encoder_heads_fwd(
                g2, (_d(fm[1].weight), _d(fm[1].bias)), (_d(fv[1].weight), _d(fv[1].bias)), (_d(fm[3].weight), _d(fm[3].bias)),
                (_d(fv[3].weight), _d(fv[3].bias)), eps=fm[1].eps, want_stats=reparam_rng is None,
                rng_seed=reparam_rng.seed if reparam_rng is not None else None,
                rng_state=reparam_rng.state(x.device) if reparam_rng is not None else None)
            return mu, lv, feats, z
        outs = []
        for hi, fc in enumerate((fm, fv)):
            gh, _, _ = ops.layernorm_fwd(g2[hi], _d(fc[1].weight), _d(fc[1].bias), eps=fc[1].eps, act=SW, save=False)
            outs.append(ops.linear(gh, _d(fc[3].weight), _d(fc[3].bias)))
        return outs[0], outs[1], feats, None
    return _head(enc.fc_mu, flat, perm(enc.fc_mu)), _head(enc.fc_logvar, flat, perm(enc.fc_logvar)), feats, None


def decoder_infer(dec, z):
    """Decoder.forward (v2:280-290). z (B,latent) -> (B,3,64,64) NCHW in [0,1]."""
    pk = dec._pk
    B = z.shape[0]
    fc = dec.fc
    h = ops.linear(z, _d(fc[0].weight), _d(fc[0].bias))
    h, _, _ = ops.layernorm_fwd(h, _d(fc[1].weight), _d(fc[1].bias), eps=fc[1].eps, act=SW, save=False)
    h = ops.linear(h, _d(fc[3].weight), _d(fc[3].bias))
    h, _, _ = ops.layernorm_fwd(h, _d(fc[4].weight), _d(fc[4].bias), eps=fc[4].eps, act=SW, save=False)
    h = ops.nchw_to_nhwc(h.view(B, 512, 8, 8))  # view(-1,512,8,8) (v2:282) -> NHWC
    bound = pk.ln_bound(fc[4], fc[4].normalized_shape[0])  # |swish(LayerNorm(.))|: range information of the f16x2 convs
    for i in (3, 2, 1):
        rb, up = getattr(dec, "res%d" % i), getattr(dec, "up%d" % i)
        h = resblock_infer(h, rb, pk, bound)
        y = ops.convT2d_fwd(h, pk.conv(up[0]), _d(up[0].bias))
        sc, sh, _, _ = ops.groupnorm_stats(y, _d(up[1].weight), _d(up[1].bias), up[1].num_groups, up[1].eps)
        h = ops.affine_act(y, sc, sh, act=SW, out=y)
        bound = pk.ln_bound(up[1], (y.shape[3] // up[1].num_groups) * y.shape[1] * y.shape[2])
    f = dec.final_conv
    y = ops.conv2d_fwd(h, pk.conv(f[0]), _d(f[0].bias), 1, 1)
    sc, sh, _, _ = ops.groupnorm_stats(y, _d(f[1].weight), _d(f[1].bias), f[1].num_groups, f[1].eps)
    h = ops.affine_act(y, sc, sh, act=SW, out=y)
    return ops.conv2d_fwd(h, pk.conv(f[3]), _d(f[3].bias), 1, 1, act=ops.ACT_SIGMOID, out_nchw=True)


# ----------------------------------------------------------------------------- denoiser (v2:501-561)
class UNetSaved:
    __slots__ = ("x", "t", "c", "feat", "te1", "te1a", "te", "emb", "ce1", "ce1a", "ce", "tc", "layers", "hpre", "mean_f",
                 "rstd_f", "hf", "r", "shared_t", "block_masks", "head_masks")


def unet_forward(un, x, t, c, block_masks=None, head_masks=None, save=False):
    """ConditionalUNet.forward. x (B,latent); t (B,) or (1,) int64; c (B,) int64 or None.
    block_masks[i] (B,d_i) / head_masks[i] (B,heads) are dropout scale tensors (None = eval).
    nn.MultiheadAttention on the length-1 sequence of v2:550 is out_proj(v_proj(.))."""
    B = x.shape[0]
    S = UNetSaved() if save else None
    te_m, ce_m = un.time_emb, un.class_emb
    feat = ops.time_sinusoid(t, un._freqs(x.device))
    if save:
        te1 = ops.linear(feat, _d(te_m.lin1.weight), _d(te_m.lin1.bias))
        te1a = ops.act_fwd(te1, SW)
    else:
        te1, te1a = None, ops.linear(feat, _d(te_m.lin1.weight), _d(te_m.lin1.bias), act=SW)
    te = ops.linear(te1a, _d(te_m.lin2.weight), _d(te_m.lin2.bias))
    shared_t = te.shape[0] == 1 and B != 1
    emb = ce1 = ce1a = ce = None
    if c is not None:
        emb = ops.embedding_fwd(_d(ce_m.embedding.weight), c)
        if save:
            ce1 = ops.linear(emb, _d(ce_m.lin1.weight), _d(ce_m.lin1.bias))
            ce1a = ops.act_fwd(ce1, SW)
        else:
            ce1a = ops.linear(emb, _d(ce_m.lin1.weight), _d(ce_m.lin1.bias), act=SW)
        ce = ops.linear(ce1a, _d(ce_m.lin2.weight), _d(ce_m.lin2.bias))
        # W(te)+b + W(ce)+b == W(te+ce) + 2b: the class embedding goes through the *time* projection (v2:544)
        tc = ops.add_rows(ce, te) if shared_t else ops.axpby(1.0, te, 1.0, ce)
        tb = 2.0
    else:
        tc, tb = te, 1.0
    h = ops.linear(x, _d(un.latent_proj.weight), _d(un.latent_proj.bias))
    layers = []
    for i, (block, norm, down) in enumerate(un.layers):
        tp = un.time_projections[i]
        if tc.shape[0] == B:
            h1 = ops.linear(tc, _d(tp.weight), _d(tp.bias), bias_scale=tb, R=h)
        else:  # one shared row (sampling without class conditioning)
            h1 = ops.add_rows(h, ops.linear(tc, _d(tp.weight), _d(tp.bias), bias_scale=tb))
        d = h1.shape[1]
        lin, ln = block[0], block[1]
        g = ops.linear(h1, _d(lin.weight), _d(lin.bias))
        bm = block_masks[i] if block_masks is not None else None
        h2, mg, rg = ops.layernorm_fwd(g, _d(ln.weight), _d(ln.bias), mask=bm, eps=ln.eps, act=SW, save=save, R=h1)
        hn, mn, rn = ops.layernorm_fwd(h2, _d(norm.weight), _d(norm.bias), eps=norm.eps, save=save)
        att = un.attention_layers[i]
        wv, bv = _d(att.in_proj_weight)[2 * d:], _d(att.in_proj_bias)[2 * d:]
        v = ops.linear(hn, wv, bv)
        vm = ops.scale_heads(v, head_masks[i], att.num_heads) if head_masks is not None else v
        h3 = ops.linear(vm, _d(att.out_proj.weight), _d(att.out_proj.bias), R=h2)
        h = ops.linear(h3, _d(down.weight), _d(down.bias))
        if save:
            layers.append((h1, g, mg, rg, h2, mn, rn, hn, vm, h3))
    ft, fcp = un.final_time_proj, un.final_class_proj
    if shared_t:
        h = ops.add_rows(h, ops.linear(te, _d(ft.weight), _d(ft.bias)))
    else:
        h = ops.linear(te, _d(ft.weight), _d(ft.bias), R=h)
    if ce is not None:
        h = ops.linear(ce, _d(fcp.weight), _d(fcp.bias), R=h)
    hf, mf, rf = ops.layernorm_fwd(h, _d(un.final_norm.weight), _d(un.final_norm.bias), eps=un.final_norm.eps, save=save)
    r = ops.linear(x, _d(un.final.weight), _d(un.final.bias))
    o1 = ops.linear(hf, _d(un.final.weight), _d(un.final.bias))
    out = ops.gated_add(o1, r, _d(un.residual_weight).view(1))  # out + sigmoid(residual_weight)*final(x) (v2:561)
    if save:
        S.x, S.t, S.c, S.feat, S.te1, S.te1a, S.te = x, t, c, feat, te1, te1a, te
        S.emb, S.ce1, S.ce1a, S.ce, S.tc, S.layers = emb, ce1, ce1a, ce, tc, layers
        S.hpre, S.mean_f, S.rstd_f, S.hf, S.r, S.shared_t = h, mf, rf, hf, r, shared_t
        S.block_masks, S.head_masks = block_masks, head_masks
    return out, S


def unet_backward(un, S, dout, need_dx=False, gout=None):
    """Manual backward of unet_forward. Returns ({parameter name: grad tensor}, dx or None).
    `gout` {name: preallocated grad tensor} lets the trainer receive gradients directly in its flat buffer.
    Parameters the forward never touches (time_projections.4, attention_layers.4) get no gradient and the q/k
    rows of in_proj get exactly zero, like autograd on the reference."""
    if S.shared_t:
        raise RuntimeError("backward with a shared (1,) time row is not supported (training uses per-sample t)")
    G = {}

    def go(name):
        return gout.get(name) if gout is not None else None

    def wgrad(name, dy, x):
        G[name] = ops.linear_wgrad(dy, x, out=go(name))

    pending = []  # bias gradients = column sums of the layer's dy: all of them in ONE launch at the end (the tensors
    #               stay alive in this list), instead of 27 links in the dependent kernel chain of the step

    def bgrad(name, dy, alpha=1.0, out=None):
        out = out if out is not None else go(name)
        if out is None:
            out = ops.new((dy.shape[1],), dy)
        pending.append((dy, out, alpha))
        G[name] = out

    rw = _d(un.residual_weight).view(1)
    Wf = _d(un.final.weight)
    # out = final(hf) + sig*final(x): one weight, two uses
    u = ops.gated_add(S.hf, S.x, rw)                      # hf + sig*x
    wgrad("final.weight", dout, u)
    cs = ops.colsum(dout)
    G["final.bias"] = ops.gated_add(cs, cs, rw, out=go("final.bias"))   # (1+sig) * colsum
    grw = go("residual_weight")
    G["residual_weight"] = ops.gate_grad(dout, S.r, rw, out=grw.view(1) if grw is not None else None).view(())
    dhf = ops.linear_dgrad(dout, Wf)                      # also the x-branch gradient / sig
    fn = un.final_norm
    dh, G["final_norm.weight"], G["final_norm.bias"] = ops.layernorm_bwd(
        dhf, S.hpre, _d(fn.weight), _d(fn.bias), None, S.mean_f, S.rstd_f, dw_out=go("final_norm.weight"),
        db_out=go("final_norm.bias"))
    dce = None
    wgrad("final_time_proj.weight", dh, S.te)
    bgrad("final_time_proj.bias", dh)
    dte = ops.linear_dgrad(dh, _d(un.final_time_proj.weight))
    if S.ce is not None:
        wgrad("final_class_proj.weight", dh, S.ce)
        bgrad("final_class_proj.bias", dh)
        dce = ops.linear_dgrad(dh, _d(un.final_class_proj.weight))
    dtc = None
    tb = 2.0 if S.ce is not None else 1.0
    for i in reversed(range(len(un.layers))):
        block, norm, down = un.layers[i]
        h1, g, mg, rg, h2, mn, rn, hn, vm, h3 = S.layers[i]
        d = h1.shape[1]
        pre = "layers.%d." % i
        wgrad(pre + "2.weight", dh, h3)
        bgrad(pre + "2.bias", dh)
        dh3 = ops.linear_dgrad(dh, _d(down.weight))
        att = un.attention_layers[i]
        ap = "attention_layers.%d." % i
        wgrad(ap + "out_proj.weight", dh3, vm)
        bgrad(ap + "out_proj.bias", dh3)
        dv = ops.linear_dgrad(dh3, _d(att.out_proj.weight))
        if S.head_masks is not None:
            dv = ops.scale_heads(dv, S.head_masks[i], att.num_heads)
        # q/k rows: exactly zero gradient (softmax over one key); preallocated buffers are zero there already
        gw = go(ap + "in_proj_weight")
        gb = go(ap + "in_proj_bias")
        if gw is None:
            gw = torch.zeros_like(att.in_proj_weight)
            gb = torch.zeros_like(att.in_proj_bias)
        ops.linear_wgrad(dv, hn, out=gw[2 * d:])
        pending.append((dv, gb[2 * d:], 1.0))
        G[ap + "in_proj_weight"], G[ap + "in_proj_bias"] = gw, gb
        dhn = ops.linear_dgrad(dv, _d(att.in_proj_weight)[2 * d:])
        dh2, G[pre + "1.weight"], G[pre + "1.bias"] = ops.layernorm_bwd(
            dhn, h2, _d(norm.weight), _d(norm.bias), None, mn, rn, R=dh3, dw_out=go(pre + "1.weight"),
            db_out=go(pre + "1.bias"))
        lin, ln = block[0], block[1]
        bm = S.block_masks[i] if S.block_masks is not None else None
        dg, G[pre + "0.1.weight"], G[pre + "0.1.bias"] = ops.layernorm_bwd(
            dh2, g, _d(ln.weight), _d(ln.bias), bm, mg, rg, act=SW, dw_out=go(pre + "0.1.weight"),
            db_out=go(pre + "0.1.bias"))
        wgrad(pre + "0.0.weight", dg, h1)
        bgrad(pre + "0.0.bias", dg)
        dh1 = ops.gemm(dg, _d(lin.weight), dg.shape[0], d, d, a_layout=ops.KC, b_layout=ops.KS, lda=d, ldb=d, R=dh2)
        tp = un.time_projections[i]
        tpp = "time_projections.%d." % i
        wgrad(tpp + "weight", dh1, S.tc)
        bgrad(tpp + "bias", dh1, alpha=tb)
        K = tp.weight.shape[1]
        dtc = ops.gemm(dh1, _d(tp.weight), dh1.shape[0], K, d, a_layout=ops.KC, b_layout=ops.KS, lda=d, ldb=K, R=dtc)
        dh = dh1
    wgrad("latent_proj.weight", dh, S.x)
    bgrad("latent_proj.bias", dh)
    dx = None
    if need_dx:
        dx = ops.gated_add(ops.linear_dgrad(dh, _d(un.latent_proj.weight)), dhf, rw)
    # time / class embedding MLPs
    dte = ops.axpby(1.0, dte, 1.0, dtc)
    tm = un.time_emb
    wgrad("time_emb.lin2.weight", dte, S.te1a)
    bgrad("time_emb.lin2.bias", dte)
    d1 = ops.act_bwd(ops.linear_dgrad(dte, _d(tm.lin2.weight)), S.te1, SW)
    wgrad("time_emb.lin1.weight", d1, S.feat)
    bgrad("time_emb.lin1.bias", d1)
    if S.ce is not None:
        dce = ops.axpby(1.0, dce, 1.0, dtc)
        cm = un.class_emb
        wgrad("class_emb.lin2.weight", dce, S.ce1a)
        bgrad("class_emb.lin2.bias", dce)
        d1 = ops.act_bwd(ops.linear_dgrad(dce, _d(cm.lin2.weight)), S.ce1, SW)
        wgrad("class_emb.lin1.weight", d1, S.emb)
        bgrad("class_emb.lin1.bias", d1)
        demb = ops.linear_dgrad(d1, _d(cm.lin1.weight))
        G["class_emb.embedding.weight"] = ops.embedding_bwd(demb, S.c, cm.embedding.weight.shape[0],
                                                            out=go("class_emb.embedding.weight"))
    ops.multi_colsum(pending)
    return G, dx
