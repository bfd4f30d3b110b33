"""The conv-UNet blocks that v2 defines but never instantiates (v2:434-498): UNetAttentionBlock
(the only genuine QK^T / softmax / PV of the file), UNetResidualBlock and SwitchSequential. Same constructors and
state-dict keys; forward and backward run on the HIP engine (NHWC inside): with grad enabled the blocks go through
torch.autograd.Function wrappers whose backward is the hand-written chain below, so they train with stock optimizers
like every other module of the surface (the blocks as a real spatial-latent denoiser)."""
import torch
import torch.nn as nn

from . import engine, ops
from .modules import LayerNorm2d, Swish, _f32c, _need_grad

SW = ops.ACT_SWISH


def _c2(t):
    return t.reshape(-1, t.shape[-1])


class _BlockFn(torch.autograd.Function):
    """forward(blk, names, n_in, *inputs, *params): blk._train_fwd(*inputs) -> (out NCHW, tape);
    backward: blk._train_bwd(tape, dout) -> ({param name: grad}, [input grads])."""

    @staticmethod
    def forward(ctx, blk, names, n_in, *args):
        out, tape = blk._train_fwd(*args[:n_in])
        ctx.blk, ctx.names, ctx.n_in, ctx.tape = blk, names, n_in, tape
        return out

    @staticmethod
    def backward(ctx, dout):
        G, dins = ctx.blk._train_bwd(ctx.tape, _f32c(dout), ctx.needs_input_grad[3:3 + ctx.n_in])
        ctx.tape = None
        return (None, None, None) + tuple(dins) + tuple(G.get(n) for n in ctx.names)


def _apply_block(blk, *inputs):
    names = [n for n, _ in blk.named_parameters()]
    return _BlockFn.apply(blk, names, len(inputs), *inputs, *[p for _, p in blk.named_parameters()])


def _flat_from(t, offset):
    """contiguous 1-D tail view of t starting at element `offset` (pointer arithmetic for strided GEMM operands)."""
    return t.view(-1)[offset:]


class UNetAttentionBlock(nn.Module):
    """v2:434-459."""

    def __init__(self, channels, num_heads=4):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads
        self.norm = nn.GroupNorm(1, channels)
        self.qkv = nn.Conv2d(channels, channels * 3, 1)
        self.proj = nn.Conv2d(channels, channels, 1)
        self._pk = engine.PackCache()

    def forward_nhwc(self, xh):
        B, H, W, C = xh.shape
        HW, nh = H * W, self.num_heads
        d = C // nh
        sc, sh, _, _ = ops.groupnorm_stats(xh, self.norm.weight.detach(), self.norm.bias.detach(), 1, self.norm.eps)
        xn = ops.affine_act(xh, sc, sh)
        # 1x1 conv = GEMM over pixels; channel j of qkv = which*C + head*d + i  (reshape(b,3,heads,d,hw), v2:447)
        qkv = ops.conv2d_fwd(xn, self._pk.conv(self.qkv), self.qkv.bias.detach(), 1, 0)  # (B,H,W,3C)
        scale = float(d) ** -0.5
        if ops.attention_ok(HW, C, nh):
            # one fused launch: QK^T, row softmax and PV for every (image, head); the merged tensor comes out in the d-major
            # channel order of v2:456-457
            merged = ops.attention_fwd(qkv, nh, scale)
        else:
            o = torch.empty(B, nh, HW, d, dtype=torch.float32, device=xh.device)
            for b in range(B):
                base = b * HW * 3 * C
                for h in range(nh):
                    q = _flat_from(qkv, base + h * d)
                    k = _flat_from(qkv, base + C + h * d)
                    v = _flat_from(qkv, base + 2 * C + h * d)
                    s = ops.gemm(q, k, HW, HW, d, lda=3 * C, ldb=3 * C)                    # S = Q K^T   (HW x HW)
                    p = ops.softmax_rows(s, scale)                                        # softmax(scale * S)
                    ops.gemm(p, v, HW, d, HW, a_layout=ops.KC, b_layout=ops.KS, lda=HW, ldb=3 * C, out=o[b, h])  # P V
            # out.permute(0,3,1,2).reshape(b,c,h,w) (v2:456-457): channel = i*heads + head  ((B,heads,HW,d)->(B,HW,d,heads))
            merged = ops.pack_conv_weight(o).view(B, H, W, C)
        return ops.conv2d_fwd(merged, self._pk.conv(self.proj), self.proj.bias.detach(), 1, 0, R=xh)

    # ---- training path -------------------------------------------------------------------------------------------
    def _train_fwd(self, x):
        xh = ops.nchw_to_nhwc(_f32c(x))
        B, H, W, C = xh.shape
        HW, nh = H * W, self.num_heads
        d = C // nh
        nw, nb = self.norm.weight.detach(), self.norm.bias.detach()
        gn = ops.groupnorm_stats(xh, nw, nb, 1, self.norm.eps)
        xn = ops.affine_act(xh, gn[0], gn[1])
        qkv = ops.conv2d_fwd(xn, self._pk.conv(self.qkv), self.qkv.bias.detach(), 1, 0)
        scale = float(d) ** -0.5
        if ops.attention_ok(HW, C, nh):
            # fused forward; the backward recomputes the probabilities from Q, K and the row statistics (max, sum of exp):
            # no (B, heads, HW, HW) tensor is kept (268 MB at B=256, C=128, 16x16)
            merged, P = ops.attention_fwd(qkv, nh, scale, want_stats=True)
        else:
            o = torch.empty(B, nh, HW, d, dtype=torch.float32, device=xh.device)
            P = torch.empty(B, nh, HW, HW, dtype=torch.float32, device=xh.device)
            for b in range(B):
                base = b * HW * 3 * C
                for h in range(nh):
                    q = _flat_from(qkv, base + h * d)
                    k = _flat_from(qkv, base + C + h * d)
                    v = _flat_from(qkv, base + 2 * C + h * d)
                    s_ = ops.gemm(q, k, HW, HW, d, lda=3 * C, ldb=3 * C)
                    P[b, h] = ops.softmax_rows(s_, scale)
                    ops.gemm(P[b, h], v, HW, d, HW, a_layout=ops.KC, b_layout=ops.KS, lda=HW, ldb=3 * C, out=o[b, h])
            merged = ops.pack_conv_weight(o).view(B, H, W, C)
        out = ops.conv2d_fwd(merged, self._pk.conv(self.proj), self.proj.bias.detach(), 1, 0, R=xh)
        return ops.nhwc_to_nchw(out), (xh, gn, xn, qkv, P, merged)

    def _train_bwd(self, tape, dout, need_in):
        xh, gn, xn, qkv, P, merged = tape
        B, H, W, C = xh.shape
        HW, nh = H * W, self.num_heads
        d = C // nh
        scale = float(d) ** -0.5
        G = {}
        dy = ops.nchw_to_nhwc(dout)                                      # also the residual branch's gradient
        G["proj.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy, merged, 1, 1, 1, 0), C, C, 1, 1)
        G["proj.bias"] = ops.colsum(_c2(dy))
        dmerged = ops.conv2d_dgrad(dy, self._pk.conv(self.proj), (H, W), 1, 0)
        if ops.attention_ok(HW, C, nh):
            # dQ / dK / dV of every (image, head) in two launches (fl_attention_bwd); `P` holds the forward's row statistics
            dqkv = ops.attention_bwd(qkv, merged, dmerged, P, nh, scale)
            return self._train_bwd_tail(G, dqkv, dy, xh, gn, xn, (H, W), C, need_in)
        do = ops.unpack_conv_weight(dmerged.view(B, HW, d, nh), B, nh, HW, d)   # inverse of the head merge: (B,nh,HW,d)
        dqkv = torch.empty_like(qkv)
        for b in range(B):
            base = b * HW * 3 * C
            for h in range(nh):
                q = _flat_from(qkv, base + h * d)
                k = _flat_from(qkv, base + C + h * d)
                v = _flat_from(qkv, base + 2 * C + h * d)
                p, g = P[b, h], do[b, h]
                # dV = P^T dO ; dP = dO V^T ; dS = scale * P o (dP - rowsum(dP o P)) ; dQ = dS K ; dK = dS^T Q
                ops.gemm(p, g, HW, d, HW, a_layout=ops.KS, b_layout=ops.KS, lda=HW, ldb=d,
                         out=_flat_from(dqkv, base + 2 * C + h * d), ldc=3 * C)
                dp = ops.gemm(g, v, HW, HW, d, lda=d, ldb=3 * C)
                ds = ops.softmax_rows_bwd(p, dp, scale)
                ops.gemm(ds, k, HW, d, HW, a_layout=ops.KC, b_layout=ops.KS, lda=HW, ldb=3 * C,
                         out=_flat_from(dqkv, base + h * d), ldc=3 * C)
                ops.gemm(ds, q, HW, d, HW, a_layout=ops.KS, b_layout=ops.KS, lda=HW, ldb=3 * C,
                         out=_flat_from(dqkv, base + C + h * d), ldc=3 * C)
        return self._train_bwd_tail(G, dqkv, dy, xh, gn, xn, (H, W), C, need_in)

    def _train_bwd_tail(self, G, dqkv, dy, xh, gn, xn, hw, C, need_in):
        H, W = hw
        G["qkv.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dqkv, xn, 1, 1, 1, 0), 3 * C, C, 1, 1)
        G["qkv.bias"] = ops.colsum(_c2(dqkv))
        dxn = ops.conv2d_dgrad(dqkv, self._pk.conv(self.qkv), (H, W), 1, 0)
        dx, G["norm.weight"], G["norm.bias"] = ops.groupnorm_bwd(dxn, xh, gn[0], gn[1], self.norm.weight.detach(), gn[2],
                                                                 gn[3], 1)
        dx = ops.axpby(1.0, dx, 1.0, dy)
        return G, [ops.nhwc_to_nchw(dx) if need_in[0] else None]

    def forward(self, x):
        if _need_grad(x, self):
            return _apply_block(self, x)
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(_f32c(x))))


class UNetResidualBlock(nn.Module):
    """v2:462-486."""

    def __init__(self, in_channels, out_channels, d_time=256, dropout_rate=0.2):
        super().__init__()
        self.norm1 = LayerNorm2d(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.time_emb = nn.Linear(d_time, out_channels)
        self.class_emb = nn.Linear(d_time, out_channels)
        self.act = Swish()
        self.dropout = nn.Dropout(dropout_rate)
        self.norm2 = LayerNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.residual = nn.Identity() if in_channels == out_channels else nn.Conv2d(in_channels, out_channels, 1)
        self._pk = engine.PackCache()
        from .rng import Philox
        self.rng = Philox(seed=46)

    def forward_nhwc(self, xh, t, c=None, mask=None):
        B, H, W, _ = xh.shape
        d = engine._d
        sc, sh, _, _ = ops.instnorm_stats(xh, d(self.norm1.weight), d(self.norm1.bias), self.norm1.eps)
        h = ops.conv2d_fwd(ops.affine_act(xh, sc, sh, act=SW), self._pk.conv(self.conv1), d(self.conv1.bias), 1, 1)
        Co = h.shape[-1]
        emb = ops.linear(_f32c(t), d(self.time_emb.weight), d(self.time_emb.bias), act=SW)
        if c is not None:
            emb = ops.axpby(1.0, emb, 1.0, ops.linear(_f32c(c), d(self.class_emb.weight), d(self.class_emb.bias), act=SW))
        ones = torch.ones(B, Co, dtype=torch.float32, device=xh.device)
        h = ops.affine_act(h, ones, emb, out=h)                       # h + emb[:, :, None, None]  (v2:479-482)
        sc, sh, _, _ = ops.instnorm_stats(h, d(self.norm2.weight), d(self.norm2.bias), self.norm2.eps)
        h = ops.affine_act(h, sc, sh, act=SW, out=h)
        if self.training and mask is None and self.dropout.p > 0:
            mask = self.rng.dropout(tuple(h.shape), xh.device, self.dropout.p)
        if mask is not None:
            h = ops.mul(h, mask, out=h)
        if isinstance(self.residual, nn.Identity):
            skip = xh
        else:
            skip = ops.conv2d_fwd(xh, self._pk.conv(self.residual), d(self.residual.bias), 1, 0)
        return ops.conv2d_fwd(h, self._pk.conv(self.conv2), d(self.conv2.bias), 1, 1, R=skip)

    # ---- training path -------------------------------------------------------------------------------------------
    def _train_fwd(self, x, t, c=None, mask=None):
        d = engine._d
        xh = ops.nchw_to_nhwc(_f32c(x))
        B, H, W, Cin = xh.shape
        n1 = ops.instnorm_stats(xh, d(self.norm1.weight), d(self.norm1.bias), self.norm1.eps)
        a0 = ops.affine_act(xh, n1[0], n1[1], act=SW)
        h = ops.conv2d_fwd(a0, self._pk.conv(self.conv1), d(self.conv1.bias), 1, 1)
        Co = h.shape[-1]
        t = _f32c(t)
        pt = ops.linear(t, d(self.time_emb.weight), d(self.time_emb.bias))
        emb = ops.act_fwd(pt, SW)
        pc = None
        if c is not None:
            c = _f32c(c)
            pc = ops.linear(c, d(self.class_emb.weight), d(self.class_emb.bias))
            emb = ops.axpby(1.0, emb, 1.0, ops.act_fwd(pc, SW))
        ones = torch.ones(B, Co, dtype=torch.float32, device=xh.device)
        h1 = ops.affine_act(h, ones, emb)                             # h + emb[:, :, None, None]
        n2 = ops.instnorm_stats(h1, d(self.norm2.weight), d(self.norm2.bias), self.norm2.eps)
        a2 = ops.affine_act(h1, n2[0], n2[1], act=SW)
        m = ops.nchw_to_nhwc(_f32c(mask)) if mask is not None else None
        if self.training and m is None and self.dropout.p > 0:
            m = self.rng.dropout(tuple(a2.shape), xh.device, self.dropout.p)
        a2m = ops.mul(a2, m) if m is not None else a2
        ident = isinstance(self.residual, nn.Identity)
        skip = xh if ident else ops.conv2d_fwd(xh, self._pk.conv(self.residual), d(self.residual.bias), 1, 0)
        out = ops.conv2d_fwd(a2m, self._pk.conv(self.conv2), d(self.conv2.bias), 1, 1, R=skip)
        return ops.nhwc_to_nchw(out), (xh, n1, a0, t, pt, c, pc, h1, n2, m, a2m, ident)

    def _train_bwd(self, tape, dout, need_in):
        d = engine._d
        xh, n1, a0, t, pt, c, pc, h1, n2, m, a2m, ident = tape
        B, H, W, Cin = xh.shape
        Co = h1.shape[-1]
        G = {}
        dy = ops.nchw_to_nhwc(dout)
        G["conv2.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy, a2m, 3, 3, 1, 1), Co, Co, 3, 3)
        G["conv2.bias"] = ops.colsum(_c2(dy))
        da2 = ops.conv2d_dgrad(dy, self._pk.conv(self.conv2), (H, W), 1, 1)
        if m is not None:
            da2 = ops.mul(da2, m)
        dh1, G["norm2.weight"], G["norm2.bias"] = ops.instnorm_bwd(da2, h1, n2[0], n2[1], d(self.norm2.weight), n2[2], n2[3],
                                                                   act=SW)
        demb = ops.chan_sum(dh1)                                          # (B, Co): gradient of the broadcast embedding
        dpt = ops.act_bwd(demb, pt, SW)
        G["time_emb.weight"], G["time_emb.bias"] = ops.linear_wgrad(dpt, t), ops.colsum(dpt)
        dt = ops.linear_dgrad(dpt, d(self.time_emb.weight)) if need_in[1] else None
        dc = None
        if pc is not None:
            dpc = ops.act_bwd(demb, pc, SW)
            G["class_emb.weight"], G["class_emb.bias"] = ops.linear_wgrad(dpc, c), ops.colsum(dpc)
            if len(need_in) > 2 and need_in[2]:
                dc = ops.linear_dgrad(dpc, d(self.class_emb.weight))
        G["conv1.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dh1, a0, 3, 3, 1, 1), Co, Cin, 3, 3)
        G["conv1.bias"] = ops.colsum(_c2(dh1))
        da0 = ops.conv2d_dgrad(dh1, self._pk.conv(self.conv1), (H, W), 1, 1)
        dx, G["norm1.weight"], G["norm1.bias"] = ops.instnorm_bwd(da0, xh, n1[0], n1[1], d(self.norm1.weight), n1[2], n1[3],
                                                                  act=SW)
        if ident:
            dx = ops.axpby(1.0, dx, 1.0, dy)
        else:
            G["residual.weight"] = ops.unpack_conv_weight(ops.conv2d_wgrad(dy, xh, 1, 1, 1, 0), Co, Cin, 1, 1)
            G["residual.bias"] = ops.colsum(_c2(dy))
            dx = ops.axpby(1.0, dx, 1.0, ops.conv2d_dgrad(dy, self._pk.conv(self.residual), (H, W), 1, 0))
        dins = [ops.nhwc_to_nchw(dx) if need_in[0] else None, dt]
        if len(need_in) > 2:
            dins.append(dc)
        if len(need_in) > 3:
            dins.append(None)   # dropout mask
        return G, dins

    def forward(self, x, t, c=None, mask=None):
        if _need_grad(x, t, c, self):
            ins = (x, t) if c is None and mask is None else ((x, t, c) if mask is None else (x, t, c, mask))
            if c is None and mask is not None:
                raise ValueError("UNetResidualBlock: an explicit dropout mask needs the class embedding argument too")
            return _apply_block(self, *ins)
        m = ops.nchw_to_nhwc(_f32c(mask)) if mask is not None else None
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(_f32c(x)), t, c, m))


class SwitchSequential(nn.Sequential):
    """v2:489-498."""

    def forward(self, x, t=None, c=None):
        for layer in self:
            if isinstance(layer, UNetResidualBlock):
                x = layer(x, t, c)
            elif isinstance(layer, UNetAttentionBlock):
                x = layer(x)
            else:
                x = layer(x)
        return x
