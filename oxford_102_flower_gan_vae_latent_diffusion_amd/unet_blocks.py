"""The conv-UNet blocks that v2 defines but never instantiates (v2:434-498): UNetAttentionBlock
(the only genuine QK^T / softmax / PV of the file), UNetResidualBlock and SwitchSequential. Same constructors and
state-dict keys; forward runs on the HIP engine (NHWC inside). Inference-only: `main()` cannot reach these classes,
so no backward chain is provided (autograd raises)."""
import torch
import torch.nn as nn

from . import engine, ops
from .modules import LayerNorm2d, Swish, _f32c, _need_grad

SW = ops.ACT_SWISH


def _no_grad_only(name, *things):
    if _need_grad(*things):
        raise NotImplementedError("%s: forward-only on this engine (the block is dead code in v2, see DESIGN.md)" % name)


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
        o = torch.empty(B, nh, HW, d, dtype=torch.float32, device=xh.device)
        scale = float(d) ** -0.5
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

    def forward(self, x):
        _no_grad_only("UNetAttentionBlock", x, self)
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

    def forward(self, x, t, c=None, mask=None):
        _no_grad_only("UNetResidualBlock", x, self)
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
