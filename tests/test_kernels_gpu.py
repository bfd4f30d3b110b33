"""Kernel-level parity of the C-ABI ops against plain torch fp32 CPU ops (floating point:
tolerances stated per test; fp32 MFMA is an exact-fp32 fma chain, so only summation order differs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oxford_102_flower_gan_vae_latent_diffusion_amd.ops as ops  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(1.0, ref), "max err %.3e (ref max %.3e)" % (err, ref)


def act_ref(x, act):
    return {0: lambda v: v, 1: lambda v: v * torch.sigmoid(v), 2: F.relu, 3: lambda v: F.leaky_relu(v, 0.2),
            4: torch.sigmoid}[act](x)


@pytest.mark.parametrize("M,N,K", [(100, 70, 36), (256, 256, 256), (256, 1024, 512), (8, 102, 256), (300, 33, 64),
                                   (256, 256, 4096), (2, 1024, 32768), (1, 1, 8192), (513, 129, 100)])
def test_gemm_nt(dev, M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias, R = rnd(N, seed=3), rnd(M, N, seed=4)
    for act in (0, 1):
        out = ops.gemm(A.to(dev), B.to(dev), M, N, K, bias=bias.to(dev), R=R.to(dev), act=act, r_scale=0.5)
        ref = act_ref(A.double() @ B.double().t() + bias.double(), act) + 0.5 * R.double()
        close(out, ref, tol=1e-5 * max(1, K ** 0.5 / 8))
    out = ops.gemm(A.to(dev), B.to(dev), M, N, K, bias=bias.to(dev), R=R.to(dev), act=1, r_pre=True, bias_scale=2.0)
    ref = act_ref(A.double() @ B.double().t() + 2 * bias.double() + R.double(), 1)
    close(out, ref, tol=1e-5 * max(1, K ** 0.5 / 8))


def test_softmax_rows_bwd_chan_sum_and_strided_gemm_output(dev):
    """small pieces of the attention / residual-block backward: softmax', per-(image, channel) pixel sums, and a GEMM
    that writes a column block of a wider matrix (ldc)."""
    x = rnd(37, 64, seed=1)
    dy = rnd(37, 64, seed=2)
    xg = x.double().requires_grad_(True)
    y = torch.softmax(0.3 * xg, dim=-1)
    (y * dy.double()).sum().backward()
    yd = ops.softmax_rows(x.to(dev), 0.3)
    close(yd, y, tol=2e-6)
    close(ops.softmax_rows_bwd(yd, dy.to(dev), 0.3), xg.grad, tol=2e-6)
    g = rnd(3, 16, 5, 7, seed=3)
    close(ops.chan_sum(nhwc(g).to(dev)), g.double().sum((2, 3)), tol=1e-5)
    A, B = rnd(48, 40, seed=4), rnd(24, 40, seed=5)
    wide = torch.full((48, 100), 7.0, device=dev)
    ops.gemm(A.to(dev), B.to(dev), 48, 24, 40, out=wide.view(-1)[30:], ldc=100)
    ref = torch.full((48, 100), 7.0, dtype=torch.float64)
    ref[:, 30:54] = A.double() @ B.double().t()
    close(wide, ref, tol=1e-5)


def test_multi_colsum(dev):
    """bias gradients of many layers in one launch: out_i = alpha_i * x_i.sum(0), incl. ragged widths and a view output."""
    shapes = [(256, 1024), (256, 256), (7, 102), (1, 64), (300, 33), (256, 1536)] + [(64, 96)] * 70   # > 64 entries: 2 launches
    xs = [rnd(M, N, seed=10 + i) for i, (M, N) in enumerate(shapes)]
    alphas = [1.0, 2.0, 0.5, 1.0, -1.0, 1.0] + [1.0] * 70
    big = torch.zeros(4096, device=dev)
    outs = [torch.empty(N, device=dev) for (_, N) in shapes]
    outs[1] = big[512:768]                                                      # a slice of a flat gradient buffer
    ops.multi_colsum([(x.to(dev), o, a) for x, o, a in zip(xs, outs, alphas)])
    for x, o, a in zip(xs, outs, alphas):
        close(o, a * x.double().sum(0), tol=2e-6 * max(1.0, x.shape[0] ** 0.5))
    assert float(big[:512].abs().max()) == 0.0 and float(big[768:].abs().max()) == 0.0
    ops.multi_colsum([])                                                        # nothing to do
    with pytest.raises(ValueError):
        ops.multi_colsum([(xs[0].to(dev), torch.empty(5, device=dev), 1.0)])


def test_gemm_splitk_policy(dev):
    """fl_set_gemm_splitk_tiles only changes HOW a small-grid GEMM is spread over the chip (split along K + a reduce
    launch, or not): the result is the same product up to fp32 summation order, the setting is restored by the caller."""
    M, N, K = 256, 1024, 1024   # 4 x 16 tiles of 64 x 64: below the default threshold of 160 -> split-K by default
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    ref = A.double() @ B.double().t() + bias.double()
    outs = {}
    old = ops.set_gemm_splitk_tiles(160)
    try:
        for tiles in (160, 16, 0):
            assert ops.set_gemm_splitk_tiles(tiles) in (160, 16, 0)
            outs[tiles] = ops.gemm(A.to(dev), B.to(dev), M, N, K, bias=bias.to(dev)).cpu()
            close(outs[tiles], ref, tol=4e-5)
    finally:
        ops.set_gemm_splitk_tiles(old)
    close(outs[0], outs[160], tol=4e-6)
    with pytest.raises(RuntimeError):
        ops.set_gemm_splitk_tiles(-1)
    assert ops.set_gemm_splitk_tiles(old) == old


@pytest.mark.parametrize("M,N,K", [(256, 512, 256), (64, 102, 256), (8, 256, 512), (256, 1024, 1024), (37, 66, 40)])
def test_linear_bwd(dev, M, N, K):
    x, w, dy = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(M, N, seed=3)
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev))
    close(dx, dy.double() @ w.double(), tol=2e-5)
    dw = ops.linear_wgrad(dy.to(dev), x.to(dev))
    close(dw, dy.double().t() @ x.double(), tol=2e-5)
    db = ops.colsum(dy.to(dev))
    close(db, dy.double().sum(0), tol=2e-5)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 32, 64, 3, 1, 1),
    (2, 8, 8, 64, 128, 3, 1, 1),
    (3, 16, 16, 4, 64, 3, 1, 1),
    (2, 16, 16, 64, 128, 4, 2, 1),
    (2, 8, 8, 128, 32, 4, 2, 1),
    (2, 16, 16, 32, 4, 3, 1, 1),
    (2, 4, 4, 64, 1, 4, 1, 0),
    (1, 10, 6, 8, 24, 3, 1, 1),
    (4, 16, 16, 4, 64, 4, 2, 1),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", CONV_CASES)
def test_conv_fwd_bwd(dev, B, H, W, Cin, Cout, k, s, p):
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, k, k, seed=2, scale=0.1), rnd(Cout, seed=3)
    xd, bd = nhwc(x).to(dev), b.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    close(wp, w.permute(0, 2, 3, 1), tol=0)  # pack = permute to [O][kh][kw][I]
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)
    y = ops.conv2d_fwd(xd, wp, bd, s, p)
    close(nchw(y), ref)
    y2 = ops.conv2d_fwd(xd, wp, bd, s, p, act=ops.ACT_SWISH)
    close(nchw(y2), act_ref(ref, 1))
    y3 = ops.conv2d_fwd(xd, wp, bd, s, p, out_nchw=True)
    close(y3, ref)
    Rr = rnd(*ref.shape, seed=9)
    y4 = ops.conv2d_fwd(xd, wp, bd, s, p, R=nhwc(Rr).to(dev))
    close(nchw(y4), ref + Rr.double())
    # backward
    dy = rnd(*ref.shape, seed=5)
    xg = x.double().requires_grad_(True)
    wg = w.double().requires_grad_(True)
    F.conv2d(xg, wg, None, stride=s, padding=p).backward(dy.double())
    dyd = nhwc(dy).to(dev)
    dwp = ops.conv2d_wgrad(dyd, xd, k, k, s, p)
    close(ops.unpack_conv_weight(dwp, Cout, Cin, k, k), wg.grad, tol=3e-5)
    if Cout % 4 == 0 and (s == 1 or (k == 4 and p == 1)) and k - 1 - p >= 0:
        dx = ops.conv2d_dgrad(dyd, wp, (H, W), s, p)
        close(nchw(dx), xg.grad, tol=3e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 8, 64, 32), (2, 4, 4, 128, 64), (3, 8, 8, 32, 64), (1, 16, 16, 128, 64)])
def test_convT_fwd(dev, B, H, W, Cin, Cout):
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cin, Cout, 4, 4, seed=2, scale=0.1), rnd(Cout, seed=3)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    wtp = ops.pack_conv_weight(w.to(dev))
    y = ops.convT2d_fwd(nhwc(x).to(dev), wtp, b.to(dev))
    close(nchw(y), ref)
    # ConvT backward = conv forward / conv wgrad with swapped roles
    dy = rnd(*ref.shape, seed=5)
    xg, wg = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv_transpose2d(xg, wg, None, stride=2, padding=1).backward(dy.double())
    dyd = nhwc(dy).to(dev)
    dx = ops.conv2d_fwd(dyd, wtp, None, 2, 1)
    close(nchw(dx), xg.grad, tol=3e-5)
    dwtp = ops.conv2d_wgrad(nhwc(x).to(dev), dyd, 4, 4, 2, 1)
    close(ops.unpack_conv_weight(dwtp, Cin, Cout, 4, 4), wg.grad, tol=3e-5)


def ln2d_ref(x, w, b, eps=1e-5):
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), keepdim=True, unbiased=False)
    return (x - mean) / torch.sqrt(var + eps) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 16, 16), (3, 128, 8, 8), (2, 4, 16, 16), (2, 512, 8, 8), (1, 96, 5, 7)])
@pytest.mark.parametrize("act", [0, 1])
def test_instnorm(dev, B, C, H, W, act):
    x = rnd(B, C, H, W, seed=1) * 2 + 0.5
    w, b, dy = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, H, W, seed=4)
    xg, wg, bg = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = act_ref(ln2d_ref(xg, wg, bg), act)
    ref.backward(dy.double())
    xd, wd, bd = nhwc(x).to(dev), w.to(dev), b.to(dev)
    scale, shift, mean, rstd = ops.instnorm_stats(xd, wd, bd)
    y = ops.affine_act(xd, scale, shift, act=act)
    close(nchw(y), ref, tol=2e-5)
    dx, dw, db = ops.instnorm_bwd(nhwc(dy).to(dev), xd, scale, shift, wd, mean, rstd, act=act)
    close(nchw(dx), xg.grad, tol=5e-5)
    close(dw, wg.grad, tol=5e-5)
    close(db, bg.grad, tol=5e-5)


@pytest.mark.parametrize("B,C,H,W,G", [(2, 64, 16, 16, 8), (2, 256, 8, 8, 32), (3, 32, 16, 16, 8)])
def test_groupnorm(dev, B, C, H, W, G):
    x = rnd(B, C, H, W, seed=1) * 2 + 0.5
    w, b, dy = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, H, W, seed=4)
    xg, wg, bg = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = act_ref(F.group_norm(xg, G, wg, bg, 1e-5), 1)
    ref.backward(dy.double())
    xd, wd, bd = nhwc(x).to(dev), w.to(dev), b.to(dev)
    scale, shift, mean, rstd = ops.groupnorm_stats(xd, wd, bd, G)
    y = ops.affine_act(xd, scale, shift, act=1)
    close(nchw(y), ref, tol=2e-5)
    dx, dw, db = ops.groupnorm_bwd(nhwc(dy).to(dev), xd, scale, shift, wd, mean, rstd, G, act=1)
    close(nchw(dx), xg.grad, tol=5e-5)
    close(dw, wg.grad, tol=5e-5)
    close(db, bg.grad, tol=5e-5)


@pytest.mark.parametrize("B,C,H,W", [(4, 128, 8, 8), (8, 64, 4, 4)])
def test_batchnorm(dev, B, C, H, W):
    x = rnd(B, C, H, W, seed=1) * 2 + 0.5
    w, b, dy = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, H, W, seed=4)
    rm, rv = 0.1 * rnd(C, seed=5), 1 + 0.1 * rnd(C, seed=6).abs()
    xg, wg, bg = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    ref = F.leaky_relu(F.batch_norm(xg, rm_ref, rv_ref, wg, bg, True, 0.1, 1e-5), 0.2)
    ref.backward(dy.double())
    xd, wd, bd = nhwc(x).to(dev), w.to(dev), b.to(dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    scale, shift, mean, rstd = ops.batchnorm_stats(xd, wd, bd, rmd, rvd, True)
    y = ops.affine_act(xd, scale, shift, act=3, act_param=0.2)
    close(nchw(y), ref, tol=2e-5)
    close(rmd, rm_ref, tol=2e-6)
    close(rvd, rv_ref, tol=2e-6)
    dx, dw, db = ops.batchnorm_bwd(nhwc(dy).to(dev), xd, scale, shift, wd, mean, rstd, act=3, act_param=0.2)
    close(nchw(dx), xg.grad, tol=5e-5)
    close(dw, wg.grad, tol=5e-5)
    close(db, bg.grad, tol=5e-5)
    # eval mode uses running stats
    scale, shift, _, _ = ops.batchnorm_stats(xd, wd, bd, rmd, rvd, False)
    y = ops.affine_act(xd, scale, shift)
    close(nchw(y), F.batch_norm(x.double(), rm_ref, rv_ref, w.double(), b.double(), False, 0.1, 1e-5), tol=2e-5)


@pytest.mark.parametrize("B,C,H,W,shards", [(4, 128, 16, 16, 2), (6, 512, 4, 4, 3), (8, 64, 32, 32, 4)])
def test_batchnorm_synchronised_shards(dev, B, C, H, W, shards):
    """SyncBatchNorm rule: each shard reduces its own sums, the (emulated) all-reduce adds them, and every shard then
    normalises / back-propagates with the statistics of the whole batch - vs F.batch_norm on the whole batch."""
    x = rnd(B, C, H, W, seed=1) * 2 + 0.5
    w, b, dy = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, H, W, seed=4)
    rm, rv = 0.1 * rnd(C, seed=5), 1 + 0.1 * rnd(C, seed=6).abs()
    xg, wg, bg = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    ref = F.leaky_relu(F.batch_norm(xg, rm_ref, rv_ref, wg, bg, True, 0.1, 1e-5), 0.2)
    ref.backward(dy.double())
    per = B // shards
    xs = [nhwc(x[i * per:(i + 1) * per]).to(dev) for i in range(shards)]
    dys = [nhwc(dy[i * per:(i + 1) * per]).to(dev) for i in range(shards)]
    wd, bd = w.to(dev), b.to(dev)

    class Exchange:  # two passes: collect every shard's sums, then hand each shard the total
        def __init__(self):
            self.parts, self.total = [], None

        def __call__(self, t):
            if self.total is None:
                self.parts.append(t.clone())
            else:
                t.copy_(self.total)

        def close(self):
            self.total = torch.stack(self.parts).sum(0)

    ex = Exchange()
    for xd in xs:
        ops.batchnorm_stats(xd, wd, bd, rm.to(dev), rv.to(dev), True, sync=ex, world=shards)
    ex.close()
    stats, dw_sum, db_sum = [], 0, 0
    for i, xd in enumerate(xs):
        rmd, rvd = rm.to(dev), rv.to(dev)
        st = ops.batchnorm_stats(xd, wd, bd, rmd, rvd, True, sync=ex, world=shards)
        stats.append(st)
        y = ops.affine_act(xd, st[0], st[1], act=3, act_param=0.2)
        close(nchw(y), ref[i * per:(i + 1) * per], tol=2e-5)
        close(rmd, rm_ref, tol=2e-6)
        close(rvd, rv_ref, tol=2e-6)
    ex = Exchange()
    for xd, dyd, st in zip(xs, dys, stats):
        ops.batchnorm_bwd(dyd, xd, st[0], st[1], wd, st[2], st[3], act=3, act_param=0.2, sync=ex, world=shards)
    ex.close()
    for i, (xd, dyd, st) in enumerate(zip(xs, dys, stats)):
        dx, dw, db = ops.batchnorm_bwd(dyd, xd, st[0], st[1], wd, st[2], st[3], act=3, act_param=0.2, sync=ex, world=shards)
        close(nchw(dx), xg.grad[i * per:(i + 1) * per], tol=5e-5)
        dw_sum, db_sum = dw_sum + dw.double(), db_sum + db.double()
    close(dw_sum, wg.grad, tol=5e-5)   # local shares add up to the global-batch dw / db
    close(db_sum, bg.grad, tol=5e-5)


@pytest.mark.parametrize("rows,D", [(8, 256), (256, 1024), (4, 32768), (3, 512), (1000, 96), (2048, 128)])  # last: > 1024 rows = two-launch dw/db path
def test_layernorm(dev, rows, D):
    x = rnd(rows, D, seed=1) * 2 + 0.3
    w, b, dy = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3), rnd(rows, D, seed=4)
    mask = (torch.rand(rows, D, generator=torch.Generator().manual_seed(7)) > 0.3).float() / 0.7
    for use_mask in (False, True):
        xg, wg, bg = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
        u = F.layer_norm(xg, (D,), wg, bg, 1e-5)
        if use_mask:
            u = u * mask.double()
        ref = act_ref(u, 1)
        ref.backward(dy.double())
        md = mask.to(dev) if use_mask else None
        y, mean, rstd = ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), mask=md, act=1)
        close(y, ref, tol=2e-5)
        dx, dw, db = ops.layernorm_bwd(dy.to(dev), x.to(dev), w.to(dev), b.to(dev), md, mean, rstd, act=1)
        close(dx, xg.grad, tol=5e-5)
        close(dw, wg.grad, tol=5e-5)
        close(db, bg.grad, tol=5e-5)


@pytest.mark.parametrize("B,C,H,W", [(2, 128, 8, 8), (2, 512, 4, 4), (1, 32, 16, 16), (2, 256, 8, 8)])
def test_res_tail(dev, B, C, H, W):
    y, xres = rnd(B, C, H, W, seed=1), rnd(B, C, H, W, seed=2)
    scale, shift, s = 1 + 0.1 * rnd(B, C, seed=3), 0.1 * rnd(B, C, seed=4), torch.sigmoid(rnd(B, C, seed=5))
    saw = rnd(1, 2, 7, 7, seed=6, scale=0.2)
    u = (y.double() * scale.double().view(B, C, 1, 1) + shift.double().view(B, C, 1, 1)) * s.double().view(B, C, 1, 1)
    m = torch.cat([u.mean(1, keepdim=True), u.max(1, keepdim=True)[0]], 1)
    att = torch.sigmoid(F.conv2d(m, saw.double(), padding=3))
    ref = act_ref(u * att + xres.double(), 1)
    yd = nhwc(y).to(dev)
    mp = ops.res_tail_map(yd, scale.to(dev), shift.to(dev), s.to(dev))
    close(mp, m, tol=1e-5)
    out, att_d = ops.res_tail_out(yd, scale.to(dev), shift.to(dev), s.to(dev), mp, saw.to(dev).contiguous(),
                                  nhwc(xres).to(dev), final_swish=True, save_att=True)
    close(nchw(out), ref, tol=2e-5)
    close(att_d, att[:, 0], tol=1e-5)
    # identity forms (stand-alone CALayer / SpatialAttention)
    out2 = ops.res_tail_out(yd, None, None, s.to(dev))
    close(nchw(out2), y.double() * s.double().view(B, C, 1, 1), tol=1e-6)


@pytest.mark.parametrize("B,C,H", [(3, 128, 8), (2, 256, 16), (2, 512, 8), (2, 128, 6), (2, 32, 8)])
def test_res_tail_with_fused_residual_preop(dev, B, C, H):
    """The inference form: the residual enters as swish(x * a[b,c] + d[b,c]) (its LayerNorm2d-apply fused in). Whole 8-pixel runs and
    C >= 64 take res_tail_stream_kernel (coefficients loaded once per run), the rest (6x6 maps, C = 32) the general kernel: both
    against fp64 (fast Swish: 3e-5)."""
    y, xres = rnd(B, C, H, H, seed=11), rnd(B, C, H, H, seed=12)
    scale, shift, s = 1 + 0.1 * rnd(B, C, seed=13), 0.1 * rnd(B, C, seed=14), torch.sigmoid(rnd(C, seed=15))
    a, d = rnd(B, C, seed=16).abs() + 0.5, rnd(B, C, seed=17, scale=0.3)
    saw = rnd(1, 2, 7, 7, seed=18, scale=0.2)
    v = lambda t: t.double().view(B, C, 1, 1)
    u = (y.double() * v(scale) + v(shift)) * s.double().view(1, C, 1, 1)
    m = torch.cat([u.mean(1, keepdim=True), u.max(1, keepdim=True)[0]], 1)
    att = torch.sigmoid(F.conv2d(m, saw.double(), padding=3))
    r = xres.double() * v(a) + v(d)
    r = r * torch.sigmoid(r)
    for fs in (True, False):
        ref = u * att + r
        ref = ref * torch.sigmoid(ref) if fs else ref
        yd = nhwc(y).to(dev)
        mp = ops.res_tail_map(yd, scale.to(dev), shift.to(dev), s.to(dev))
        out = ops.res_tail_out(yd, scale.to(dev), shift.to(dev), s.to(dev), mp, saw.to(dev).contiguous(), nhwc(xres).to(dev),
                               final_swish=fs, xres_pre=(a.to(dev), d.to(dev)))
        close(nchw(out), ref, tol=3e-5)


def test_elementwise_and_layout(dev):
    x = rnd(3, 3, 8, 8, seed=1)
    y = ops.nchw_to_nhwc(x.to(dev), 4)
    assert y.shape == (3, 8, 8, 4)
    close(y[..., :3], x.permute(0, 2, 3, 1), tol=0)
    assert float(y[..., 3].abs().max()) == 0.0
    close(ops.nhwc_to_nchw(y, 3), x, tol=0)
    v = rnd(1000, seed=2) * 3
    for act in (1, 2, 3, 4):
        vg = v.double().requires_grad_(True)
        r = act_ref(vg, act)
        r.sum().backward()
        close(ops.act_fwd(v.to(dev), act, 0.2), r, tol=1e-6)
        close(ops.act_bwd(torch.ones_like(v).to(dev), v.to(dev), act, 0.2), vg.grad, tol=1e-6)
    a, b = rnd(100, seed=3), rnd(100, seed=4)
    close(ops.axpby(2.0, a.to(dev), -0.5, b.to(dev)), 2 * a - 0.5 * b, tol=1e-6)


def test_embeddings_and_diffusion_ops(dev):
    import math
    t = torch.tensor([0, 1, 17, 500, 999], dtype=torch.int64)
    half = 128
    emb = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -emb)
    e = t[:, None] * f[None, :]
    ref = torch.cat((e.sin(), e.cos()), dim=1)
    close(ops.time_sinusoid(t.to(dev), ops.time_freqs(256, dev)), ref, tol=2e-6)
    table, idx = rnd(102, 256, seed=1), torch.tensor([5, 5, 101, 0, 7, 5], dtype=torch.int64)
    close(ops.embedding_fwd(table.to(dev), idx.to(dev)), table[idx], tol=0)
    dout = rnd(6, 256, seed=2)
    dt = torch.zeros(102, 256).index_add_(0, idx, dout)
    close(ops.embedding_bwd(dout.to(dev), idx.to(dev), 102), dt, tol=1e-6)
    beta = torch.linspace(0.0001, 0.02, 1000)
    alpha = 1 - beta
    ab = torch.cumprod(alpha, 0)
    x0, eps = rnd(5, 256, seed=3), rnd(5, 256, seed=4)
    ref = torch.sqrt(ab[t].reshape(-1, 1)) * x0 + torch.sqrt(1 - ab[t].reshape(-1, 1)) * eps
    close(ops.q_sample(x0.to(dev), eps.to(dev), t.to(dev), ab.to(dev)), ref, tol=1e-6)
    for tt in (999, 1, 0):
        tp = torch.tensor([tt], dtype=torch.int32, device=dev)
        et, nz = rnd(5, 256, seed=5), rnd(5, 256, seed=6)
        mean = (x0 - ((1 - alpha[tt]) / torch.sqrt(1 - ab[tt])) * et) / torch.sqrt(alpha[tt])
        ref = mean + torch.sqrt(beta[tt]) * nz if tt > 0 else mean
        out = ops.p_sample_update(x0.to(dev), et.to(dev), nz.to(dev), tp, alpha.to(dev), ab.to(dev), beta.to(dev))
        close(out, ref, tol=1e-6)
    mu, lv = rnd(4, 256, seed=7), rnd(4, 256, seed=8) * 4
    z, lvc = ops.reparameterize(mu.to(dev), lv.to(dev), eps[:4].contiguous().to(dev), want_clamped=True)
    lvr = torch.clamp(lv, -2.0, 10.0)
    close(lvc, lvr, tol=0)
    close(z, mu + eps[:4] * torch.exp(0.5 * lvr), tol=1e-6)


def test_philox(dev):
    out = torch.empty(1 << 20, device=dev)
    ops.philox_fill(out, seed=1234, mode=0)
    assert abs(float(out.mean())) < 5e-3 and abs(float(out.std()) - 1.0) < 5e-3
    out2 = torch.empty(1 << 20, device=dev)
    ops.philox_fill(out2, seed=1234, mode=0)
    assert torch.equal(out, out2)
    ops.philox_fill(out2, seed=1234, offset=1 << 18, mode=0)
    assert not torch.equal(out, out2)
    m = torch.empty(1 << 20, device=dev)
    ops.philox_fill(m, seed=7, mode=1, p=0.3)
    keep = float((m > 0).float().mean())
    assert abs(keep - 0.7) < 5e-3
    assert abs(float(m.max()) - 1 / 0.7) < 1e-6
    r = torch.empty(1 << 16, dtype=torch.int64, device=dev)
    ops.philox_randint(r, seed=3, hi=1000)
    assert int(r.min()) >= 0 and int(r.max()) <= 999 and abs(float(r.float().mean()) - 499.5) < 10


def test_losses_and_optimizer(dev):
    x, y = rnd(6, 3, 8, 8, seed=1), rnd(6, 3, 8, 8, seed=2)
    xg = x.double().requires_grad_(True)
    ref = torch.sqrt(((xg - y.double()) ** 2).view(6, -1).sum(1) + 1e-8).mean()
    ref.backward()
    loss, norms = ops.euclid_loss_fwd(x.to(dev), y.to(dev))
    close(loss, ref.reshape(1), tol=1e-6)
    close(ops.euclid_loss_bwd(x.to(dev), y.to(dev), norms), xg.grad, tol=1e-6)
    # KL
    mu, lv = rnd(5, 256, seed=3) * 4, rnd(5, 256, seed=4) * 3
    mug, lvg = mu.double().requires_grad_(True), lv.double().requires_grad_(True)
    m_, l_ = torch.clamp(mug, -10, 10), torch.clamp(lvg, -2, 10)
    kl = torch.clamp(-0.5 * torch.sum(1 + l_ - m_.pow(2) - l_.exp(), dim=1), 0, 100).mean() + 1e-4 * torch.sum(m_.pow(2))
    kl.backward()
    loss, kl_row = ops.kl_fwd(mu.to(dev), lv.to(dev))
    close(loss, kl.reshape(1), tol=2e-6)
    dmu, dlv = ops.kl_bwd(mu.to(dev), lv.to(dev), kl_row)
    close(dmu, mug.grad, tol=2e-6)
    close(dlv, lvg.grad, tol=2e-6)
    # BCE
    p = torch.sigmoid(rnd(16, seed=5))
    for tgt in (0.0, 1.0):
        pg = p.double().requires_grad_(True)
        r = F.binary_cross_entropy(pg, torch.full_like(pg, tgt))
        r.backward()
        close(ops.bce_fwd(p.to(dev), tgt), r.reshape(1), tol=1e-6)
        close(ops.bce_bwd(p.to(dev), tgt), pg.grad, tol=1e-6)
    # CE
    lg, lab = rnd(7, 102, seed=6) * 3, torch.randint(0, 102, (7,), generator=torch.Generator().manual_seed(1))
    lgg = lg.double().requires_grad_(True)
    r = F.cross_entropy(lgg, lab)
    r.backward()
    loss, probs = ops.cross_entropy_fwd(lg.to(dev), lab.to(dev))
    close(loss, r.reshape(1), tol=1e-6)
    close(ops.cross_entropy_bwd(probs, lab.to(dev)), lgg.grad, tol=1e-6)
    # clip + AdamW vs torch.optim
    n = 10007
    p0, g0 = rnd(n, seed=7), rnd(n, seed=8) * 3
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    pd, md, vd = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lr_dev, step_dev = torch.tensor([1e-3], device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    sq, coef, nrm = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    for it in range(3):
        g = g0 * (it + 1)
        pr.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = g.to(dev)
        ops.sqnorm(gd, sq)
        ops.clip_coef(sq, 1.0, coef, nrm)
        ops.step_counter_add(step_dev, None, 1)
        ops.adam_step(pd, gd, md, vd, lr_dev, step_dev, wd=1e-5, coef_dev=coef)
        close(nrm, tn.reshape(1), tol=1e-5)
        close(pd, pr.data, tol=2e-6)


@pytest.mark.parametrize("B,HW,act", [(1, 64, 0), (3, 64, 2), (2, 32, 1)])
def test_conv_stem_dedicated_kernel(dev, B, HW, act):
    """nn.Conv2d(3, 64, 3, padding=1) (v2:186; VGG16 features[0]): routed to the dedicated K=36 kernel (conv_first.hip),
    with and without the fused LayerNorm2d statistics; reference = F.conv2d in fp64 on CPU."""
    x = rnd(B, 3, HW, HW, seed=1)
    w = rnd(64, 3, 3, 3, seed=2, scale=0.3)
    bias = rnd(64, seed=3)
    lw, lb = rnd(64, seed=4), rnd(64, seed=5)
    xh = ops.nchw_to_nhwc(x.to(dev), 4)
    wp = ops.pack_conv_weight(w.to(dev), Ipad=4)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    y = ops.conv2d_fwd(xh, wp, bias.to(dev), 1, 1, act=act, act_param=0.2)
    close(nchw(y), act_ref(ref, act), tol=1e-5)
    y2, (sc, sh, mean, rstd) = ops.conv2d_fwd_instnorm(xh, wp, bias.to(dev), 1, 1, lw.to(dev), lb.to(dev), 1e-5)
    close(nchw(y2), ref, tol=1e-5)
    m = ref.mean((2, 3))
    v = ref.var((2, 3), unbiased=False)
    close(mean, m, tol=1e-5)
    close(rstd, 1 / torch.sqrt(v + 1e-5), tol=2e-5)
    close(sc, lw.double() / torch.sqrt(v + 1e-5), tol=2e-5)
    close(sh, lb.double() - m * lw.double() / torch.sqrt(v + 1e-5), tol=3e-5)


def test_multi_axpby(dev):
    """batched per-parameter gradient gather: > 96 entries (two launches), ragged sizes, both assign and accumulate."""
    g = torch.Generator().manual_seed(0)
    sizes = [int(torch.randint(1, 5000, (1,), generator=g)) for _ in range(130)] + [1, 1023, 1024, 1025, 70000]
    src = [torch.randn(n, generator=g) for n in sizes]
    dst0 = [torch.randn(n, generator=g) for n in sizes]
    ds, dd = [s.to(dev) for s in src], [d.to(dev) for d in dst0]
    ops.multi_axpby(list(zip(ds, dd)))
    for s, d in zip(src, dd):
        assert torch.equal(d.cpu(), s)
    dd = [d.to(dev) for d in dst0]
    ops.multi_axpby(list(zip(ds, dd)), 0.5, 2.0)
    for s, d0, d in zip(src, dst0, dd):
        close(d, 0.5 * s.double() + 2.0 * d0.double(), tol=1e-6)


@pytest.fixture
def each_gemm_math():
    """run the body once per matrix-core mode; always restore the default"""
    old = ops.get_gemm_math()
    yield ("fp32", "split")
    ops.set_gemm_math(old)


@pytest.mark.parametrize("M,N,K", [(256, 256, 1024), (512, 128, 4608), (128, 512, 36864)])
def test_split_mfma_is_fp32_equivalent(dev, each_gemm_math, M, N, K):
    """The default GEMM math splits each fp32 operand exactly into three bf16 pieces and sums 6 of the 9 piece products
    on the bf16 matrix pipe. Claim: its error against the exact (fp64) product is of the size of the native fp32-MFMA
    kernel's own accumulation error - measured here on the same data, several K, and on badly scaled operands."""
    for scale_a, scale_b in ((1.0, 1.0), (1e-3, 1e4)):
        A = rnd(M, K, seed=1) * scale_a
        B = rnd(N, K, seed=2) * scale_b
        # rows with a wide dynamic range inside one dot product
        A[:, ::7] *= 64.0
        B[:, ::5] *= 1.0 / 64.0
        ref = A.double() @ B.double().t()
        err = {}
        for mode in each_gemm_math:
            ops.set_gemm_math(mode)
            out = ops.gemm(A.to(dev), B.to(dev), M, N, K)
            err[mode] = ((out.cpu().double() - ref).abs().max() / ref.abs().max()).item()
        assert err["split"] <= max(2.0 * err["fp32"], 3e-7), "split %.3e vs fp32 MFMA %.3e (K=%d)" % (err["split"], err["fp32"], K)
        assert err["fp32"] < 1e-5 and err["split"] < 1e-5


def test_conv_fwd_both_gemm_math_modes(dev, each_gemm_math):
    x = rnd(4, 128, 32, 32, seed=1)
    w = rnd(128, 128, 3, 3, seed=2, scale=0.05)
    b = rnd(128, seed=3)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xh, wp = nhwc(x).to(dev), ops.pack_conv_weight(w.to(dev))
    outs = {}
    for mode in each_gemm_math:
        ops.set_gemm_math(mode)
        outs[mode] = nchw(ops.conv2d_fwd(xh, wp, b.to(dev), 1, 1))
        close(outs[mode], ref, tol=1e-5)
    close(outs["split"], outs["fp32"].cpu(), tol=2e-6)


@pytest.fixture(params=[0, 1])
def halo_mode(request):
    old = ops.set_conv_halo_mode(request.param)
    yield request.param
    ops.set_conv_halo_mode(old)


@pytest.mark.parametrize("B,Cin,Cout,HW,k,s,p", [(2, 128, 128, 32, 3, 1, 1), (3, 64, 128, 64, 4, 2, 1), (1, 256, 256, 16, 3, 1, 1),
                                                 (2, 256, 512, 16, 4, 2, 1),
                                                 # 3x3 s1 p1 on 8/16/32-wide maps -> halo-tile kernel (conv_halo.hip):
                                                 # two whole 8x8 images per tile, tiles below the top image row, 4 N-tiles
                                                 (4, 512, 512, 8, 3, 1, 1), (3, 128, 256, 32, 3, 1, 1), (2, 256, 128, 16, 3, 1, 1),
                                                 # 4x4 s2 p1 halo kernel: 16-wide outputs, and four 8x8 output images (two per tile)
                                                 (2, 128, 256, 32, 4, 2, 1), (4, 256, 512, 16, 4, 2, 1),
                                                 # grids that are not a multiple of the 8 XCDs (10 and 6 blocks), three N-tiles
                                                 (5, 128, 128, 16, 3, 1, 1), (1, 256, 384, 16, 3, 1, 1), (6, 64, 128, 16, 4, 2, 1)])
def test_conv_fwd_presplit_weight(dev, B, Cin, Cout, HW, k, s, p, halo_mode):
    """pre-split weight fragments (B_FRAGS) give the same convolution as the in-kernel split and as fp64 F.conv2d,
    with and without the fused LayerNorm2d statistics; both variants of the 3x3 halo kernel (fl_set_conv_halo_mode)."""
    x = rnd(B, Cin, HW, HW, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=0.05)
    bias = rnd(Cout, seed=3)
    lw, lb = rnd(Cout, seed=4), rnd(Cout, seed=5)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=p)
    xh = nhwc(x).to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    ws = ops.split_pack_weight(wp)
    geom = tuple(wp.shape)
    y = ops.conv2d_fwd_presplit(xh, ws, geom, bias.to(dev), s, p, act=ops.ACT_RELU)
    close(nchw(y), torch.relu(ref), tol=1e-5)
    y2, (sc, sh, mean, rstd) = ops.conv2d_fwd_presplit(xh, ws, geom, bias.to(dev), s, p, ln_w=lw.to(dev), ln_b=lb.to(dev))
    close(nchw(y2), ref, tol=1e-5)
    y3, (sc3, sh3, _, _) = ops.conv2d_fwd_instnorm(xh, wp, bias.to(dev), s, p, lw.to(dev), lb.to(dev), 1e-5)
    close(y2, y3.cpu(), tol=5e-6)  # same math, different K order (halo kernel: channel chunk outer, tap inner)
    close(sc, sc3.cpu(), tol=1e-5)
    close(sh, sh3.cpu(), tol=1e-5)
    close(mean, ref.mean((2, 3)), tol=1e-5)


# ----------------------------------------------------------------------------------------------- f16x2 conv kernels (conv_h2.hip)
H2_CASES = [  # B, Cin, Cout, H, k, s, p: every width of both kernels; several N tiles; tiles below an image's first row; 4 images per tile
    (2, 128, 128, 32, 3, 1, 1), (1, 256, 256, 16, 3, 1, 1), (4, 512, 512, 8, 3, 1, 1), (1, 128, 256, 32, 3, 1, 1), (8, 64, 128, 8, 3, 1, 1),
    (1, 64, 128, 64, 4, 2, 1), (1, 128, 256, 32, 4, 2, 1), (4, 256, 512, 16, 4, 2, 1), (2, 16, 128, 32, 4, 2, 1), (3, 32, 128, 32, 3, 1, 1),
]


def _h2_setup(B, Cin, Cout, H, k, s, p, seed=0):
    x = rnd(B, Cin, H, H, seed=seed + 1)
    x = x * torch.sigmoid(x)
    w = rnd(Cout, Cin, k, k, seed=seed + 2, scale=(2.0 / (k * k * Cin)) ** 0.5)
    b = rnd(Cout, seed=seed + 3, scale=0.1)
    return x, w, b


@pytest.mark.parametrize("B,Cin,Cout,H,k,s,p", H2_CASES)
def test_f16x2_conv_is_fp32_equivalent(dev, B, Cin, Cout, H, k, s, p):
    """conv_h2.hip: operands scaled by powers of two and split (round to nearest) into two fp16 pieces, 3 of 4 piece products
    on v_mfma_f32_32x32x16_f16. Each operand carries a 2^-22 relative representation error (unbiased), so against the exact
    (fp64) convolution the kernel sits at 2-4x the native fp32 MFMA kernel's own accumulation error: <= 3e-6 of the output's
    max for K up to 4608, the same class as the bf16x3 math (measured side by side in tools/conv_h2_bench.py) and 30x inside
    the 1e-4 parity tolerance. Checked with a static activation exponent, with an amax bound, with tiny activations (low pieces in fp16's
    subnormal range) and with the fused LayerNorm2d statistics / bias / activation epilogue."""
    x, w, b = _h2_setup(B, Cin, Cout, H, k, s, p)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)
    xh, wp, bd = nhwc(x).to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev)
    assert ops.f16x2_ok(xh, Cout, k, k, s, p)
    geom = tuple(wp.shape)
    wh2 = ops.pack_weight_f16x2(wp)
    old = ops.get_gemm_math()
    ops.set_gemm_math("fp32")
    y32 = nchw(ops.conv2d_fwd(xh, wp, bd, s, p)).cpu().double()
    ops.set_gemm_math(old)
    e32 = ((y32 - ref).abs().max() / ref.abs().max()).item()
    amax = torch.full((1,), float(x.abs().max()) * 1.5, device=dev)
    for kw in (dict(), dict(x_amax=amax)):
        y = nchw(ops.conv2d_fwd_f16x2(xh, wh2, geom, bd, s, p, **kw)).cpu().double()
        e = ((y - ref).abs().max() / ref.abs().max()).item()
        assert e <= 3e-6 and e <= max(6.0 * e32, 1e-6), "f16x2 %.3e vs fp32 MFMA %.3e %s" % (e, e32, kw.keys())
    # tiny activations under the static exponent
    ys = nchw(ops.conv2d_fwd_f16x2(xh * 1e-3, wh2, geom, None, s, p)).cpu().double()
    refs = (ref - b.double().view(1, -1, 1, 1)) * 1e-3
    assert ((ys - refs).abs().max() / refs.abs().max()).item() <= 4e-6
    # activation epilogue + fused LayerNorm2d statistics
    yr = nchw(ops.conv2d_fwd_f16x2(xh, wh2, geom, bd, s, p, act=ops.ACT_RELU, x_amax=amax)).cpu().double()
    close(yr, ref.clamp_min(0), tol=4e-6)
    Ho = ref.shape[-1]
    if (Ho * Ho) % 64 == 0:
        lnw, lnb = rnd(Cout, seed=9).abs() + 0.5, rnd(Cout, seed=10)
        _, (sc, sh, mean, rstd) = ops.conv2d_fwd_f16x2(xh, wh2, geom, bd, s, p, ln_w=lnw.to(dev), ln_b=lnb.to(dev), x_amax=amax)
        m = ref.mean(dim=(2, 3))
        r = 1.0 / torch.sqrt(ref.var(dim=(2, 3), unbiased=False) + 1e-5)
        close(mean, m, tol=2e-5)
        close(rstd, r, tol=2e-5)
        close(sc, r * lnw.double(), tol=2e-5)
        close(sh, lnb.double() - m * r * lnw.double(), tol=5e-5)


@pytest.mark.parametrize("B,Cin,Cout,H,k,s,p", [H2_CASES[0], H2_CASES[2], H2_CASES[5], H2_CASES[7]])
def test_f16x2_conv_fused_preop(dev, B, Cin, Cout, H, k, s, p):
    """pre-op: the conv consumes swish(x * scale[b,c] + shift[b,c]) (the producer's LayerNorm2d-apply + Swish) inside its halo
    staging; zero padding applies to the TRANSFORMED tensor. Against fl_affine_act + the same conv, and against fp64."""
    x, w, b = _h2_setup(B, Cin, Cout, H, k, s, p, seed=20)
    sc, sh = rnd(B, Cin, seed=31).abs() + 0.5, rnd(B, Cin, seed=32, scale=0.5)
    a = x.double() * sc.double().view(B, Cin, 1, 1) + sh.double().view(B, Cin, 1, 1)
    a = a * torch.sigmoid(a)
    ref = F.conv2d(a, w.double(), b.double(), stride=s, padding=p)
    xh, wp, bd = nhwc(x).to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev)
    wh2, geom = ops.pack_weight_f16x2(wp), tuple(wp.shape)
    bound = torch.full((1,), float(a.abs().max()) * 1.5, device=dev)
    y = nchw(ops.conv2d_fwd_f16x2(xh, wh2, geom, bd, s, p, x_amax=bound, pre=(sc.to(dev), sh.to(dev))))
    close(y, ref, tol=3e-6)
    xa = ops.affine_act(xh, sc.to(dev), sh.to(dev), act=ops.ACT_SWISH)
    y2 = nchw(ops.conv2d_fwd_f16x2(xa, wh2, geom, bd, s, p, x_amax=bound))
    close(y, y2.cpu(), tol=3e-6)


@pytest.mark.parametrize("B,Cin,Cout,H,k,s,p,pre", [(16, 512, 512, 8, 3, 1, 1, True), (16, 256, 256, 16, 3, 1, 1, True), (8, 128, 128, 32, 3, 1, 1, False),
                                                    (8, 32, 128, 32, 3, 1, 1, True), (8, 64, 128, 64, 4, 2, 1, True), (16, 256, 512, 16, 4, 2, 1, False)])
def test_f16x2_convs_are_bit_stable_next_to_a_streaming_queue(dev, B, Cin, Cout, H, k, s, p, pre):
    """The conv kernels' halo loads are asynchronous inline-asm loads waited for by hand-counted s_waitcnt: their results must not depend
    on WHEN a load lands. Forty launches of each kernel family on one stream while a second stream saturates the memory system (268 MB
    streaming passes) must reproduce, bit for bit, the output of a launch on an idle device. (Round 3's fault - DESIGN 7c - was a halo
    load landing late into a register the compiler had reused: it needed exactly this, a loaded memory system, to show.)"""
    x, w, b = _h2_setup(B, Cin, Cout, H, k, s, p, seed=60)
    xh, wp, bd = nhwc(x).to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev)
    wh2, geom = ops.pack_weight_f16x2(wp), tuple(wp.shape)
    amax = torch.full((1,), float(x.abs().max()) * 4.0, device=dev)
    kw = dict(x_amax=amax)
    if pre:
        kw["pre"] = (rnd(B, Cin, seed=61).abs().to(dev) + 0.5, rnd(B, Cin, seed=62, scale=0.5).to(dev))
    ref = ops.conv2d_fwd_f16x2(xh, wh2, geom, bd, s, p, **kw).clone()
    torch.cuda.synchronize()
    big, out = torch.empty(256, 32, 32, 256, device=dev), torch.empty(256, 32, 32, 256, device=dev)
    ops.philox_fill(big, seed=3, mode=0)
    sc, sh = torch.ones(256, 256, device=dev), torch.zeros(256, 256, device=dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    ys = []
    old_slot = ops.set_workspace_slot(1)  # (these launches take no scratch; the second queue gets its own slot all the same)
    try:
        with torch.cuda.stream(side):
            for _ in range(40):
                ys.append(ops.conv2d_fwd_f16x2(xh, wh2, geom, bd, s, p, **kw))
    finally:
        ops.set_workspace_slot(old_slot)
    for _ in range(24):  # ~0.1 ms each: the hog outlasts the convs on the other queue
        ops.affine_act(big, sc, sh, act=ops.ACT_SWISH, out=out)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    for i, y in enumerate(ys):
        assert torch.equal(y, ref), "launch %d differs from the idle-device result" % i


def test_f16x2_range_is_loud_not_silent(dev):
    """|x| * 2^e beyond fp16's range must poison the output (inf - inf = NaN), never pass silently; a correct amax bound
    makes the same data exact again."""
    x, w, b = _h2_setup(1, 128, 128, 32, 3, 1, 1, seed=40)
    x = x * 5000.0  # static exponent 6: |x| * 64 overflows fp16
    xh, wp = nhwc(x).to(dev), ops.pack_conv_weight(w.to(dev))
    wh2, geom = ops.pack_weight_f16x2(wp), tuple(wp.shape)
    y = ops.conv2d_fwd_f16x2(xh, wh2, geom, None, 1, 1)
    assert not bool(torch.isfinite(y).all())
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y = nchw(ops.conv2d_fwd_f16x2(xh, wh2, geom, None, 1, 1, x_amax=torch.full((1,), float(x.abs().max()), device=dev)))
    close(y, ref, tol=2e-6)


def test_f16x2_rejects_unsupported_shapes(dev):
    x = torch.zeros(1, 24, 24, 128, device=dev)
    assert not ops.f16x2_ok(x, 128, 3, 3, 1, 1)
    wh2 = ops.pack_weight_f16x2(torch.zeros(128, 3, 3, 128, device=dev))
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.conv2d_fwd_f16x2(x, wh2, (128, 3, 3, 128), None, 1, 1)
    with pytest.raises(RuntimeError, match="pre_scale"):
        from oxford_102_flower_gan_vae_latent_diffusion_amd import lib as _l
        xx = torch.zeros(1, 32, 32, 128, device=dev)
        yy = torch.zeros(1, 32, 32, 128, device=dev)
        import ctypes
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _l.call("fl_conv2d_fwd_f16x2", P(xx), P(wh2), None, P(yy), 1, 32, 32, 128, 128, 3, 3, 1, 1, 0, 0.0, None, None, 1e-5, None, None,
                None, None, None, 6, P(xx), None, None, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def test_encoder_infer_f16x2_fused_vs_oracle(dev):
    """the frozen encoder at a batch where every conv runs on the f16x2 kernels with the fused pre-op (B*8*8 % 256 == 0),
    against the CPU oracle, with and without materialised skip features."""
    from oracle import procedural_init as pinit, v2_oracle as O
    from oxford_102_flower_gan_vae_latent_diffusion_amd import engine, modules as M
    ae = M.SimpleAutoencoder()
    ae.load_state_dict(pinit.procedural_state_dict(ae.state_dict(), 42))
    sd = {k: v.clone() for k, v in ae.state_dict().items()}
    ae = ae.to(dev).eval()
    B = 8
    data = torch.rand(B, 3, 64, 64, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        mu_r, lv_r, feats_r = O.encoder_forward(sd, "encoder.", data, return_features=True)
        mu, lv, feats = engine.encoder_infer(ae.encoder, data.to(dev), want_feats=True)
        mu2, lv2, feats2 = engine.encoder_infer(ae.encoder, data.to(dev), want_feats=False)
    close(mu, mu_r, tol=1e-4)
    close(lv, lv_r, tol=1e-4)
    assert torch.equal(mu, mu2) and torch.equal(lv, lv2) and feats2[0] is None
    for f, fr in zip(feats, feats_r):
        close(f.permute(0, 3, 1, 2), fr, tol=1e-4)


@pytest.mark.parametrize("B,C_in,C_out,H", [(64, 128, 128, 32), (64, 256, 256, 16), (256, 512, 512, 8), (64, 64, 128, 32)])
def test_f16x2_training_convs(dev, B, C_in, C_out, H):
    """The training paths' 3x3 s1 convs on the f16x2 kernel: forward with the measured input range (fl_absmax) and the input
    gradient as the mirror convolution (taps flipped, channel roles swapped) vs torch fp64. Bar: 1e-5 of the output max
    (the kernel's error class, test_f16x2_conv_is_fp32_equivalent)."""
    import types
    from oxford_102_flower_gan_vae_latent_diffusion_amd import engine, engine_bwd
    conv = torch.nn.Conv2d(C_in, C_out, 3, 1, 1).to(dev)
    with torch.no_grad():
        conv.weight.copy_(rnd(C_out, C_in, 3, 3, seed=1, scale=0.05))
        conv.bias.copy_(rnd(C_out, seed=2))
    pk = engine.PackCache()
    x = rnd(B, C_in, H, H, seed=3, scale=3.0)
    dy = rnd(B, C_out, H, H, seed=4, scale=1e-3)              # gradients are small and badly scaled
    dy[:, :, 0, 0] *= 300.0
    xh, dyh = nhwc(x).to(dev), nhwc(dy).to(dev)
    assert float(ops.absmax(xh)) == float(x.abs().max())
    assert engine_bwd._h2_ok(xh, C_out) and engine_bwd._h2_ok(dyh, C_in) == (C_in % 128 == 0)   # else: the bf16x3 dgrad
    y = engine_bwd.conv3_fwd_act(xh, conv, pk, ops.ACT_RELU)
    ref = F.relu(F.conv2d(x.double(), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), 1, 1))
    close(nchw(y), ref, 1e-5)
    dx = engine_bwd.conv3_dgrad(dyh, conv, (H, H), pk)
    dref = F.conv_transpose2d(dy.double(), conv.weight.detach().cpu().double(), None, 1, 1)
    a, b = nchw(dx).cpu().double(), dref
    assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    # the pack follows the weight version
    with torch.no_grad():
        conv.weight.mul_(2.0)
    dx2 = engine_bwd.conv3_dgrad(dyh, conv, (H, H), pk)
    assert (nchw(dx2).cpu().double() - 2 * b).abs().max().item() <= 2e-5 * b.abs().max().item()


@pytest.mark.parametrize("M,N,K,nb", [(256, 1024, 32768, 512), (256, 128, 192, 128), (512, 256, 64, 64), (256, 128, 32, 128)])
def test_f16x2_linear_long_reduction(dev, M, N, K, nb):
    """fl_linear_fwd_f16x2 (the frozen encoder heads' 32768-long contraction, v2:210-221) vs fp64: split-K slabs + reduce,
    concatenated heads written as separate matrices, one- and two-chunk splits. Bar: 3e-6 of the output max at K = 32768
    (the conv kernels' error class), and it must beat-or-match the native fp32 MFMA GEMM's error."""
    x = rnd(M, K, seed=1, scale=1.5)
    x = x * torch.sigmoid(x)
    w = rnd(N, K, seed=2, scale=(1.0 / K) ** 0.5)
    b = rnd(N, seed=3)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    assert ops.linear_f16x2_ok(M, N, K)
    wh = ops.pack_weight_f16x2(wd)
    y = ops.linear_fwd_f16x2(xd, wh, N, bd, n_block=nb, x_amax=ops.absmax(xd))
    ref = x.double() @ w.double().t() + b.double()
    if nb != N:
        assert y.shape == (N // nb, M, nb)
        y = torch.cat([y[i] for i in range(N // nb)], 1)
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 3e-6, err
    y2 = ops.linear_fwd_f16x2(xd, wh, N, bd, n_block=N, x_amax=ops.absmax(xd))
    assert torch.equal(y2, y.to(dev)) if nb != N else True     # deterministic, layout-independent
    assert not ops.linear_f16x2_ok(100, N, K) and not ops.linear_f16x2_ok(M, 96, K)


@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 4), (3, 16, 16, 128, 8), (1, 8, 4, 96, 4), (2, 16, 16, 128, 4), (1, 16, 16, 96, 4),
                                   (1, 32, 16, 64, 2)])
def test_fused_attention_backward_vs_autograd(dev, shape):
    """fl_attention_bwd (dQ kernel + dK / dV kernel, probabilities recomputed from the forward's row statistics, no HW x HW tensor
    in HBM) against torch fp64 autograd of softmax(scale Q K^T) V with the d-major head merge of v2:456-457; head dims 16, 16, 24,
    32, 24, 32: the LDS-staged 16x16x4 kernels (HW % 64 == 0; 512 tokens = two LDS chunks) and the round-3a kernels they fall
    back to (HW = 32)."""
    B, H, W, C, nh = shape
    HW, d = H * W, C // nh
    g = torch.Generator().manual_seed(B * 100 + C)
    qkv = torch.randn(B, H, W, 3 * C, generator=g)
    dmerged = torch.randn(B, H, W, C, generator=g)
    scale = float(d) ** -0.5
    q64 = qkv.double().requires_grad_(True)
    t = q64.reshape(B, HW, 3, nh, d).permute(2, 0, 3, 1, 4)                      # (3, B, nh, HW, d)
    att = torch.softmax(scale * t[0] @ t[1].transpose(-1, -2), dim=-1) @ t[2]     # (B, nh, HW, d)
    merged_ref = att.permute(0, 2, 3, 1).reshape(B, H, W, C)                      # channel = i * heads + head
    (merged_ref * dmerged.double()).sum().backward()
    qd = qkv.to(dev)
    merged, st = ops.attention_fwd(qd, nh, scale, want_stats=True)
    close(merged, merged_ref.detach(), 2e-5)
    dq = ops.attention_bwd(qd, merged, dmerged.to(dev), st, nh, scale)
    for which, name in enumerate(("dQ", "dK", "dV")):
        close(dq[..., which * C:(which + 1) * C], q64.grad[..., which * C:(which + 1) * C], 3e-5)   # dQ, dK, dV


@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 4), (3, 16, 16, 128, 8), (2, 16, 16, 128, 4), (1, 32, 32, 64, 2), (2, 8, 8, 128, 2),
                                   (1, 16, 16, 96, 4), (1, 8, 4, 96, 4), (1, 16, 8, 192, 4)])
def test_attention_forward_onepass_vs_fp64(dev, shape):
    """fl_attention_fwd: the one-pass kernel (online softmax, 16x16x4 fp32 MFMAs, K / V^T in LDS; head dims 16, 16, 32, 32, 64, 24,
    and 1024 keys = four LDS chunks) and the two-pass kernel it falls back to (HW = 32; P output) against torch fp64 softmax(scale
    Q K^T) V with the d-major head merge of v2:456-457: 2e-5 of the output's max (exp2 on the transcendental pipe: ~1 ulp per
    weight); the row statistics reproduce the weights: exp(scale s - m) / l."""
    B, H, W, C, nh = shape
    HW, d = H * W, C // nh
    g = torch.Generator().manual_seed(B * 100 + C + HW)
    qkv = torch.randn(B, H, W, 3 * C, generator=g) * 1.5
    scale = float(d) ** -0.5
    t = qkv.double().reshape(B, HW, 3, nh, d).permute(2, 0, 3, 1, 4)               # (3, B, nh, HW, d)
    logits = scale * t[0] @ t[1].transpose(-1, -2)
    P = torch.softmax(logits, dim=-1)
    ref = (P @ t[2]).permute(0, 2, 3, 1).reshape(B, H, W, C)                        # channel = i * heads + head
    qd = qkv.to(dev)
    merged, st = ops.attention_fwd(qd, nh, scale, want_stats=True)
    close(merged, ref, 2e-5)
    st = st.cpu().double()
    Pst = torch.exp(logits - st[..., 0:1]) / st[..., 1:2]
    assert (Pst - P).abs().max().item() <= 2e-6
    if d <= 32:   # the P output takes the two-pass kernel: same numbers
        merged2, P2 = ops.attention_fwd(qd, nh, scale, want_p=True)
        close(merged2, ref, 2e-5)
        assert (P2.cpu().double() - P).abs().max().item() <= 2e-6


@pytest.mark.parametrize("B,C,H", [(64, 128, 32), (128, 256, 16), (256, 512, 8)])
def test_f16x2_pair_kernels_on_full_256_pixel_tiles(dev, B, C, H):
    """The headline configuration's 3x3 kernels - `conv3x3_h2p_kernel<W, PRE, 256>`, 256-pixel tiles on >= 256 tiles (the small cases
    above run its 128-pixel variant) - with and without the fused LayerNorm2d-apply + Swish pre-op, against the native fp32-MFMA
    implicit-GEMM kernel on the WHOLE tensor (5e-6 of the output's max: each side is <= 3e-6 from fp64 at K = 4608) and its fused LayerNorm2d statistics
    (finished in-kernel for the 16 / 8-wide maps, by the finalize launch for the 32-wide one) against the reference path's."""
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, H, H, C, generator=g)
    x = (x * torch.sigmoid(x)).to(dev)
    w = (torch.randn(C, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev)
    b = (torch.randn(C, generator=g) * 0.1).to(dev)
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    wh2, geom = ops.pack_weight_f16x2(w), tuple(w.shape)
    assert (B * H * H // 256) * (C // 128) >= 256
    amax = (x.abs().max() * 1.25).reshape(1)
    old = ops.get_gemm_math()
    ops.set_gemm_math("fp32")
    y32, (sc32, sh32, _, _) = ops.conv2d_fwd_instnorm(x, w, b, 1, 1, lnw, lnb, 1e-5)
    ops.set_gemm_math(old)
    y, (sc, sh, _, _) = ops.conv2d_fwd_f16x2(x, wh2, geom, b, 1, 1, ln_w=lnw, ln_b=lnb, x_amax=amax)
    scale = y32.abs().max().item()
    assert (y - y32).abs().max().item() <= 5e-6 * scale
    assert (sc - sc32).abs().max().item() <= 1e-5 * sc32.abs().max().item()
    assert (sh - sh32).abs().max().item() <= 2e-5 * max(1.0, sh32.abs().max().item())
    # fused pre-op: conv(swish(x * s + t)) == conv of the materialised activation
    ps, pt = (torch.rand(B, C, generator=g) + 0.5).to(dev), (torch.randn(B, C, generator=g) * 0.3).to(dev)
    xa = ops.affine_act(x, ps, pt, act=ops.ACT_SWISH)
    bnd = (xa.abs().max() * 1.25).reshape(1)
    ya = ops.conv2d_fwd_f16x2(xa, wh2, geom, b, 1, 1, x_amax=bnd)
    yp = ops.conv2d_fwd_f16x2(x, wh2, geom, b, 1, 1, x_amax=bnd, pre=(ps, pt))
    assert (yp - ya).abs().max().item() <= 4e-6 * ya.abs().max().item()   # (the conv-side Swish uses v_exp / v_rcp: a few ulp)
    assert torch.isfinite(yp).all()
