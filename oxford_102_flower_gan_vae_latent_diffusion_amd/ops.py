"""Tensor-level wrappers over the C ABI (include/flowers_hip.h).

PyTorch is used for device memory (torch.empty on the current device), the current HIP
stream and nothing else: every arithmetic op below is a hand-written gfx950 kernel in
libflowers_hip.so. All tensors are fp32, contiguous, on a CUDA(HIP) device; activations
are NHWC. Functions raise RuntimeError if the library reports an error; there is no
fallback.
"""
import ctypes

import torch

from . import lib as _lib

ACT_NONE, ACT_SWISH, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3, 4
KC, KS = 0, 1

_workspaces = {}
_retired = []  # superseded scratch buffers (kept alive: graphs captured before the growth still write into them)
_ws_slot = 0


def set_workspace_slot(slot):
    """Scratch buffers are per (device, slot): work captured/launched on concurrently running streams must use
    different slots (the split-K slabs / norm partials of one stream would otherwise race with the other's)."""
    global _ws_slot
    old, _ws_slot = _ws_slot, int(slot)
    return old


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("flowers_hip ops need device tensors (got %s); there is no CPU path" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("flowers_hip ops need contiguous tensors")
        if t.is_floating_point() and t.dtype != torch.float32:
            raise TypeError("flowers_hip kernels are fp32 (got %s)" % t.dtype)


def workspace(nbytes, device=None):
    """Grow-only per-device scratch buffer (split-K slabs, norm partials)."""
    device = device if device is not None else torch.cuda.current_device()
    if isinstance(device, torch.device):
        device = device.index if device.index is not None else torch.cuda.current_device()
    nbytes = max(int(nbytes), 1 << 20)
    device = (device, _ws_slot)
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("workspace would grow during graph capture; run a warm-up step first")
        if ws is not None:
            _retired.append(ws)  # a captured hipGraph may have the old pointer baked in: never hand it back to the allocator
        ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device="cuda:%d" % device[0])
        _workspaces[device] = ws
    return ws


def _ws(nbytes, like):
    ws = workspace(nbytes, like.device)
    return ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())


def new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------- GEMM
def set_gemm_math(mode):
    """'split' (default): exact 3-way bf16 split of the fp32 operands, 6 piece products on the bf16 matrix pipe;
    'fp32': native fp32 MFMA. Applies to every GEMM / conv launched afterwards (re-capture graphs after a change)."""
    _lib.call("fl_set_gemm_math", {"fp32": 0, "split": 1}[mode] if isinstance(mode, str) else int(mode))


def set_gemm_splitk_tiles(tiles):
    """GEMM grids with fewer tiles than this are split along K (default 160). Returns the previous value."""
    old = int(_lib.load().fl_get_gemm_splitk_tiles())
    _lib.call("fl_set_gemm_splitk_tiles", int(tiles))
    return old


def set_conv_halo_mode(mode):
    """Variant of the 3x3 halo conv kernel of the pre-split path: 0 = 64x64 wave tiles, two blocks per CU (fastest
    alone); 1 = 32x64 wave tiles, one 8-wave block per CU, leaving registers / LDS for a concurrent stream's waves.
    Returns the previous mode."""
    old = int(_lib.load().fl_get_conv_halo_mode())
    _lib.call("fl_set_conv_halo_mode", int(mode))
    return old


def get_gemm_math():
    return "split" if _lib.load().fl_get_gemm_math() else "fp32"


def gemm(A, B, M, N, K, a_layout=KC, b_layout=KC, lda=None, ldb=None, bias=None, bias_scale=1.0, R=None,
         r_scale=1.0, r_pre=False, act=ACT_NONE, act_param=0.0, out=None, ldc=None):
    """C[m,n] = epi(sum_k A(m,k) B(n,k)); see flowers_hip.h fl_gemm_f32. ldc: row stride of `out` when it is a
    column block of a wider matrix (out is then any tensor whose data_ptr is C[0,0]; no residual R)."""
    _chk(A, B, bias, R, out)
    if ldc is not None:
        if out is None or R is not None:
            raise ValueError("gemm: ldc needs an explicit out and no residual")
        need = _lib.load().fl_gemm_workspace_bytes(M, N, K)
        wsp, wsn = _ws(min(need, 256 << 20), A)
        _lib.call("fl_gemm_f32", _p(A), lda if lda is not None else (K if a_layout == KC else M), a_layout, _p(B),
                  ldb if ldb is not None else (K if b_layout == KC else N), b_layout, _p(out), int(ldc), M, N, K, _p(bias),
                  float(bias_scale), None, N, float(r_scale), int(r_pre), act, float(act_param), wsp, wsn, _stream())
        return out
    if lda is None:
        lda = K if a_layout == KC else M
    if ldb is None:
        ldb = K if b_layout == KC else N
    if out is None:
        out = new((M, N), A)
    need = _lib.load().fl_gemm_workspace_bytes(M, N, K)
    wsp, wsn = _ws(min(need, 256 << 20), A)
    _lib.call("fl_gemm_f32", _p(A), lda, a_layout, _p(B), ldb, b_layout, _p(out), N, M, N, K, _p(bias),
              float(bias_scale), _p(R), N, float(r_scale), int(r_pre), act, float(act_param), wsp, wsn, _stream())
    return out


def linear(x, w, b=None, act=ACT_NONE, R=None, r_pre=False, bias_scale=1.0, r_scale=1.0, out=None):
    """nn.Linear forward: x (M,K) @ w(N,K)^T + b."""
    M, K = x.shape
    N = w.shape[0]
    return gemm(x, w, M, N, K, bias=b, act=act, R=R, r_pre=r_pre, bias_scale=bias_scale, r_scale=r_scale, out=out)


def linear_dgrad(dy, w, out=None):
    """dx (M,K) = dy (M,N) @ w (N,K)."""
    M, N = dy.shape
    K = w.shape[1]
    return gemm(dy, w, M, K, N, a_layout=KC, b_layout=KS, lda=N, ldb=K, out=out)


def linear_wgrad(dy, x, out=None):
    """dw (N,K) = dy^T (N,M) @ x (M,K)."""
    M, N = dy.shape
    K = x.shape[1]
    return gemm(dy, x, N, K, M, a_layout=KS, b_layout=KS, lda=N, ldb=K, out=out)


def colsum(x, out=None, alpha=1.0, beta=0.0):
    M, N = x.shape
    _chk(x, out)
    if out is None:
        out = new((N,), x)
    wsp, wsn = _ws(_lib.load().fl_colsum_workspace_bytes(M, N), x)
    _lib.call("fl_colsum", _p(x), N, _p(out), M, N, float(alpha), float(beta), wsp, wsn, _stream())
    return out


def gated_add(a, b, w, out=None):
    """out = a + sigmoid(w[0]) * b (a may be None)."""
    _chk(a, b, w, out)
    if out is None:
        out = torch.empty_like(b)
    _lib.call("fl_gated_add", _p(a), _p(b), _p(w), _p(out), b.numel(), _stream())
    return out


def gate_grad(x, y, w, out=None):
    _chk(x, y, w, out)
    if out is None:
        out = new((1,), x)
    _lib.call("fl_gate_grad", _p(x), _p(y), _p(w), _p(out), x.numel(), _stream())
    return out


# --------------------------------------------------------------------------- conv
def pack_conv_weight(w, Ipad=None, Opad=None):
    """(O,I,kh,kw) -> packed [Opad][kh][kw][Ipad] (zero padded)."""
    _chk(w)
    O, I, kh, kw = w.shape
    Ipad = Ipad or I
    Opad = Opad or O
    wp = torch.zeros((Opad, kh, kw, Ipad), dtype=torch.float32, device=w.device) if Opad != O else new((O, kh, kw, Ipad), w)
    _lib.call("fl_pack_conv_weight", _p(w), _p(wp), O, I, kh, kw, Ipad, _stream())
    return wp


def unpack_conv_weight(wp, O, I, kh, kw):
    _chk(wp)
    Ipad = wp.shape[3]
    w = new((O, I, kh, kw), wp)
    _lib.call("fl_unpack_conv_weight", _p(wp), _p(w), O, I, kh, kw, Ipad, _stream())
    return w


def conv2d_fwd(x, wp, bias, stride, pad, act=ACT_NONE, act_param=0.0, R=None, out_nchw=False, out=None):
    """x (B,H,W,Cin) NHWC, wp [Cout][kh][kw][Cin] -> (B,Ho,Wo,Cout)."""
    _chk(x, wp, bias, R, out)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = wp.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = new((B, Cout, Ho, Wo) if out_nchw else (B, Ho, Wo, Cout), x)
    need = _lib.load().fl_gemm_workspace_bytes(B * Ho * Wo, Cout, kh * kw * Cin)
    wsp, wsn = _ws(min(need, 256 << 20), x)
    _lib.call("fl_conv2d_fwd", _p(x), _p(wp), _p(bias), _p(out), B, H, W, Cin, Cout, kh, kw, stride, pad, act,
              float(act_param), _p(R), int(out_nchw), wsp, wsn, _stream())
    return out


def conv2d_fwd_instnorm(x, wp, bias, stride, pad, ln_w, ln_b, eps=1e-5):
    """conv (+bias) whose epilogue also accumulates the LayerNorm2d statistics of the output.
    -> y (B,Ho,Wo,Cout), (scale, shift, mean, rstd) each (B,Cout)."""
    _chk(x, wp, bias, ln_w, ln_b)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = wp.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if (Ho * Wo) % 64 != 0:  # tiny maps: separate statistics pass
        y = conv2d_fwd(x, wp, bias, stride, pad)
        return y, instnorm_stats(y, ln_w, ln_b, eps)
    y = new((B, Ho, Wo, Cout), x)
    scale, shift, mean, rstd = (new((B, Cout), x) for _ in range(4))
    wsp, wsn = _ws(_lib.load().fl_conv2d_fwd_instnorm_workspace_bytes(B, Ho, Wo, Cout), x)
    _lib.call("fl_conv2d_fwd_instnorm", _p(x), _p(wp), _p(bias), _p(y), B, H, W, Cin, Cout, kh, kw, stride, pad, _p(ln_w),
              _p(ln_b), float(eps), _p(scale), _p(shift), _p(mean), _p(rstd), wsp, wsn, _stream())
    return y, (scale, shift, mean, rstd)


def split_pack_weight(wp):
    """packed conv weight [O][kh][kw][I] (or any [N][K] K-contiguous matrix) -> its three bf16 pieces in MFMA-fragment
    order (uint8 tensor of 6*N*K bytes) for conv2d_fwd_presplit. N % 32 == 0, K % 16 == 0."""
    _chk(wp)
    N = wp.shape[0]
    K = wp.numel() // N
    out = torch.empty(_lib.load().fl_split_pack_weight_bytes(N, K), dtype=torch.uint8, device=wp.device)
    _lib.call("fl_split_pack_weight", _p(wp), _p(out), N, K, _stream())
    return out


def conv2d_fwd_presplit(x, wsplit, geom, bias, stride, pad, act=ACT_NONE, act_param=0.0, ln_w=None, ln_b=None, eps=1e-5,
                        out=None):
    """conv forward with a pre-split weight (geom = (Cout, kh, kw, Cin) of the packed weight it was made from).
    With ln_w/ln_b: also the fused LayerNorm2d statistics -> (y, (scale, shift, mean, rstd)); else y."""
    _chk(x, bias, ln_w, ln_b, out)
    B, H, W, Cin = x.shape
    Cout, kh, kw, Ci2 = geom
    assert Ci2 == Cin
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = out if out is not None else new((B, Ho, Wo, Cout), x)
    if ln_w is None:
        _lib.call("fl_conv2d_fwd_presplit", _p(x), _p(wsplit), _p(bias), _p(y), B, H, W, Cin, Cout, kh, kw, stride, pad, act,
                  float(act_param), None, None, float(eps), None, None, None, None, None, 0, _stream())
        return y
    scale, shift, mean, rstd = (new((B, Cout), x) for _ in range(4))
    wsp, wsn = _ws(_lib.load().fl_conv2d_fwd_instnorm_workspace_bytes(B, Ho, Wo, Cout), x)
    _lib.call("fl_conv2d_fwd_presplit", _p(x), _p(wsplit), _p(bias), _p(y), B, H, W, Cin, Cout, kh, kw, stride, pad, act,
              float(act_param), _p(ln_w), _p(ln_b), float(eps), _p(scale), _p(shift), _p(mean), _p(rstd), wsp, wsn, _stream())
    return y, (scale, shift, mean, rstd)


def pack_weight_f16x2(wp):
    """packed conv weight [O][kh][kw][I] (3x3 or 4x4) -> header + fp16 (h | l) pieces in MFMA-fragment order for
    conv2d_fwd_f16x2 (fl_conv2d_f16x2_weight_layout), or a 2-D [N][K] operand of linear_fwd_f16x2 (layout 0)."""
    _chk(wp)
    N = wp.shape[0]
    K = wp.numel() // N
    # conv weights: the layout the conv kernel of that shape reads (4x4: 1; 3x3: 2 = tap pairs for the 16x16x32 MFMA when Cin % 64 == 0)
    layout = _lib.load().fl_conv2d_f16x2_weight_layout(int(wp.shape[1]), int(wp.shape[3])) if wp.dim() == 4 else 0
    out = torch.empty(_lib.load().fl_pack_weight_f16x2_bytes(N, K), dtype=torch.uint8, device=wp.device)
    _lib.call("fl_pack_weight_f16x2", _p(wp), _p(out), N, K, layout, _stream())
    return out


def program_pack_weight(w, N, K, ldw, offset=0):
    """[N][K] weight (row pitch ldw, starting `offset` floats into w) -> the tile-contiguous image FL_PG_WPACK jobs read."""
    _chk(w)
    out = torch.empty(N * K, dtype=torch.float32, device=w.device)
    _lib.call("fl_program_pack_weight", ctypes.c_void_p(w.data_ptr() + 4 * offset), _p(out), N, K, ldw, _stream())
    return out


def f16x2_ok(x, Cout, kh, kw, stride, pad):
    B, H, W, Cin = x.shape
    return bool(_lib.load().fl_conv2d_fwd_f16x2_applicable(_p(x), B, H, W, Cin, Cout, kh, kw, stride, pad))


def conv2d_fwd_f16x2(x, wpack, geom, bias, stride=1, pad=1, act=ACT_NONE, act_param=0.0, ln_w=None, ln_b=None, eps=1e-5,
                     out=None, x_amax=None, x_exp=6, pre=None):
    """3x3 s1 p1 / 4x4 s2 p1 conv forward in f16x2 split math (geom = (Cout, kh, kw, Cin) of the packed weight). x_amax: device float
    bounding max|x| (else activations are scaled by 2^x_exp). With ln_w/ln_b: also the fused LayerNorm2d statistics.
    pre = (scale, shift) (B, Cin): the conv consumes swish(x * scale[b,c] + shift[b,c]) (x_amax bounds THAT)."""
    ps, ph = pre if pre is not None else (None, None)
    _chk(x, bias, ln_w, ln_b, out, x_amax, ps, ph)
    B, H, W, Cin = x.shape
    Cout, kh, kw, Ci2 = geom
    assert Ci2 == Cin
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = out if out is not None else new((B, Ho, Wo, Cout), x)
    if ln_w is None:
        _lib.call("fl_conv2d_fwd_f16x2", _p(x), _p(wpack), _p(bias), _p(y), B, H, W, Cin, Cout, kh, kw, stride, pad, act,
                  float(act_param), None, None, float(eps), None, None, None, None, _p(x_amax), int(x_exp), _p(ps), _p(ph), None, 0,
                  _stream())
        return y
    scale, shift, mean, rstd = (new((B, Cout), x) for _ in range(4))
    wsp, wsn = _ws(_lib.load().fl_conv2d_fwd_instnorm_workspace_bytes(B, Ho, Wo, Cout), x)
    _lib.call("fl_conv2d_fwd_f16x2", _p(x), _p(wpack), _p(bias), _p(y), B, H, W, Cin, Cout, kh, kw, stride, pad, act,
              float(act_param), _p(ln_w), _p(ln_b), float(eps), _p(scale), _p(shift), _p(mean), _p(rstd), _p(x_amax), int(x_exp),
              _p(ps), _p(ph), wsp, wsn, _stream())
    return y, (scale, shift, mean, rstd)


def linear_f16x2_ok(M, N, K):
    return bool(_lib.load().fl_linear_fwd_f16x2_applicable(int(M), int(N), int(K)))


def linear_fwd_f16x2(x, wpack, N, bias=None, n_block=None, x_amax=None, x_exp=6):
    """x (M, K) @ W^T + bias with W (N, K) packed by pack_weight_f16x2; returns (N / n_block, M, n_block) (n_block = N: (M, N))."""
    _chk(x, bias, x_amax)
    M, K = x.shape
    nb = N if n_block is None else n_block
    y = new((N // nb, M, nb) if nb != N else (M, N), x)
    wsp, wsn = _ws(_lib.load().fl_linear_fwd_f16x2_workspace_bytes(M, N, K), x)
    _lib.call("fl_linear_fwd_f16x2", _p(x), K, _p(wpack), _p(bias), _p(y), nb, M, N, K, _p(x_amax), int(x_exp), wsp, wsn, _stream())
    return y


def absmax(x):
    """(1,) device float = max|x| (fl_absmax): range information of the f16x2 convs for training activations / gradients."""
    _chk(x)
    out = new((1,), x)
    _lib.call("fl_absmax", _p(x), x.numel(), _p(out), _stream())
    return out


def affine_bound(w, b, mult, prev=None):
    """(1,) device float: mult * max|w| + max|b| (+ prev) - bound on |LayerNorm-style output| (see fl_affine_bound)."""
    ref = w if w is not None else b
    _chk(w, b, prev)
    out = new((1,), ref)
    _lib.call("fl_affine_bound", _p(w), _p(b), ref.numel(), float(mult), _p(prev), _p(out), _stream())
    return out


def presplit_ok(Cin, Cout, HoWo):
    return Cin % 32 == 0 and Cout % 128 == 0 and HoWo % 64 == 0


def conv2d_dgrad(dy, wp, in_hw, stride, pad, out=None):
    _chk(dy, wp, out)
    B, Ho, Wo, Cout = dy.shape
    _, kh, kw, Cin = wp.shape
    H, W = in_hw
    if out is None:
        out = new((B, H, W, Cin), dy)
    wsp, wsn = _ws(1 << 20, dy)
    _lib.call("fl_conv2d_dgrad", _p(dy), _p(wp), _p(out), B, H, W, Cin, Cout, kh, kw, stride, pad, wsp, wsn, _stream())
    return out


def conv2d_wgrad(dy, x, kh, kw, stride, pad, out=None):
    _chk(dy, x, out)
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    if out is None:
        out = new((Cout, kh, kw, Cin), x)
    need = _lib.load().fl_conv2d_wgrad_workspace_bytes(B, H, W, Cin, Cout, kh, kw, stride, pad)
    wsp, wsn = _ws(min(need, 256 << 20), x)
    _lib.call("fl_conv2d_wgrad", _p(dy), _p(x), _p(out), B, H, W, Cin, Cout, kh, kw, stride, pad, wsp, wsn, _stream())
    return out


def convT2d_fwd(x, wtp, bias, act=ACT_NONE, act_param=0.0, out=None):
    """ConvTranspose2d(4,2,1): x (B,H,W,Cin), wtp [Cin][4][4][Cout] -> (B,2H,2W,Cout)."""
    _chk(x, wtp, bias, out)
    B, H, W, Cin = x.shape
    Cout = wtp.shape[3]
    if out is None:
        out = new((B, 2 * H, 2 * W, Cout), x)
    wsp, wsn = _ws(1 << 20, x)
    _lib.call("fl_convT2d_fwd", _p(x), _p(wtp), _p(bias), _p(out), B, H, W, Cin, Cout, act, float(act_param), wsp, wsn,
              _stream())
    return out


# --------------------------------------------------------------------------- norms
def _norm_ws(B, HW, C, like):
    return _ws(_lib.load().fl_norm_workspace_bytes(B, HW, C), like)


def instnorm_stats(x, w, b, eps=1e-5):
    """-> scale, shift, mean, rstd, each (B,C): LayerNorm2d(x) == x*scale + shift."""
    _chk(x, w, b)
    B, H, W, C = x.shape
    scale, shift, mean, rstd = (new((B, C), x) for _ in range(4))
    wsp, wsn = _norm_ws(B, H * W, C, x)
    _lib.call("fl_instnorm_stats", _p(x), _p(w), _p(b), float(eps), _p(scale), _p(shift), _p(mean), _p(rstd), B, H * W,
              C, wsp, wsn, _stream())
    return scale, shift, mean, rstd


def groupnorm_stats(x, w, b, G, eps=1e-5):
    _chk(x, w, b)
    B, H, W, C = x.shape
    scale, shift = new((B, C), x), new((B, C), x)
    mean, rstd = new((B, G), x), new((B, G), x)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    _lib.call("fl_groupnorm_stats", _p(x), _p(w), _p(b), float(eps), G, _p(scale), _p(shift), _p(mean), _p(rstd), B,
              H * W, C, wsp, wsn, _stream())
    return scale, shift, mean, rstd


def batchnorm_stats(x, w, b, running_mean, running_var, training, momentum=0.1, eps=1e-5, sync=None, world=1):
    """sync: callable all-reducing (SUM) a float64 tensor in place across `world` ranks -> statistics of
    the global batch (SyncBatchNorm rule); None -> this rank's batch only."""
    _chk(x, w, b, running_mean, running_var)
    B, H, W, C = x.shape
    scale, shift, mean, rstd = (new((C,), x) for _ in range(4))
    wsp, wsn = _norm_ws(B, H * W, C, x)
    if sync is not None and training:
        sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        _lib.call("fl_batchnorm_sums", _p(x), None, None, None, 0, 0.0, _p(sums), B, H * W, C, wsp, wsn, _stream())
        sync(sums)
        _lib.call("fl_batchnorm_stats_from_sums", _p(sums), int(world) * B * H * W, _p(w), _p(b), float(eps), float(momentum),
                  _p(running_mean), _p(running_var), _p(scale), _p(shift), _p(mean), _p(rstd), C, _stream())
        return scale, shift, mean, rstd
    _lib.call("fl_batchnorm_stats", _p(x), _p(w), _p(b), float(eps), float(momentum), _p(running_mean),
              _p(running_var), int(training), _p(scale), _p(shift), _p(mean), _p(rstd), B, H * W, C, wsp, wsn, _stream())
    return scale, shift, mean, rstd


def batchnorm_stats_pair(x, w, b, running_mean, running_var, momentum=0.1, eps=1e-5, sync=None, world=1):
    """Training-mode BatchNorm statistics of TWO batches stacked along dim 0 (x: (2B,H,W,C) = [first | second], e.g. the real
    and the fake pass of the discriminator update, v2:1097-1098): each half gets its own statistics and updates the running
    buffers in order (first, then second - what two separate passes do), and with `sync` both halves' sums travel in ONE
    all-reduce (4C doubles) instead of one per pass. -> (scale, shift, mean, rstd) of the first half, of the second half."""
    _chk(x, w, b, running_mean, running_var)
    B2, H, W, C = x.shape
    assert B2 % 2 == 0, "two stacked batches"
    B = B2 // 2
    halves = (x[:B], x[B:])
    if sync is None:
        return tuple(batchnorm_stats(h, w, b, running_mean, running_var, True, momentum, eps) for h in halves)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    sums = torch.empty(2, 2 * C, device=x.device, dtype=torch.float64)
    for i, h in enumerate(halves):
        _lib.call("fl_batchnorm_sums", _p(h), None, None, None, 0, 0.0, _p(sums[i]), B, H * W, C, wsp, wsn, _stream())
    sync(sums)
    out = []
    for i in range(2):
        scale, shift, mean, rstd = (new((C,), x) for _ in range(4))
        _lib.call("fl_batchnorm_stats_from_sums", _p(sums[i]), int(world) * B * H * W, _p(w), _p(b), float(eps), float(momentum),
                  _p(running_mean), _p(running_var), _p(scale), _p(shift), _p(mean), _p(rstd), C, _stream())
        out.append((scale, shift, mean, rstd))
    return tuple(out)


def affine_act(x, scale, shift, act=ACT_NONE, act_param=0.0, R=None, out=None):
    """y = act(x*scale + shift) (+R); scale/shift (B,C) or (C,)."""
    _chk(x, scale, shift, R, out)
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.call("fl_affine_act", _p(x), _p(scale), _p(shift), int(scale.dim() == 2), _p(R), _p(out), B, H * W, C, act,
              float(act_param), _stream())
    return out


def instnorm_bwd(dy, x, scale, shift, w, mean, rstd, act=ACT_NONE, act_param=0.0, need_wb=True):
    _chk(dy, x, scale, shift, w, mean, rstd)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    dw = new((C,), x) if need_wb else None
    db = new((C,), x) if need_wb else None
    coef = new((3, B, C), x)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    _lib.call("fl_instnorm_bwd", _p(dy), _p(x), _p(scale), _p(shift), _p(w), _p(mean), _p(rstd), act, float(act_param),
              _p(dx), _p(dw), _p(db), _p(coef), B, H * W, C, wsp, wsn, _stream())
    return dx, dw, db


def groupnorm_bwd(dy, x, scale, shift, w, mean, rstd, G, act=ACT_NONE, act_param=0.0):
    _chk(dy, x, scale, shift, w, mean, rstd)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    dw, db = new((C,), x), new((C,), x)
    coef = new((3, B, C), x)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    _lib.call("fl_groupnorm_bwd", _p(dy), _p(x), _p(scale), _p(shift), _p(w), _p(mean), _p(rstd), G, act,
              float(act_param), _p(dx), _p(dw), _p(db), _p(coef), B, H * W, C, wsp, wsn, _stream())
    return dx, dw, db


def batchnorm_bwd(dy, x, scale, shift, w, mean, rstd, act=ACT_NONE, act_param=0.0, sync=None, world=1):
    """sync/world as in batchnorm_stats: dx from the global-batch sums, dw/db this rank's share."""
    _chk(dy, x, scale, shift, w, mean, rstd)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    dw, db = new((C,), x), new((C,), x)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    if sync is not None:
        local = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        _lib.call("fl_batchnorm_sums", _p(x), _p(dy), _p(scale), _p(shift), act, float(act_param), _p(local), B, H * W, C,
                  wsp, wsn, _stream())
        glob = local.clone()
        sync(glob)
        coef = new((8, C), x)
        _lib.call("fl_batchnorm_bwd_from_sums", _p(dy), _p(x), _p(scale), _p(shift), _p(w), _p(mean), _p(rstd), act,
                  float(act_param), _p(glob), _p(local), int(world) * B * H * W, _p(dx), _p(dw), _p(db), _p(coef), B, H * W, C,
                  _stream())
        return dx, dw, db
    coef = new((3, C), x)
    _lib.call("fl_batchnorm_bwd", _p(dy), _p(x), _p(scale), _p(shift), _p(w), _p(mean), _p(rstd), act,
              float(act_param), _p(dx), _p(dw), _p(db), _p(coef), B, H * W, C, wsp, wsn, _stream())
    return dx, dw, db


def batchnorm_bwd_pair(dy, x, nb_a, nb_b, w, act=ACT_NONE, act_param=0.0, sync=None, world=1):
    """Backward of batchnorm_stats_pair (+ the activation behind it): dy, x (2B,H,W,C) stacked like the forward, nb_* the halves'
    (scale, shift, mean, rstd). -> dx (2B,H,W,C), dw, db summed over both halves (the two passes' gradients add, v2:1099-1101).
    With `sync` the four sum vectors of both halves share ONE all-reduce."""
    _chk(dy, x, w)
    B2, H, W, C = x.shape
    B = B2 // 2
    dx = torch.empty_like(x)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    dws, dbs = [], []
    if sync is None:
        for sl, nb in ((slice(0, B), nb_a), (slice(B, B2), nb_b)):
            dw, db = new((C,), x), new((C,), x)
            coef = new((3, C), x)
            _lib.call("fl_batchnorm_bwd", _p(dy[sl]), _p(x[sl]), _p(nb[0]), _p(nb[1]), _p(w), _p(nb[2]), _p(nb[3]), act,
                      float(act_param), _p(dx[sl]), _p(dw), _p(db), _p(coef), B, H * W, C, wsp, wsn, _stream())
            dws.append(dw)
            dbs.append(db)
    else:
        local = torch.empty(2, 2 * C, device=x.device, dtype=torch.float64)
        for i, (sl, nb) in enumerate(((slice(0, B), nb_a), (slice(B, B2), nb_b))):
            _lib.call("fl_batchnorm_sums", _p(x[sl]), _p(dy[sl]), _p(nb[0]), _p(nb[1]), act, float(act_param), _p(local[i]), B, H * W, C,
                      wsp, wsn, _stream())
        glob = local.clone()
        sync(glob)
        for i, (sl, nb) in enumerate(((slice(0, B), nb_a), (slice(B, B2), nb_b))):
            dw, db = new((C,), x), new((C,), x)
            coef = new((8, C), x)
            _lib.call("fl_batchnorm_bwd_from_sums", _p(dy[sl]), _p(x[sl]), _p(nb[0]), _p(nb[1]), _p(w), _p(nb[2]), _p(nb[3]), act,
                      float(act_param), _p(glob[i]), _p(local[i]), int(world) * B * H * W, _p(dx[sl]), _p(dw), _p(db), _p(coef), B, H * W, C,
                      _stream())
            dws.append(dw)
            dbs.append(db)
    return dx, axpby(1.0, dws[0], 1.0, dws[1]), axpby(1.0, dbs[0], 1.0, dbs[1])


def layernorm_fwd(x, w, b, mask=None, eps=1e-5, act=ACT_NONE, act_param=0.0, save=True, R=None):
    """Row LayerNorm: y = act(mask*(xhat*w+b)) + R. -> y, mean, rstd."""
    _chk(x, w, b, mask, R)
    rows, D = x.shape
    y = torch.empty_like(x)
    mean = new((rows,), x) if save else None
    rstd = new((rows,), x) if save else None
    _lib.call("fl_layernorm_fwd", _p(x), _p(w), _p(b), _p(mask), _p(R), _p(y), _p(mean), _p(rstd), rows, D, float(eps), act,
              float(act_param), _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, w, b, mask, mean, rstd, act=ACT_NONE, act_param=0.0, R=None, dw_out=None, db_out=None):
    _chk(dy, x, w, b, mask, mean, rstd, R, dw_out, db_out)
    rows, D = x.shape
    dx = torch.empty_like(x)
    dw = dw_out if dw_out is not None else new((D,), x)
    db = db_out if db_out is not None else new((D,), x)
    wsp, wsn = _ws(_lib.load().fl_layernorm_bwd_workspace_bytes(rows, D), x)
    _lib.call("fl_layernorm_bwd", _p(dy), _p(x), _p(w), _p(b), _p(mask), _p(mean), _p(rstd), _p(R), _p(dx), _p(dw), _p(db),
              rows, D, act, float(act_param), wsp, wsn, _stream())
    return dx, dw, db


# --------------------------------------------------------------------------- residual-block tail
def res_tail_map(y, scale=None, shift=None, s=None):
    _chk(y, scale, shift, s)
    B, H, W, C = y.shape
    m = new((B, 2, H, W), y)
    spi = int(s is not None and s.dim() == 2)  # s (B,C): per image; s (C,): one gate vector for the batch
    _lib.call("fl_res_tail_map", _p(y), _p(scale), _p(shift), _p(s), spi, _p(m), B, H * W, C, _stream())
    return m


def res_tail_out(y, scale=None, shift=None, s=None, map_=None, saw=None, xres=None, final_swish=False, save_att=False,
                 xres_pre=None):
    """xres_pre = (scale, shift): the residual is swish(xres * scale[b,c] + shift[b,c]) (its norm pass is fused here)."""
    _chk(y, scale, shift, s, map_, saw, xres)
    B, H, W, C = y.shape
    out = torch.empty_like(y)
    # with a map the library computes the gate in its own small kernel INTO this buffer first, then streams (fl_res_tail_out_pre)
    att = new((B, H, W), y) if (save_att or map_ is not None) else None
    spi = int(s is not None and s.dim() == 2)
    xs, xh = xres_pre if xres_pre is not None else (None, None)
    _chk(xs, xh)
    _lib.call("fl_res_tail_out_pre", _p(y), _p(scale), _p(shift), _p(s), spi, _p(map_), _p(saw), _p(xres), _p(xs), _p(xh), _p(out),
              _p(att), B, H, W, C, int(final_swish), _stream())
    return (out, att) if save_att else out


def attention_fwd(qkv, heads, scale, want_p=False, want_stats=False):
    """qkv (B,H,W,3C) -> merged (B,H,W,C) [channel = i*heads + head], optionally P (B,heads,HW,HW) and / or the row statistics
    (B,heads,HW,2) the fused backward recomputes P from. See fl_attention_fwd."""
    _chk(qkv)
    B, H, W, C3 = qkv.shape
    C, HW = C3 // 3, H * W
    out = new((B, H, W, C), qkv)
    P = new((B, heads, HW, HW), qkv) if want_p else None
    st = new((B, heads, HW, 2), qkv) if want_stats else None
    _lib.call("fl_attention_fwd", _p(qkv), _p(out), _p(P), _p(st), B, HW, C, int(heads), float(scale), _stream())
    res = (out,) + ((P,) if want_p else ()) + ((st,) if want_stats else ())
    return res if len(res) > 1 else out


def attention_bwd(qkv, merged, dmerged, stats, heads, scale):
    """gradient of attention_fwd w.r.t. qkv (same layout) from the gradient of `merged`; stats from attention_fwd(want_stats=True)."""
    _chk(qkv, merged, dmerged, stats)
    B, H, W, C3 = qkv.shape
    C, HW = C3 // 3, H * W
    dqkv = torch.empty_like(qkv)
    n = _lib.load().fl_attention_bwd_workspace_bytes(B, HW, int(heads))
    ws = workspace(n, qkv.device)
    _lib.call("fl_attention_bwd", _p(qkv), _p(merged), _p(dmerged), _p(stats), _p(dqkv), B, HW, C, int(heads), float(scale),
              ctypes.c_void_p(ws.data_ptr()), n, _stream())
    return dqkv


def attention_ok(HW, C, heads):
    return HW % 32 == 0 and C % heads == 0 and C // heads <= 32


# --------------------------------------------------------------------------- elementwise & co
def nchw_to_nhwc(x, Cpad=None):
    _chk(x)
    B, C, H, W = x.shape
    Cpad = Cpad or C
    y = new((B, H, W, Cpad), x)
    _lib.call("fl_nchw_to_nhwc", _p(x), _p(y), B, C, H * W, Cpad, _stream())
    return y


def nhwc_to_nchw(x, C=None):
    _chk(x)
    B, H, W, Cpad = x.shape
    C = C or Cpad
    y = new((B, C, H, W), x)
    _lib.call("fl_nhwc_to_nchw", _p(x), _p(y), B, C, H * W, Cpad, _stream())
    return y


def act_fwd(x, act, p=0.0, out=None):
    _chk(x, out)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("fl_act_fwd", _p(x), _p(out), x.numel(), act, float(p), _stream())
    return out


def act_bwd(dy, x, act, p=0.0, out=None):
    _chk(dy, x, out)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("fl_act_bwd", _p(dy), _p(x), _p(out), x.numel(), act, float(p), _stream())
    return out


def axpby(a, x, b=0.0, y=None, out=None):
    _chk(x, y, out)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("fl_axpby", float(a), _p(x), float(b), _p(y), _p(out), x.numel(), _stream())
    return out


def multi_axpby(pairs, a=1.0, b=0.0):
    """dst = a*src + b*dst for every (src, dst) pair, batched into as few launches as possible."""
    pairs = [(s_, d_) for s_, d_ in pairs if s_ is not None and s_.numel() > 0]
    if not pairs:
        return
    k = len(pairs)
    for s_, d_ in pairs:
        _chk(s_, d_)
        if s_.numel() != d_.numel():
            raise ValueError("multi_axpby: size mismatch %s vs %s" % (tuple(s_.shape), tuple(d_.shape)))
    src = (ctypes.c_void_p * k)(*[s_.data_ptr() for s_, _ in pairs])
    dst = (ctypes.c_void_p * k)(*[d_.data_ptr() for _, d_ in pairs])
    n = (ctypes.c_int64 * k)(*[s_.numel() for s_, _ in pairs])
    _lib.call("fl_multi_axpby", src, dst, n, k, float(a), float(b), _stream())


def multi_colsum(items):
    """items: [(x (M,N), out (N,), alpha)] -> out = alpha * x.sum(0) for every item, in one launch per 64 items."""
    items = [it for it in items if it[0].numel() > 0]
    if not items:
        return
    k = len(items)
    for x, out, _ in items:
        _chk(x, out)
        if x.dim() != 2 or out.numel() != x.shape[1]:
            raise ValueError("multi_colsum: x must be (M,N) and out (N,), got %s / %s" % (tuple(x.shape), tuple(out.shape)))
    xs = (ctypes.c_void_p * k)(*[x.data_ptr() for x, _, _ in items])
    outs = (ctypes.c_void_p * k)(*[o.data_ptr() for _, o, _ in items])
    Ms = (ctypes.c_int * k)(*[x.shape[0] for x, _, _ in items])
    Ns = (ctypes.c_int * k)(*[x.shape[1] for x, _, _ in items])
    al = (ctypes.c_float * k)(*[float(a) for _, _, a in items])
    _lib.call("fl_multi_colsum", xs, outs, Ms, Ns, al, k, _stream())


def scale_heads(v, head_mask, heads, out=None):
    _chk(v, head_mask, out)
    B, D = v.shape
    if out is None:
        out = torch.empty_like(v)
    _lib.call("fl_scale_heads", _p(v), _p(head_mask), _p(out), B, D, heads, _stream())
    return out


def add_rows(x, y, out=None):
    _chk(x, y, out)
    B, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.call("fl_add_rows", _p(x), _p(y), y.shape[0], _p(out), B, D, _stream())
    return out


def time_freqs(n_ch, device):
    """Constant frequency table of v2:411-412, computed on the host exactly as the reference does."""
    import math
    half = n_ch // 2
    return torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(device)


def time_sinusoid(t, freqs):
    _chk(t, freqs)
    assert t.dtype == torch.int64
    n_ch = 2 * freqs.shape[0]
    out = torch.empty((t.shape[0], n_ch), dtype=torch.float32, device=t.device)
    _lib.call("fl_time_sinusoid", _p(t), _p(freqs), _p(out), t.shape[0], n_ch, _stream())
    return out


def embedding_fwd(table, idx):
    _chk(table, idx)
    assert idx.dtype == torch.int64
    out = new((idx.shape[0], table.shape[1]), table)
    _lib.call("fl_embedding_fwd", _p(table), _p(idx), _p(out), idx.shape[0], table.shape[1], _stream())
    return out


def embedding_bwd(dout, idx, rows, out=None):
    _chk(dout, idx, out)
    dt = out if out is not None else new((rows, dout.shape[1]), dout)
    _lib.call("fl_embedding_bwd", _p(dout), _p(idx), _p(dt), idx.shape[0], dout.shape[1], rows, _stream())
    return dt


def q_sample(x0, eps, t, alpha_bar, out=None):
    _chk(x0, eps, t, alpha_bar, out)
    B, D = x0.shape
    if out is None:
        out = torch.empty_like(x0)
    _lib.call("fl_q_sample", _p(x0), _p(eps), _p(t), _p(alpha_bar), _p(out), B, D, _stream())
    return out


def p_sample_update(xt, eps_theta, noise, t_ptr, alpha, alpha_bar, beta, out=None):
    _chk(xt, eps_theta, noise, t_ptr, alpha, alpha_bar, beta, out)
    assert t_ptr.dtype == torch.int32
    if out is None:
        out = torch.empty_like(xt)
    _lib.call("fl_p_sample_update", _p(xt), _p(eps_theta), _p(noise), _p(t_ptr), _p(alpha), _p(alpha_bar), _p(beta),
              _p(out), xt.numel(), _stream())
    return out


def step_counter_add(t32, t64, delta):
    _lib.call("fl_step_counter_add", _p(t32), _p(t64), int(delta), _stream())


def reparameterize(mu, logvar, eps, want_clamped=False):
    _chk(mu, logvar, eps)
    z = torch.empty_like(mu) if eps is not None else None
    lvc = torch.empty_like(mu) if want_clamped else None
    _lib.call("fl_reparameterize", _p(mu), _p(logvar), _p(eps), _p(z), _p(lvc), mu.numel(), _stream())
    return z, lvc


def philox_fill(out, seed, offset=0, dev_offset=None, mode=0, p=0.0):
    _chk(out, dev_offset)
    _lib.call("fl_philox_fill", _p(out), out.numel(), int(seed), int(offset), _p(dev_offset), mode, float(p), _stream())
    return out


def philox_randint(out, seed, hi, offset=0, dev_offset=None):
    _chk(out, dev_offset)
    _lib.call("fl_philox_randint", _p(out), out.numel(), int(seed), int(offset), _p(dev_offset), int(hi), _stream())
    return out


def encoder_heads_ok(B, K, L):
    return bool(_lib.load().fl_encoder_heads_applicable(int(B), int(K), int(L)))


_heads_done = {}


def encoder_heads_fwd(g2, ln_mu, ln_lv, lin_mu, lin_lv, eps=1e-5, want_stats=True, rng_seed=None, rng_state=None):
    """Both encoder heads behind their first Linear (v2:210-221) in one launch: g2 (2, B, K) -> mu, logvar (B, L) and - with
    rng_seed - z = reparameterize(mu, logvar) (v2:339-343) on the Philox draws of (rng_seed, *rng_state); the kernel moves the
    device-resident counter on itself. ln_* = (weight, bias) of the LayerNorms, lin_* = (weight (L, K), bias) of the second Linears."""
    _chk(g2, *ln_mu, *ln_lv, *lin_mu, *lin_lv, rng_state)
    _, B, K = g2.shape
    L = lin_mu[0].shape[0]
    mu = new((B, L), g2) if want_stats else None
    lv = new((B, L), g2) if want_stats else None
    z = new((B, L), g2) if rng_seed is not None else None
    done = None
    if z is not None and rng_state is not None:
        key = (g2.device.type, g2.device.index)
        done = _heads_done.get(key)
        if done is None:
            done = _heads_done[key] = torch.zeros(1, dtype=torch.int32, device=g2.device)
    _lib.call("fl_encoder_heads_fwd", _p(g2), _p(ln_mu[0]), _p(ln_mu[1]), _p(ln_lv[0]), _p(ln_lv[1]), float(eps), _p(lin_mu[0]),
              _p(lin_mu[1]), _p(lin_lv[0]), _p(lin_lv[1]), _p(mu), _p(lv), _p(z), int(rng_seed or 0), _p(rng_state), _p(done), B, K, L,
              _stream())
    return mu, lv, z


def diffuse_prep(z, alpha_bar, n_steps, seed, state, offset=0, masks=None, mask_seed=0, mask_state=None, mask_offset=0, p=0.0):
    """t, noise, x_t of one latent-DDPM training step (v2:1260-1261 + q_sample v2:574-578) and, optionally, the step's dropout
    scales in one launch; the draws are those of philox_randint / philox_fill at the same counters (not advanced here)."""
    _chk(z, alpha_bar, state, masks, mask_state)
    B, L = z.shape
    t = torch.empty((B,), dtype=torch.int64, device=z.device)
    noise, xt = torch.empty_like(z), torch.empty_like(z)
    _lib.call("fl_diffuse_prep", _p(z), _p(alpha_bar), int(n_steps), int(seed), int(offset), _p(state), _p(t), _p(noise), _p(xt), B, L,
              _p(masks), masks.numel() if masks is not None else 0, int(mask_seed), int(mask_offset), _p(mask_state), float(p),
              _stream())
    return t, noise, xt


def advance_offset(dev_offset, delta):
    _lib.call("fl_advance_offset", _p(dev_offset), int(delta), _stream())


# --------------------------------------------------------------------------- losses / optimizer
def euclid_loss_fwd(x, y, reduction=1):
    """-> (loss scalar tensor (1,), per-sample norms (B,))."""
    _chk(x, y)
    B = x.shape[0]
    D = x.numel() // max(B, 1)
    norms = new((B,), x)
    loss = new((1,), x)
    _lib.call("fl_euclid_loss_fwd", _p(x), _p(y), _p(norms), _p(loss), B, D, reduction, _stream())
    return loss, norms


def euclid_loss_bwd(x, y, norms, gscale=1.0, gdev=None, reduction=1, need_dy=False):
    _chk(x, y, norms, gdev)
    B = x.shape[0]
    D = x.numel() // max(B, 1)
    dx = torch.empty_like(x)
    dy = torch.empty_like(x) if need_dy else None
    _lib.call("fl_euclid_loss_bwd", _p(x), _p(y), _p(norms), float(gscale), _p(gdev), _p(dx), _p(dy), B, D, reduction,
              _stream())
    return (dx, dy) if need_dy else dx


def sqnorm(x, out, accumulate=False):
    _chk(x, out)
    wsp, wsn = _ws(_lib.load().fl_sqnorm_workspace_bytes(), x)
    _lib.call("fl_sqnorm", _p(x), x.numel(), _p(out), int(accumulate), wsp, wsn, _stream())
    return out


def clip_coef(sq, max_norm, coef, norm_out=None):
    _lib.call("fl_clip_coef", _p(sq), float(max_norm), _p(coef), _p(norm_out), _stream())
    return coef


def adam_step(p, g, m, v, lr_dev, step_dev, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, coef_dev=None, gscale=1.0,
              decoupled=True, betas_dev=None):
    """betas_dev: device tensor (beta1, beta2) that overrides b1, b2 (read at run time: graph replays follow a scheduler)."""
    _chk(p, g, m, v, lr_dev, step_dev, coef_dev, betas_dev)
    _lib.call("fl_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_dev), _p(step_dev), float(b1), float(b2),
              float(eps), float(wd), _p(coef_dev), float(gscale), int(decoupled), _p(betas_dev), _stream())


def kl_fwd(mu, logvar, want_musq=False):
    """-> (loss, kl_row[, musq_row]): loss = mean(clamp(kl_row, 0, 100)) + 1e-4 * sum(musq_row)  (v2:380-386)."""
    _chk(mu, logvar)
    B, D = mu.shape
    kl_row, musq = new((B,), mu), new((B,), mu)
    loss = new((1,), mu)
    _lib.call("fl_kl_fwd", _p(mu), _p(logvar), _p(kl_row), _p(musq), _p(loss), B, D, _stream())
    return (loss, kl_row, musq) if want_musq else (loss, kl_row)


def kl_bwd(mu, logvar, kl_row, gscale=1.0, gdev=None, reg_scale=1.0):
    _chk(mu, logvar, kl_row, gdev)
    B, D = mu.shape
    dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
    _lib.call("fl_kl_bwd", _p(mu), _p(logvar), _p(kl_row), float(gscale), _p(gdev), _p(dmu), _p(dlv), B, D,
              float(reg_scale), _stream())
    return dmu, dlv


def bce_fwd(p, target):
    _chk(p)
    loss = new((1,), p)
    _lib.call("fl_bce_fwd", _p(p), float(target), _p(loss), p.numel(), _stream())
    return loss


def bce_bwd(p, target, gscale=1.0, gdev=None):
    _chk(p, gdev)
    dp = torch.empty_like(p)
    _lib.call("fl_bce_bwd", _p(p), float(target), float(gscale), _p(gdev), _p(dp), p.numel(), _stream())
    return dp


def cross_entropy_fwd(logits, labels):
    _chk(logits, labels)
    B, C = logits.shape
    row, probs, loss = new((B,), logits), torch.empty_like(logits), new((1,), logits)
    _lib.call("fl_cross_entropy_fwd", _p(logits), _p(labels), _p(row), _p(probs), _p(loss), B, C, _stream())
    return loss, probs


def cross_entropy_bwd(probs, labels, gscale=1.0, gdev=None):
    _chk(probs, labels, gdev)
    B, C = probs.shape
    d = torch.empty_like(probs)
    _lib.call("fl_cross_entropy_bwd", _p(probs), _p(labels), float(gscale), _p(gdev), _p(d), B, C, _stream())
    return d


def softmax_rows(x, scale=1.0):
    _chk(x)
    rows, D = x.shape
    y = torch.empty_like(x)
    _lib.call("fl_softmax_rows", _p(x), _p(y), rows, D, float(scale), _stream())
    return y


def softmax_rows_bwd(y, dy, scale=1.0):
    _chk(y, dy)
    rows, D = y.shape
    dx = torch.empty_like(y)
    _lib.call("fl_softmax_rows_bwd", _p(y), _p(dy), _p(dx), rows, D, float(scale), _stream())
    return dx


def gather_rows(table, idx):
    _chk(table, idx)
    out = new((idx.shape[0], table.shape[1]), table)
    _lib.call("fl_gather_rows", _p(table), _p(idx), _p(out), idx.shape[0], table.shape[1], _stream())
    return out


def augment_batch(cache, index=None, params=None, labels=None, out=None):
    """transform_train / transform_test (v2:24-35) on a uint8 NHWC image cache resident in HBM.
    cache (N,64,64,3) uint8; index (B,) int64 or None; params (B,16) int32 or None (= ToTensor only);
    labels (N,) int64 or None -> data (B,3,64,64) float32 [, labels (B,) int64]."""
    for t, dt in ((cache, torch.uint8), (index, torch.int64), (params, torch.int32), (labels, torch.int64)):
        if t is None:
            continue
        if not t.is_cuda or not t.is_contiguous() or t.dtype != dt:
            raise RuntimeError("augment_batch: need contiguous device tensors (uint8 cache, int64 index/labels, int32 params)")
    N, H, W, C = cache.shape
    if C != 3:
        raise ValueError("augment_batch: the cache is NHWC RGB")
    B = index.shape[0] if index is not None else (params.shape[0] if params is not None else N)
    if params is not None and tuple(params.shape) != (B, 16):
        raise ValueError("augment_batch: params must be (B,16) int32, got %s for B=%d" % (tuple(params.shape), B))
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=cache.device)
    lab_out = torch.empty((B,), dtype=torch.int64, device=cache.device) if labels is not None else None
    _lib.call("fl_augment_batch", _p(cache), _p(labels), _p(index), _p(params), _p(out), _p(lab_out), B, N, H, W, _stream())
    return (out, lab_out) if labels is not None else out


def update_centers(centers, z, labels, momentum=0.9):
    _chk(centers, z, labels)
    _lib.call("fl_update_centers", _p(centers), _p(z), _p(labels), float(momentum), z.shape[0], z.shape[1],
              centers.shape[0], _stream())


def res_tail_bwd_a(dout, y, scale, shift, s, att, xres, final_swish=True, want_da=True):
    """-> dw (B,H,W,C), da (B,H,W) or None."""
    _chk(dout, y, scale, shift, s, att, xres)
    B, H, W, C = y.shape
    dw = torch.empty_like(y)
    da = new((B, H, W), y) if (want_da and att is not None) else None
    _lib.call("fl_res_tail_bwd_a", _p(dout), _p(y), _p(scale), _p(shift), _p(s), int(s is not None and s.dim() == 2),
              _p(att), _p(xres), int(final_swish), _p(dw), _p(da), B, H * W, C, _stream())
    return dw, da


def res_tail_bwd_c(dw, y, scale, shift, s, att, dmap):
    _chk(dw, y, scale, shift, s, att, dmap)
    B, H, W, C = y.shape
    dn = torch.empty_like(y)
    _lib.call("fl_res_tail_bwd_c", _p(dw), _p(y), _p(scale), _p(shift), _p(s), int(s is not None and s.dim() == 2),
              _p(att), _p(dmap), _p(dn), B, H * W, C, _stream())
    return dn


def sa_conv_bwd(da, map_, saw, want_dsaw=True):
    _chk(da, map_, saw)
    B, H, W = da.shape
    dmap = new((B, 2, H, W), da)
    dsaw = new((1, 2, 7, 7), da) if want_dsaw else None
    _lib.call("fl_sa_conv_bwd", _p(da), _p(map_), _p(saw), _p(dmap), _p(dsaw), B, H, W, _stream())
    return dmap, dsaw


def ca_gate_grad(lnw, lnb, dlnw, dlnb, s):
    _chk(lnw, lnb, dlnw, dlnb, s)
    ds = torch.empty_like(s)
    _lib.call("fl_ca_gate_grad", _p(lnw), _p(lnb), _p(dlnw), _p(dlnb), _p(s), _p(ds), s.numel(), _stream())
    return ds


def chan_dot(dy, x):
    """(B,C): sum over pixels of dy*x (NHWC)."""
    _chk(dy, x)
    B, H, W, C = x.shape
    out = new((B, C), x)
    wsp, wsn = _norm_ws(B, H * W, C, x)
    _lib.call("fl_chan_sums", _p(x), _p(dy), None, _p(out), B, H * W, C, wsp, wsn, _stream())
    return out


def chan_sum(dy):
    """(B,C): sum over pixels of dy (NHWC) - gradient of a per-(image, channel) bias broadcast over the map."""
    _chk(dy)
    B, H, W, C = dy.shape
    out = new((B, C), dy)
    wsp, wsn = _norm_ws(B, H * W, C, dy)
    _lib.call("fl_chan_sums", _p(dy), None, _p(out), None, B, H * W, C, wsp, wsn, _stream())
    return out


def maxpool2_fwd(x):
    _chk(x)
    B, H, W, C = x.shape
    y = new((B, H // 2, W // 2, C), x)
    _lib.call("fl_maxpool2_fwd", _p(x), _p(y), B, H, W, C, _stream())
    return y


def maxpool2_bwd(dy, x):
    _chk(dy, x)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    _lib.call("fl_maxpool2_bwd", _p(dy), _p(x), _p(dx), B, H, W, C, _stream())
    return dx


def mul(a, b, out=None):
    _chk(a, b, out)
    if out is None:
        out = torch.empty_like(a)
    _lib.call("fl_mul", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def clamp_bwd(d, x, lo, hi):
    _chk(d, x)
    out = torch.empty_like(x)
    _lib.call("fl_clamp_bwd", _p(d), _p(x), float(lo), float(hi), _p(out), x.numel(), _stream())
    return out


def reparam_bwd(dz, logvar, eps):
    _chk(dz, logvar, eps)
    out = torch.empty_like(logvar)
    _lib.call("fl_reparam_bwd", _p(dz), _p(logvar), _p(eps), _p(out), logvar.numel(), _stream())
    return out


def balance_losses(terms, weights):
    _chk(terms, weights)
    coeff, total = new((6,), terms), new((1,), terms)
    _lib.call("fl_balance_losses", _p(terms), _p(weights), _p(coeff), _p(total), _stream())
    return coeff, total
