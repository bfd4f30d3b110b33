// Backward of the residual-block tail (v2:173-177) + small kernels used only by the VAE-GAN training step:
// 2x2 max-pool (VGG16, v2:965), clamp / reparameterize gradients, elementwise product.
//
// Forward recap:  n = y*scale + shift ; u = s*n ; map = [mean_c u, max_c u] ; att = sigmoid(conv7x7(map)) ;
//                 w = u*att + x ; out = swish(w)
// Backward passes (all HBM-streaming, NHWC, a pixel owned by LPP = min(64, C/4) lanes):
//   A: dw = dout*swish'(w)  (written: it is also the residual-branch gradient) ; da = att(1-att) * sum_c dw*u
//   B: dmap = conv7x7^T(da) ; dsaw = sum da (x) map          (tiny, 2 planes / 98 taps)
//   C: du = dw*att + dmap0/C + [c = argmax_c u] dmap1 ; dn = s*du   -> fl_instnorm_bwd consumes dn
// The gate gradient needs no extra pass: sum du*n = (w_ln*dw_ln + b_ln*db_ln)/s per channel (ca_gate_grad).
#include "fl_common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPP>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float4 tail_u(const float* y, const float* scale, const float* shift, const float* s,
                                         long long pix_off, long long bc_off, long long s_off) {
  float4 v = ld4(y + pix_off);
  if (scale) {
    const float4 sc = ld4(scale + bc_off), sh = ld4(shift + bc_off);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  }
  if (s) {
    const float4 ss = ld4(s + s_off);
    v.x *= ss.x; v.y *= ss.y; v.z *= ss.z; v.w *= ss.w;
  }
  return v;
}

struct TailArgs {
  const float *y, *scale, *shift, *s, *att, *xres;
  int s_per_image, B, HW, C, final_swish;
};

#define PIXEL_LOOP_HEAD                                                                         \
  constexpr int PPW = 64 / LPP;                                                                  \
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;                                     \
  const int gl = lane % LPP, gp = lane / LPP;                                                    \
  const long long npix = (long long)a.B * a.HW;                                                  \
  const int C4 = a.C >> 2;                                                                       \
  const long long wave0 = ((long long)blockIdx.x * (blockDim.x >> 6) + wid) * PPW;               \
  const long long wstride = (long long)gridDim.x * (blockDim.x >> 6) * PPW;                      \
  for (long long pw = wave0; pw < npix; pw += wstride) {                                         \
    const long long pix = pw + gp;                                                               \
    const bool ok = pix < npix;                                                                  \
    const long long pp = ok ? pix : 0;                                                           \
    const int b = (int)(pp / a.HW);

template <int LPP>
__global__ void tail_bwd_a_kernel(const TailArgs a, const float* __restrict__ dout, float* __restrict__ dw,
                                  float* __restrict__ da) {
  PIXEL_LOOP_HEAD
    const float att = a.att ? a.att[pp] : 1.f;
    float acc = 0.f;
    for (int c4 = gl; c4 < C4; c4 += LPP) {
      const long long off = pp * a.C + c4 * 4;
      const float4 u = tail_u(a.y, a.scale, a.shift, a.s, off, (long long)b * a.C + c4 * 4,
                              (a.s_per_image ? (long long)b * a.C : 0) + c4 * 4);
      float4 g = ld4(dout + off);
      if (a.final_swish) {
        float4 w = make_float4(u.x * att, u.y * att, u.z * att, u.w * att);
        if (a.xres) {
          const float4 r = ld4(a.xres + off);
          w.x += r.x; w.y += r.y; w.z += r.z; w.w += r.w;
        }
        g.x *= fl::swish_grad(w.x); g.y *= fl::swish_grad(w.y); g.z *= fl::swish_grad(w.z); g.w *= fl::swish_grad(w.w);
      }
      acc += (g.x * u.x + g.y * u.y) + (g.z * u.z + g.w * u.w);
      if (ok && dw) st4(dw + off, g);
    }
    acc = group_sum<LPP>(acc);
    if (ok && gl == 0 && da) da[pp] = acc * att * (1.f - att);
  }
}

template <int LPP>
__global__ void tail_bwd_c_kernel(const TailArgs a, const float* __restrict__ dw, const float* __restrict__ dmap,
                                  float* __restrict__ dn) {
  PIXEL_LOOP_HEAD
    const float att = a.att ? a.att[pp] : 1.f;
    const int p = (int)(pp - (long long)b * a.HW);
    float d0 = 0.f, d1 = 0.f, mx = -INFINITY;
    if (dmap) {
      d0 = dmap[((long long)b * 2) * a.HW + p] / a.C;
      d1 = dmap[((long long)b * 2 + 1) * a.HW + p];
      for (int c4 = gl; c4 < C4; c4 += LPP) {
        const float4 u = tail_u(a.y, a.scale, a.shift, a.s, pp * a.C + c4 * 4, (long long)b * a.C + c4 * 4,
                                (a.s_per_image ? (long long)b * a.C : 0) + c4 * 4);
        mx = fmaxf(fmaxf(mx, fmaxf(u.x, u.y)), fmaxf(u.z, u.w));
      }
      mx = group_max<LPP>(mx);
    }
    if (!ok) continue;
    for (int c4 = gl; c4 < C4; c4 += LPP) {
      const long long off = pp * a.C + c4 * 4;
      const long long s_off = (a.s_per_image ? (long long)b * a.C : 0) + c4 * 4;
      const float4 g = ld4(dw + off);
      float4 du = make_float4(g.x * att + d0, g.y * att + d0, g.z * att + d0, g.w * att + d0);
      if (dmap) {
        const float4 u = tail_u(a.y, a.scale, a.shift, a.s, off, (long long)b * a.C + c4 * 4, s_off);
        if (u.x == mx) du.x += d1;
        if (u.y == mx) du.y += d1;
        if (u.z == mx) du.z += d1;
        if (u.w == mx) du.w += d1;
      }
      if (a.s) {
        const float4 ss = ld4(a.s + s_off);
        du.x *= ss.x; du.y *= ss.y; du.z *= ss.z; du.w *= ss.w;
      }
      st4(dn + off, du);
    }
  }
}

// dmap[b][ch][y][x] = sum_{ky,kx} saw[ch][ky][kx] * da[b][y-(ky-3)][x-(kx-3)]
__global__ void sa_dmap_kernel(const float* __restrict__ da, const float* __restrict__ saw, float* __restrict__ dmap,
                               int B, int H, int W) {
  __shared__ float wsm[98];
  for (int i = threadIdx.x; i < 98; i += blockDim.x) wsm[i] = saw[i];
  __syncthreads();
  const long long total = (long long)B * 2 * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int ch = (int)(r % 2), b = (int)(r / 2);
    float acc = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
      const int yy = y - (ky - 3);
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int kx = 0; kx < 7; ++kx) {
        const int xx = x - (kx - 3);
        if ((unsigned)xx < (unsigned)W) acc += wsm[ch * 49 + ky * 7 + kx] * da[((long long)b * H + yy) * W + xx];
      }
    }
    dmap[i] = acc;
  }
}
// dsaw[t] = sum_{b,y,x} da[b][y][x] * map[b][ch][y+ky-3][x+kx-3]; one block per tap t = ch*49+ky*7+kx
__global__ void sa_dsaw_kernel(const float* __restrict__ da, const float* __restrict__ map, float* __restrict__ dsaw,
                               int B, int H, int W) {
  __shared__ float red[16];
  const int t = blockIdx.x, ch = t / 49, r = t % 49, dy = r / 7 - 3, dx = r % 7 - 3;
  const long long n = (long long)B * H * W;
  float acc = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const int x = (int)(i % W);
    const long long q = i / W;
    const int y = (int)(q % H), b = (int)(q / H);
    const int yy = y + dy, xx = x + dx;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
      acc += da[i] * map[(((long long)b * 2 + ch) * H + yy) * W + xx];
  }
  acc = fl::block_sum(acc, red);
  if (threadIdx.x == 0) dsaw[t] = acc;
}

__global__ void ca_gate_grad_kernel(const float* lnw, const float* lnb, const float* dlnw, const float* dlnb,
                                    const float* s, float* ds, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) ds[c] = (lnw[c] * dlnw[c] + lnb[c] * dlnb[c]) / s[c];
}

// ---- 2x2 max pool, NHWC ---------------------------------------------------------------------
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)B * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho), b = (int)(r / Ho);
    const float* p = x + ((((long long)b * H + 2 * oy) * W + 2 * ox) * C4 + c4) * 4;
    const float4 a = ld4(p), bq = ld4(p + C4 * 4), c = ld4(p + (long long)W * C4 * 4), d = ld4(p + ((long long)W + 1) * C4 * 4);
    float4 m;
    m.x = fmaxf(fmaxf(a.x, bq.x), fmaxf(c.x, d.x));
    m.y = fmaxf(fmaxf(a.y, bq.y), fmaxf(c.y, d.y));
    m.z = fmaxf(fmaxf(a.z, bq.z), fmaxf(c.z, d.z));
    m.w = fmaxf(fmaxf(a.w, bq.w), fmaxf(c.w, d.w));
    st4(y + i * 4, m);
  }
}
// dx: gradient goes to the FIRST maximum of each window in scan order (torch max_pool2d semantics)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                    int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)B * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho), b = (int)(r / Ho);
    const long long base = (((long long)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const long long o[4] = {base, base + C, base + (long long)W * C, base + ((long long)W + 1) * C};
    int best = 0;
    float mv = x[o[0]];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float v = x[o[k]];
      if (v > mv) { mv = v; best = k; }
    }
    const float g = dy[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) dx[o[k]] = (k == best) ? g : 0.f;
  }
}

// ---- tiny elementwise -------------------------------------------------------------------------
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = a[i] * b[i];
}
__global__ void clamp_bwd_kernel(const float* __restrict__ d, const float* __restrict__ x, float lo, float hi,
                                 float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = (x[i] >= lo && x[i] <= hi) ? d[i] : 0.f;
}
// z = mu + eps*exp(0.5*clamp(lv,-2,10)): dlv = dz*eps*0.5*exp(0.5*lvc) inside the clamp window
__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ logvar,
                                   const float* __restrict__ eps, float* __restrict__ dlv, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float lv = logvar[i];
    dlv[i] = (lv >= -2.0f && lv <= 10.0f) ? dz[i] * eps[i] * 0.5f * expf(0.5f * lv) : 0.f;
  }
}

static int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
static int tail_blocks(long long npix, int ppw) {
  long long waves = (npix + ppw - 1) / ppw;
  long long blocks = (waves + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define DISPATCH_LPP(KERNEL, ...)                                                                                  \
  do {                                                                                                             \
    const int C4_ = C / 4;                                                                                         \
    if (C4_ >= 64) hipLaunchKernelGGL(KERNEL<64>, dim3(tail_blocks(npix, 1)), dim3(256), 0, st, __VA_ARGS__);      \
    else if (C4_ >= 32) hipLaunchKernelGGL(KERNEL<32>, dim3(tail_blocks(npix, 2)), dim3(256), 0, st, __VA_ARGS__); \
    else if (C4_ >= 16) hipLaunchKernelGGL(KERNEL<16>, dim3(tail_blocks(npix, 4)), dim3(256), 0, st, __VA_ARGS__); \
    else if (C4_ >= 8) hipLaunchKernelGGL(KERNEL<8>, dim3(tail_blocks(npix, 8)), dim3(256), 0, st, __VA_ARGS__);   \
    else hipLaunchKernelGGL(KERNEL<1>, dim3(tail_blocks(npix, 64)), dim3(256), 0, st, __VA_ARGS__);                \
  } while (0)

extern "C" {

int fl_res_tail_bwd_a(const float* dout, const float* y, const float* scale, const float* shift, const float* s,
                      int s_per_image, const float* att, const float* xres, int final_swish, float* dw, float* da,
                      int B, int HW, int C, fl_stream_t stream) {
  FL_REQUIRE(dout && y && C % 4 == 0, "fl_res_tail_bwd_a: bad args");
  hipStream_t st = ST(stream);
  const long long npix = (long long)B * HW;
  if (!npix) return FL_OK;
  TailArgs a = {y, scale, shift, s, att, xres, s_per_image, B, HW, C, final_swish};
  DISPATCH_LPP(tail_bwd_a_kernel, a, dout, dw, da);
  return fl::check_launch("tail_bwd_a");
}

int fl_res_tail_bwd_c(const float* dw, const float* y, const float* scale, const float* shift, const float* s,
                      int s_per_image, const float* att, const float* dmap, float* dn, int B, int HW, int C,
                      fl_stream_t stream) {
  FL_REQUIRE(dw && y && dn && C % 4 == 0, "fl_res_tail_bwd_c: bad args");
  hipStream_t st = ST(stream);
  const long long npix = (long long)B * HW;
  if (!npix) return FL_OK;
  TailArgs a = {y, scale, shift, s, att, nullptr, s_per_image, B, HW, C, 0};
  DISPATCH_LPP(tail_bwd_c_kernel, a, dw, dmap, dn);
  return fl::check_launch("tail_bwd_c");
}

int fl_sa_conv_bwd(const float* da, const float* map, const float* saw, float* dmap, float* dsaw, int B, int H, int W,
                   fl_stream_t stream) {
  FL_REQUIRE(da && saw && dmap, "fl_sa_conv_bwd: bad args");
  const long long total = (long long)B * 2 * H * W;
  if (!total) return FL_OK;
  hipLaunchKernelGGL(sa_dmap_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), da, saw, dmap, B, H, W);
  if (dsaw) {
    FL_REQUIRE(map, "fl_sa_conv_bwd: dsaw needs the forward map");
    hipLaunchKernelGGL(sa_dsaw_kernel, dim3(98), dim3(1024), 0, ST(stream), da, map, dsaw, B, H, W);
  }
  return fl::check_launch("sa_conv_bwd");
}

int fl_ca_gate_grad(const float* lnw, const float* lnb, const float* dlnw, const float* dlnb, const float* s, float* ds,
                    int C, fl_stream_t stream) {
  hipLaunchKernelGGL(ca_gate_grad_kernel, dim3(fl::cdiv(C, 128)), dim3(128), 0, ST(stream), lnw, lnb, dlnw, dlnb, s, ds, C);
  return fl::check_launch("ca_gate_grad");
}

int fl_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, fl_stream_t stream) {
  FL_REQUIRE(x && y && C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "fl_maxpool2_fwd: bad args");
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
  if (!total) return FL_OK;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), x, y, B, H, W, C / 4);
  return fl::check_launch("maxpool2_fwd");
}
int fl_maxpool2_bwd(const float* dy, const float* x, float* dx, int B, int H, int W, int C, fl_stream_t stream) {
  FL_REQUIRE(dy && x && dx && H % 2 == 0 && W % 2 == 0, "fl_maxpool2_bwd: bad args");
  const long long total = (long long)B * (H / 2) * (W / 2) * C;
  if (!total) return FL_OK;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), dy, x, dx, B, H, W, C);
  return fl::check_launch("maxpool2_bwd");
}
int fl_mul(const float* a, const float* b, float* out, int64_t n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), a, b, out, (long long)n);
  return fl::check_launch("mul");
}
int fl_clamp_bwd(const float* d, const float* x, float lo, float hi, float* out, int64_t n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(clamp_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), d, x, lo, hi, out, (long long)n);
  return fl::check_launch("clamp_bwd");
}
int fl_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, int64_t n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), dz, logvar, eps, dlogvar, (long long)n);
  return fl::check_launch("reparam_bwd");
}

}  // extern "C"
