// Normalisation family over NHWC fp32 activations (HBM-bound kernels).
//
// All three spatial norms of the v2 surface reduce to per-(image, channel) sums:
//   LayerNorm2d (v2:144-156) = instance norm: stats over (H,W) per (b,c)
//   nn.GroupNorm (v2:257,263,269,274): stats over (H,W,C/G) per (b,g)
//   nn.BatchNorm2d (v2:991-999): stats over (B,H,W) per c (+ running stats)
// so they share ONE streaming pass (chan_stats_partial: float4 loads, 64-lane
// coalesced rows, deterministic two-stage reduction, no atomics) plus tiny
// finalizers that emit per-(b,c) scale/shift; the normalise+affine+activation is
// a single fused elementwise pass (affine_act) that later rounds can fold into
// the consuming conv's A-tile loader. Backward has the same shape: one pass for
// (sum g, sum g*x), finalizers produce per-(b,c) coefficients (A,Bc,Cc) and
// dx = A*g + Bc*x + Cc is a second fused pass.
#include "fl_common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float act_grad(float u, int act, float p) {
  switch (act) {
    case FL_ACT_SWISH: return fl::swish_grad(u);
    case FL_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case FL_ACT_LRELU: return u > 0.f ? 1.f : p;
    case FL_ACT_SIGMOID: { float s = fl::sigmoid_acc(u); return s * (1.f - s); }
    default: return 1.f;
  }
}

// ---------------------------------------------------------------------------
// partial[((b*S + s)*2 + j)*C + c], j=0: sum a, j=1: sum a*x' where
//   MODE 0: a = x, x' = x            (sum x, sum x^2)
//   MODE 1: a = dy*act'(x*scale+shift), x' = x   (sum g, sum g*x)
// grid = (col tiles of 64 float4, S, B), block = 256
template <int MODE>
__global__ void chan_stats_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   int sb_stride, int act, float act_param, float* __restrict__ partial,
                                   int HW, int C, int S, int pix_per_slab) {
  __shared__ float4 red[2][256];
  const int C4 = C >> 2;
  const int cols = C4 < 64 ? C4 : 64;
  const int rows = 256 / cols;
  const int tid = threadIdx.x;
  const int col = tid % cols, row = tid / cols;
  const int c4 = blockIdx.x * 64 + col;
  const int s = blockIdx.y, b = blockIdx.z;
  const bool active = row < rows && c4 < C4;
  float4 a1 = make_float4(0, 0, 0, 0), a2 = make_float4(0, 0, 0, 0);
  if (active) {
    const int p0 = s * pix_per_slab;
    int p1 = p0 + pix_per_slab;
    if (p1 > HW) p1 = HW;
    float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
    if (MODE == 1 && scale) {
      sc = ld4(scale + (long long)b * sb_stride + c4 * 4);
      sh = ld4(shift + (long long)b * sb_stride + c4 * 4);
    }
    const float* xp = x + ((long long)b * HW) * C + c4 * 4;
    const float* dp = MODE == 1 ? dy + ((long long)b * HW) * C + c4 * 4 : nullptr;
    for (int p = p0 + row; p < p1; p += rows) {
      float4 v = ld4(xp + (long long)p * C);
      float4 a = v;
      if (MODE == 1) {
        float4 g = ld4(dp + (long long)p * C);
        a.x = g.x * act_grad(v.x * sc.x + sh.x, act, act_param);
        a.y = g.y * act_grad(v.y * sc.y + sh.y, act, act_param);
        a.z = g.z * act_grad(v.z * sc.z + sh.z, act, act_param);
        a.w = g.w * act_grad(v.w * sc.w + sh.w, act, act_param);
      }
      a1.x += a.x; a1.y += a.y; a1.z += a.z; a1.w += a.w;
      a2.x += a.x * v.x; a2.y += a.y * v.y; a2.z += a.z * v.z; a2.w += a.w * v.w;
    }
  }
  red[0][tid] = a1;
  red[1][tid] = a2;
  __syncthreads();
  if (row == 0 && c4 < C4 && tid < cols) {
    for (int r = 1; r < rows; ++r) {
      float4 t1 = red[0][r * cols + col], t2 = red[1][r * cols + col];
      a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
      a2.x += t2.x; a2.y += t2.y; a2.z += t2.z; a2.w += t2.w;
    }
    float* out = partial + ((long long)(b * S + s) * 2) * C + c4 * 4;
    st4(out, a1);
    st4(out + C, a2);
  }
}

struct Slabs { int S, pix; };
static Slabs pick_slabs(int B, int HW, int C) {
  const int C4 = C / 4;
  const int cols = C4 < 64 ? C4 : 64;
  const int rows = 256 / (cols > 0 ? cols : 1);
  const int coltiles = fl::cdiv(C4, 64);
  long long blocks = (long long)B * coltiles;
  int S = (int)(2048 / (blocks > 0 ? blocks : 1));
  int maxS = HW / (rows * 4);
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  Slabs r;
  r.pix = fl::cdiv(HW, S);
  r.S = fl::cdiv(HW, r.pix);
  return r;
}

// reduce S partial slabs -> sums[b][2][C] in place at slab 0 semantics (returns via out arrays)
template <class T>
__device__ __forceinline__ void sum_slabs(const T* partial, int b, int S, int C, int c, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
  for (int s = 0; s < S; ++s) {
    const T* p = partial + ((long long)(b * S + s) * 2) * C + c;
    s1 += (double)p[0];
    s2 += (double)p[C];
  }
}

// out[g][j] = sum_{r<n} partial[(g*n + r)*W + j], j < W (= 2C: sums then sums of squares), in double. The batch-norm
// and group-norm finalizers used to walk B*S (up to 4096) slabs per channel in one thread: 74 us for a 2 KB result.
// 1024 threads = 64 columns x 16 row lanes, tree-reduced through LDS.
__global__ __launch_bounds__(1024) void collapse_rows(const float* __restrict__ partial, double* __restrict__ out, int n, int W) {
  __shared__ double red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  const float* p = partial + (long long)blockIdx.y * n * W;
  double a = 0.0;
  if (j < W)
    for (int r = rl; r < n; r += 16) a += (double)p[(long long)r * W + j];
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && j < W) {
#pragma unroll
    for (int k = 1; k < 16; ++k) a += red[k][cl];
    out[(long long)blockIdx.y * W + j] = a;
  }
}

// ---- instance norm ---------------------------------------------------------
__global__ void inorm_finalize_fwd(const float* partial, const float* w, const float* bias, float eps, float* scale,
                                   float* shift, float* mean, float* rstd, int B, int HW, int C, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  double s1, s2;
  sum_slabs(partial, b, S, C, c, s1, s2);
  const double m = s1 / HW;
  double var = s2 / HW - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = r * (w ? w[c] : 1.f);
  scale[i] = sc;
  shift[i] = (bias ? bias[c] : 0.f) - (float)m * sc;
  if (mean) mean[i] = (float)m;
  if (rstd) rstd[i] = r;
}

// coefficients for dx = A*g + Bc*x + Cc ; also per-(b,c) dw/db contributions
__global__ void inorm_finalize_bwd(const float* partial, const float* w, const float* mean, const float* rstd,
                                   float* A, float* Bc, float* Cc, float* dwb /*[B][2][C]*/, int B, int HW, int C, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  double s1, sx;
  sum_slabs(partial, b, S, C, c, s1, sx);
  const double m = mean[i], r = rstd[i], wc = w ? w[c] : 1.f;
  const double s2 = r * (sx - m * s1);  // sum g*xhat
  const double n = HW;
  A[i] = (float)(r * wc);
  Bc[i] = (float)(-r * r * wc * s2 / n);
  Cc[i] = (float)(-r * wc * s1 / n + r * r * wc * m * s2 / n);
  dwb[((long long)b * 2) * C + c] = (float)s2;
  dwb[((long long)b * 2 + 1) * C + c] = (float)s1;
}

// ---- group norm -------------------------------------------------------------
template <class T>
__global__ void gnorm_finalize_fwd(const T* partial, const float* w, const float* bias, float eps, float* scale,
                                   float* shift, float* mean, float* rstd, int B, int HW, int C, int G, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, gi = i - b * G, cg = C / G;
  double t1 = 0, t2 = 0;
  for (int c = gi * cg; c < (gi + 1) * cg; ++c) {
    double s1, s2;
    sum_slabs(partial, b, S, C, c, s1, s2);
    t1 += s1; t2 += s2;
  }
  const double n = (double)HW * cg, m = t1 / n;
  double var = t2 / n - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  if (mean) mean[i] = (float)m;
  if (rstd) rstd[i] = r;
  for (int c = gi * cg; c < (gi + 1) * cg; ++c) {
    const float sc = r * w[c];
    scale[(long long)b * C + c] = sc;
    shift[(long long)b * C + c] = bias[c] - (float)m * sc;
  }
}

template <class T>
__global__ void gnorm_finalize_bwd(const T* partial, const float* w, const float* mean, const float* rstd,
                                   float* A, float* Bc, float* Cc, float* dwb, int B, int HW, int C, int G, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, gi = i - b * G, cg = C / G;
  const double m = mean[i], r = rstd[i];
  double T1 = 0, T2 = 0;
  for (int c = gi * cg; c < (gi + 1) * cg; ++c) {
    double s1, sx;
    sum_slabs(partial, b, S, C, c, s1, sx);
    const double s2 = r * (sx - m * s1);
    T1 += (double)w[c] * s1;
    T2 += (double)w[c] * s2;
    dwb[((long long)b * 2) * C + c] = (float)s2;
    dwb[((long long)b * 2 + 1) * C + c] = (float)s1;
  }
  const double n = (double)HW * cg;
  for (int c = gi * cg; c < (gi + 1) * cg; ++c) {
    const long long o = (long long)b * C + c;
    A[o] = (float)(r * w[c]);
    Bc[o] = (float)(-r * r * T2 / n);
    Cc[o] = (float)(-r * T1 / n + r * r * m * T2 / n);
  }
}

// ---- batch norm (training statistics; running stats like nn.BatchNorm2d) -------
template <class T>
__global__ void bnorm_finalize_fwd(const T* partial, const float* w, const float* bias, float eps, float momentum,
                                   float* running_mean, float* running_var, int training, float* scale, float* shift,
                                   float* mean, float* rstd, int B, int HW, int C, int S, int nb /* slab groups: B, or 1 if collapsed */) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double m, var;
  if (training) {
    double t1 = 0, t2 = 0;
    for (int b = 0; b < nb; ++b) {
      double s1, s2;
      sum_slabs(partial, b, S, C, c, s1, s2);
      t1 += s1; t2 += s2;
    }
    const double n = (double)B * HW;
    m = t1 / n;
    var = t2 / n - m * m;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      const double unb = n > 1 ? var * n / (n - 1) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  } else {
    m = running_mean[c];
    var = running_var[c];
  }
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = r * w[c];
  scale[c] = sc;
  shift[c] = bias[c] - (float)m * sc;
  if (mean) mean[c] = (float)m;
  if (rstd) rstd[c] = r;
}

template <class T>
__global__ void bnorm_finalize_bwd(const T* partial, const float* w, const float* mean, const float* rstd,
                                   float* A, float* Bc, float* Cc, float* dw, float* db, int B, int HW, int C, int S, int nb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t1 = 0, tx = 0;
  for (int b = 0; b < nb; ++b) {
    double s1, sx;
    sum_slabs(partial, b, S, C, c, s1, sx);
    t1 += s1; tx += sx;
  }
  const double m = mean[c], r = rstd[c], wc = w[c], n = (double)B * HW;
  const double s2 = r * (tx - m * t1);
  A[c] = (float)(r * wc);
  Bc[c] = (float)(-r * r * wc * s2 / n);
  Cc[c] = (float)(-r * wc * t1 / n + r * r * wc * m * s2 / n);
  dw[c] = (float)s2;
  db[c] = (float)t1;
}

// dw[c] = sum_b dwb[b][0][c], db[c] = sum_b dwb[b][1][c]
__global__ void dwb_reduce(const float* dwb, float* dw, float* db, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0, d = 0;
  for (int b = 0; b < B; ++b) {
    a += dwb[((long long)b * 2) * C + c];
    d += dwb[((long long)b * 2 + 1) * C + c];
  }
  if (dw) dw[c] = (float)a;
  if (db) db[c] = (float)d;
}

// ---- fused elementwise passes ------------------------------------------------
// y = act(x*scale[b,c] + shift[b,c]) (+ R)
__global__ void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, int sb_stride, const float* __restrict__ R,
                                  float* __restrict__ y, long long total4, int HW, int C4, int act, float ap) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int b = (int)(i / ((long long)HW * C4));
    const float4 v = ld4(x + i * 4);
    const float4 sc = ld4(scale + (long long)b * sb_stride + c4 * 4);
    const float4 sh = ld4(shift + (long long)b * sb_stride + c4 * 4);
    float4 o;
    o.x = fl::apply_act(v.x * sc.x + sh.x, act, ap);
    o.y = fl::apply_act(v.y * sc.y + sh.y, act, ap);
    o.z = fl::apply_act(v.z * sc.z + sh.z, act, ap);
    o.w = fl::apply_act(v.w * sc.w + sh.w, act, ap);
    if (R) {
      const float4 r = ld4(R + i * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    st4(y + i * 4, o);
  }
}

// dx = A*g + Bc*x + Cc, g = dy*act'(x*scale+shift)   (A/Bc/Cc/scale/shift per (b,c) or per c)
__global__ void norm_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                      const float* __restrict__ A, const float* __restrict__ Bc,
                                      const float* __restrict__ Cc, int sb_stride, float* __restrict__ dx,
                                      long long total4, int HW, int C4, int act, float ap) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int b = (int)(i / ((long long)HW * C4));
    const long long o = (long long)b * sb_stride + c4 * 4;
    const float4 v = ld4(x + i * 4), g0 = ld4(dy + i * 4);
    const float4 sc = ld4(scale + o), sh = ld4(shift + o), a = ld4(A + o), bb = ld4(Bc + o), cc = ld4(Cc + o);
    float4 r;
    r.x = a.x * (g0.x * act_grad(v.x * sc.x + sh.x, act, ap)) + bb.x * v.x + cc.x;
    r.y = a.y * (g0.y * act_grad(v.y * sc.y + sh.y, act, ap)) + bb.y * v.y + cc.y;
    r.z = a.z * (g0.z * act_grad(v.z * sc.z + sh.z, act, ap)) + bb.z * v.z + cc.z;
    r.w = a.w * (g0.w * act_grad(v.w * sc.w + sh.w, act, ap)) + bb.w * v.w + cc.w;
    st4(dx + i * 4, r);
  }
}

static int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

static int run_partial(int mode, const float* x, const float* dy, const float* scale, const float* shift, int sb_stride,
                       int act, float ap, float* partial, int B, int HW, int C, Slabs sl, hipStream_t st) {
  dim3 grid(fl::cdiv(C / 4, 64), sl.S, B);
  if (mode == 0)
    hipLaunchKernelGGL(chan_stats_partial<0>, grid, dim3(256), 0, st, x, dy, scale, shift, sb_stride, act, ap, partial, HW, C, sl.S, sl.pix);
  else
    hipLaunchKernelGGL(chan_stats_partial<1>, grid, dim3(256), 0, st, x, dy, scale, shift, sb_stride, act, ap, partial, HW, C, sl.S, sl.pix);
  return fl::check_launch("chan_stats_partial");
}

// ---- row LayerNorm (nn.LayerNorm over the last dim) ------------------------------
// y = act(mask * (xhat*w + b)); one 256-thread block per row, three passes over an L2-hot row.
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                     const float* __restrict__ mask, const float* __restrict__ R, float* __restrict__ y,
                                     float* __restrict__ mean, float* __restrict__ rstd, int D, float eps, int act, float ap) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* xr = x + row * D;
  const int D4 = D >> 2;
  float s = 0.f;
  for (int i = threadIdx.x; i < D4; i += blockDim.x) {
    const float4 v = ld4(xr + i * 4);
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float m = fl::block_sum(s, red) / D;
  float q = 0.f;
  for (int i = threadIdx.x; i < D4; i += blockDim.x) {
    const float4 v = ld4(xr + i * 4);
    const float a = v.x - m, bq = v.y - m, c = v.z - m, d = v.w - m;
    q += (a * a + bq * bq) + (c * c + d * d);
  }
  const float var = fl::block_sum(q, red) / D;
  const float r = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    if (mean) mean[row] = m;
    if (rstd) rstd[row] = r;
  }
  float* yr = y + row * D;
  const float* mr = mask ? mask + row * D : nullptr;
  for (int i = threadIdx.x; i < D4; i += blockDim.x) {
    const float4 v = ld4(xr + i * 4), ww = ld4(w + i * 4), bb = ld4(b + i * 4);
    float4 o;
    o.x = (v.x - m) * r * ww.x + bb.x;
    o.y = (v.y - m) * r * ww.y + bb.y;
    o.z = (v.z - m) * r * ww.z + bb.z;
    o.w = (v.w - m) * r * ww.w + bb.w;
    if (mr) {
      const float4 mk = ld4(mr + i * 4);
      o.x *= mk.x; o.y *= mk.y; o.z *= mk.z; o.w *= mk.w;
    }
    o.x = fl::apply_act(o.x, act, ap);
    o.y = fl::apply_act(o.y, act, ap);
    o.z = fl::apply_act(o.z, act, ap);
    o.w = fl::apply_act(o.w, act, ap);
    if (R) {
      const float4 rr = ld4(R + row * D + i * 4);
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    st4(yr + i * 4, o);
  }
}

// dx for one row; gl = dy*act'(u)*mask is the gradient w.r.t. the LN output (u = mask*(xhat*w+b))
__global__ void layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                        const float* __restrict__ w, const float* __restrict__ b,
                                        const float* __restrict__ mask, const float* __restrict__ mean,
                                        const float* __restrict__ rstd, const float* __restrict__ R,
                                        float* __restrict__ dx, int D, int act, float ap) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float m = mean[row], r = rstd[row];
  const float* xr = x + row * D;
  const float* dr = dy + row * D;
  const float* mr = mask ? mask + row * D : nullptr;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float xh = (xr[i] - m) * r;
    const float mk = mr ? mr[i] : 1.f;
    const float u = mk * (xh * w[i] + b[i]);
    const float gl = dr[i] * act_grad(u, act, ap) * mk;
    const float dxh = gl * w[i];
    s1 += dxh;
    s2 += dxh * xh;
  }
  const float t1 = fl::block_sum(s1, red) / D;
  const float t2 = fl::block_sum(s2, red) / D;
  float* dxr = dx + row * D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float xh = (xr[i] - m) * r;
    const float mk = mr ? mr[i] : 1.f;
    const float u = mk * (xh * w[i] + b[i]);
    const float gl = dr[i] * act_grad(u, act, ap) * mk;
    dxr[i] = r * (gl * w[i] - t1 - xh * t2) + (R ? R[row * D + i] : 0.f);
  }
}

// column reductions for dw/db: partial[slab][2][D]
__global__ void layernorm_bwd_wb_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                         const float* __restrict__ w, const float* __restrict__ b,
                                         const float* __restrict__ mask, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, float* __restrict__ partial, int rows, int D,
                                         int rows_per_slab, int act, float ap) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  const int r0 = blockIdx.y * rows_per_slab;
  int r1 = r0 + rows_per_slab;
  if (r1 > rows) r1 = rows;
  float aw = 0.f, ab = 0.f;
  const float wc = w[c], bc = b[c];
  for (int rr = r0; rr < r1; ++rr) {
    const long long o = (long long)rr * D + c;
    const float xh = (x[o] - mean[rr]) * rstd[rr];
    const float mk = mask ? mask[o] : 1.f;
    const float u = mk * (xh * wc + bc);
    const float gl = dy[o] * act_grad(u, act, ap) * mk;
    aw += gl * xh;
    ab += gl;
  }
  partial[((long long)blockIdx.y * 2) * D + c] = aw;
  partial[((long long)blockIdx.y * 2 + 1) * D + c] = ab;
}

}  // namespace

namespace {
// out[0] = mult * max|w| + max|b| (+ prev[0]): a rigorous bound on |w * n + b| for a normalised n with |n| <= mult
__global__ void affine_bound_kernel(const float* w, const float* b, int C, float mult, const float* prev, float* out) {
  __shared__ float red[8];
  float mw = w ? 0.f : 1.f, mb = 0.f;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    if (w) mw = fmaxf(mw, fabsf(w[i]));
    if (b) mb = fmaxf(mb, fabsf(b[i]));
  }
  mw = fl::block_max(mw, red);
  mb = fl::block_max(mb, red);
  if (threadIdx.x == 0) out[0] = mult * mw + mb + (prev ? prev[0] : 0.f);
}
}  // namespace

extern "C" {

size_t fl_norm_workspace_bytes(int B, int HW, int C) {
  Slabs sl = pick_slabs(B, HW, C);
  // partial slabs + per-(b,c) dw/db staging + collapsed (double) sums of the slabs
  return ((size_t)B * sl.S * 2 * C + (size_t)B * 2 * C) * sizeof(float) + 8 + (size_t)B * 2 * C * sizeof(double);
}

#define NORM_CHECKS(name)                                                                   \
  FL_REQUIRE(C % 4 == 0, name ": C=%d must be a multiple of 4", C);                          \
  FL_REQUIRE(ws && ws_bytes >= fl_norm_workspace_bytes(B, HW, C), name ": workspace too small"); \
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);                                     \
  Slabs sl = pick_slabs(B, HW, C);                                                             \
  float* partial = reinterpret_cast<float*>(ws);                                               \
  float* dwb = partial + (size_t)B * sl.S * 2 * C;                                             \
  double* csum = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(dwb + (size_t)B * 2 * C) + 7) & ~(uintptr_t)7); \
  (void)dwb; (void)csum;

int fl_instnorm_stats(const float* x, const float* w, const float* b, float eps, float* scale, float* shift,
                      float* mean, float* rstd, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_instnorm_stats");
  int rc = run_partial(0, x, nullptr, nullptr, nullptr, 0, 0, 0.f, partial, B, HW, C, sl, st);
  if (rc) return rc;
  hipLaunchKernelGGL(inorm_finalize_fwd, dim3(fl::cdiv((long long)B * C, 256)), dim3(256), 0, st, partial, w, b, eps, scale, shift, mean, rstd, B, HW, C, sl.S);
  return fl::check_launch("inorm_finalize_fwd");
}

int fl_groupnorm_stats(const float* x, const float* w, const float* b, float eps, int G, float* scale, float* shift,
                       float* mean, float* rstd, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_groupnorm_stats");
  FL_REQUIRE(G > 0 && C % G == 0 && w && b, "fl_groupnorm_stats: bad groups");
  int rc = run_partial(0, x, nullptr, nullptr, nullptr, 0, 0, 0.f, partial, B, HW, C, sl, st);
  if (rc) return rc;
  if (sl.S > 2) {  // many slabs: tree-reduce them first
    hipLaunchKernelGGL(collapse_rows, dim3(fl::cdiv(2 * C, 64), B), dim3(1024), 0, st, partial, csum, sl.S, 2 * C);
    hipLaunchKernelGGL(gnorm_finalize_fwd<double>, dim3(fl::cdiv((long long)B * G, 128)), dim3(128), 0, st, csum, w, b, eps, scale, shift, mean, rstd, B, HW, C, G, 1);
  } else {
    hipLaunchKernelGGL(gnorm_finalize_fwd<float>, dim3(fl::cdiv((long long)B * G, 128)), dim3(128), 0, st, partial, w, b, eps, scale, shift, mean, rstd, B, HW, C, G, sl.S);
  }
  return fl::check_launch("gnorm_finalize_fwd");
}

int fl_batchnorm_stats(const float* x, const float* w, const float* b, float eps, float momentum, float* running_mean,
                       float* running_var, int training, float* scale, float* shift, float* mean, float* rstd, int B,
                       int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_batchnorm_stats");
  FL_REQUIRE(w && b, "fl_batchnorm_stats: affine params required");
  if (training) {
    int rc = run_partial(0, x, nullptr, nullptr, nullptr, 0, 0, 0.f, partial, B, HW, C, sl, st);
    if (rc) return rc;
  }
  if (training && B * sl.S > 8) {
    hipLaunchKernelGGL(collapse_rows, dim3(fl::cdiv(2 * C, 64), 1), dim3(1024), 0, st, partial, csum, B * sl.S, 2 * C);
    hipLaunchKernelGGL(bnorm_finalize_fwd<double>, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, csum, w, b, eps, momentum, running_mean, running_var, training, scale, shift, mean, rstd, B, HW, C, 1, 1);
  } else {
    hipLaunchKernelGGL(bnorm_finalize_fwd<float>, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, partial, w, b, eps, momentum, running_mean, running_var, training, scale, shift, mean, rstd, B, HW, C, sl.S, B);
  }
  return fl::check_launch("bnorm_finalize_fwd");
}

int fl_affine_act(const float* x, const float* scale, const float* shift, int per_image, const float* R, float* y, int B,
                  int HW, int C, int act, float act_param, fl_stream_t stream) {
  FL_REQUIRE(C % 4 == 0 && x && y && scale && shift, "fl_affine_act: bad args");
  const long long total4 = (long long)B * HW * (C / 4);
  if (total4 == 0) return FL_OK;
  hipLaunchKernelGGL(affine_act_kernel, dim3(ew_blocks(total4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     scale, shift, per_image ? C : 0, R, y, total4, HW, C / 4, act, act_param);
  return fl::check_launch("affine_act_kernel");
}

// Backward of y = act(norm(x)): inputs dy (grad of y), x, the forward's scale/shift/mean/rstd.
int fl_instnorm_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                    const float* mean, const float* rstd, int act, float act_param, float* dx, float* dw, float* db,
                    float* coef /*3*B*C*/, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_instnorm_bwd");
  int rc = run_partial(1, x, dy, scale, shift, C, act, act_param, partial, B, HW, C, sl, st);
  if (rc) return rc;
  float *A = coef, *Bc = coef + (size_t)B * C, *Cc = coef + 2 * (size_t)B * C;
  hipLaunchKernelGGL(inorm_finalize_bwd, dim3(fl::cdiv((long long)B * C, 256)), dim3(256), 0, st, partial, w, mean, rstd, A, Bc, Cc, dwb, B, HW, C, sl.S);
  if (dw || db) hipLaunchKernelGGL(dwb_reduce, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, dwb, dw, db, B, C);
  const long long total4 = (long long)B * HW * (C / 4);
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ew_blocks(total4)), dim3(256), 0, st, dy, x, scale, shift, A, Bc, Cc, C, dx, total4, HW, C / 4, act, act_param);
  return fl::check_launch("fl_instnorm_bwd");
}

int fl_groupnorm_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                     const float* mean, const float* rstd, int G, int act, float act_param, float* dx, float* dw,
                     float* db, float* coef, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_groupnorm_bwd");
  int rc = run_partial(1, x, dy, scale, shift, C, act, act_param, partial, B, HW, C, sl, st);
  if (rc) return rc;
  float *A = coef, *Bc = coef + (size_t)B * C, *Cc = coef + 2 * (size_t)B * C;
  if (sl.S > 2) {
    hipLaunchKernelGGL(collapse_rows, dim3(fl::cdiv(2 * C, 64), B), dim3(1024), 0, st, partial, csum, sl.S, 2 * C);
    hipLaunchKernelGGL(gnorm_finalize_bwd<double>, dim3(fl::cdiv((long long)B * G, 128)), dim3(128), 0, st, csum, w, mean, rstd, A, Bc, Cc, dwb, B, HW, C, G, 1);
  } else {
    hipLaunchKernelGGL(gnorm_finalize_bwd<float>, dim3(fl::cdiv((long long)B * G, 128)), dim3(128), 0, st, partial, w, mean, rstd, A, Bc, Cc, dwb, B, HW, C, G, sl.S);
  }
  if (dw || db) hipLaunchKernelGGL(dwb_reduce, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, dwb, dw, db, B, C);
  const long long total4 = (long long)B * HW * (C / 4);
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ew_blocks(total4)), dim3(256), 0, st, dy, x, scale, shift, A, Bc, Cc, C, dx, total4, HW, C / 4, act, act_param);
  return fl::check_launch("fl_groupnorm_bwd");
}

int fl_batchnorm_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                     const float* mean, const float* rstd, int act, float act_param, float* dx, float* dw, float* db,
                     float* coef /*3*C*/, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_batchnorm_bwd");
  int rc = run_partial(1, x, dy, scale, shift, 0, act, act_param, partial, B, HW, C, sl, st);
  if (rc) return rc;
  float *A = coef, *Bc = coef + C, *Cc = coef + 2 * (size_t)C;
  if (B * sl.S > 8) {
    hipLaunchKernelGGL(collapse_rows, dim3(fl::cdiv(2 * C, 64), 1), dim3(1024), 0, st, partial, csum, B * sl.S, 2 * C);
    hipLaunchKernelGGL(bnorm_finalize_bwd<double>, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, csum, w, mean, rstd, A, Bc, Cc, dw, db, B, HW, C, 1, 1);
  } else {
    hipLaunchKernelGGL(bnorm_finalize_bwd<float>, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, partial, w, mean, rstd, A, Bc, Cc, dw, db, B, HW, C, sl.S, B);
  }
  const long long total4 = (long long)B * HW * (C / 4);
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ew_blocks(total4)), dim3(256), 0, st, dy, x, scale, shift, A, Bc, Cc, 0, dx, total4, HW, C / 4, act, act_param);
  return fl::check_launch("fl_batchnorm_bwd");
}

/* ---- data-parallel (synchronised) batch norm: the two reductions are split at the point where
 * ranks exchange their per-channel sums (2C doubles), so the statistics are those of the global
 * batch as torch's SyncBatchNorm computes them. */
int fl_batchnorm_sums(const float* x, const float* dy, const float* scale, const float* shift, int act,
                      float act_param, double* sums, int B, int HW, int C, void* ws, size_t ws_bytes,
                      fl_stream_t stream) {
  NORM_CHECKS("fl_batchnorm_sums");
  FL_REQUIRE(x && sums, "fl_batchnorm_sums: null argument");
  FL_REQUIRE(!dy || (scale && shift), "fl_batchnorm_sums: backward sums need scale/shift");
  int rc = dy ? run_partial(1, x, dy, scale, shift, 0, act, act_param, partial, B, HW, C, sl, st)
              : run_partial(0, x, nullptr, nullptr, nullptr, 0, 0, 0.f, partial, B, HW, C, sl, st);
  if (rc) return rc;
  hipLaunchKernelGGL(collapse_rows, dim3(fl::cdiv(2 * C, 64), 1), dim3(1024), 0, st, partial, sums, B * sl.S, 2 * C);
  return fl::check_launch("fl_batchnorm_sums");
}

int fl_batchnorm_stats_from_sums(const double* sums, int64_t n_total, const float* w, const float* b, float eps,
                                 float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                 float* mean, float* rstd, int C, fl_stream_t stream) {
  FL_REQUIRE(sums && w && b && scale && shift, "fl_batchnorm_stats_from_sums: null argument");
  FL_REQUIRE(n_total > 0 && n_total < (1ll << 31), "fl_batchnorm_stats_from_sums: n_total=%lld out of range", (long long)n_total);
  hipLaunchKernelGGL(bnorm_finalize_fwd<double>, dim3(fl::cdiv(C, 128)), dim3(128), 0, reinterpret_cast<hipStream_t>(stream),
                     sums, w, b, eps, momentum, running_mean, running_var, 1, scale, shift, mean, rstd, 1, (int)n_total, C, 1, 1);
  return fl::check_launch("fl_batchnorm_stats_from_sums");
}

int fl_batchnorm_bwd_from_sums(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                               const float* mean, const float* rstd, int act, float act_param,
                               const double* sums_global, const double* sums_local, int64_t n_total, float* dx,
                               float* dw, float* db, float* coef /*8*C*/, int B, int HW, int C, fl_stream_t stream) {
  FL_REQUIRE(C % 4 == 0 && dy && x && scale && shift && w && mean && rstd && sums_global && sums_local && dx && dw && db && coef,
             "fl_batchnorm_bwd_from_sums: bad args (C=%d)", C);
  FL_REQUIRE(n_total > 0 && n_total < (1ll << 31), "fl_batchnorm_bwd_from_sums: n_total=%lld out of range", (long long)n_total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float *A = coef, *Bc = coef + C, *Cc = coef + 2 * (size_t)C, *t0 = coef + 3 * (size_t)C;
  // dx coefficients from the global sums; dw/db (this rank's share, all-reduced with the other gradients) from the local ones
  hipLaunchKernelGGL(bnorm_finalize_bwd<double>, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, sums_global, w, mean, rstd, A, Bc, Cc, t0, t0 + C, 1, (int)n_total, C, 1, 1);
  hipLaunchKernelGGL(bnorm_finalize_bwd<double>, dim3(fl::cdiv(C, 128)), dim3(128), 0, st, sums_local, w, mean, rstd, t0, t0 + C, t0 + 2 * (size_t)C, dw, db, 1, (int)n_total, C, 1, 1);
  const long long total4 = (long long)B * HW * (C / 4);
  if (total4 == 0) return FL_OK;
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ew_blocks(total4)), dim3(256), 0, st, dy, x, scale, shift, A, Bc, Cc, 0, dx, total4, HW, C / 4, act, act_param);
  return fl::check_launch("fl_batchnorm_bwd_from_sums");
}

int fl_layernorm_fwd(const float* x, const float* w, const float* b, const float* mask, const float* R, float* y,
                     float* mean, float* rstd, int rows, int D, float eps, int act, float act_param, fl_stream_t stream) {
  FL_REQUIRE(x && w && b && y && D % 4 == 0, "fl_layernorm_fwd: bad args (D=%d)", D);
  if (rows == 0) return FL_OK;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w, b, mask, R, y, mean, rstd, D, eps, act, act_param);
  return fl::check_launch("layernorm_fwd_kernel");
}

size_t fl_layernorm_bwd_workspace_bytes(int rows, int D) {
  const int slabs = fl::cdiv(rows, 16);
  return (size_t)slabs * 2 * D * sizeof(float);
}

int fl_layernorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* mask,
                     const float* mean, const float* rstd, const float* R, float* dx, float* dw, float* db, int rows, int D,
                     int act, float act_param, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(dy && x && w && b && mean && rstd && dx, "fl_layernorm_bwd: null arg");
  if (rows == 0) return FL_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3(rows), dim3(256), 0, st, dy, x, w, b, mask, mean, rstd, R, dx, D, act, act_param);
  if (dw || db) {
    FL_REQUIRE(ws && ws_bytes >= fl_layernorm_bwd_workspace_bytes(rows, D), "fl_layernorm_bwd: workspace too small");
    const int rps = 16, slabs = fl::cdiv(rows, rps);
    float* partial = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(layernorm_bwd_wb_partial, dim3(fl::cdiv(D, 128), slabs), dim3(128), 0, st, dy, x, w, b, mask, mean, rstd, partial, rows, D, rps, act, act_param);
    hipLaunchKernelGGL(dwb_reduce, dim3(fl::cdiv(D, 128)), dim3(128), 0, st, partial, dw, db, slabs, D);
  }
  return fl::check_launch("fl_layernorm_bwd");
}

}  // extern "C"

namespace {
__global__ void chan_sums_finalize(const float* partial, float* out1, float* out2, int B, int C, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  double s1, s2;
  sum_slabs(partial, b, S, C, c, s1, s2);
  if (out1) out1[i] = (float)s1;
  if (out2) out2[i] = (float)s2;
}
}  // namespace

extern "C" {
// out1[b,c] = sum_pix a, out2[b,c] = sum_pix a*x with a = dy (a = x when dy is NULL): per-image channel sums
// (AdaptiveAvgPool2d(1) of v2:56,65 and its backward).
int fl_chan_sums(const float* x, const float* dy, float* out1, float* out2, int B, int HW, int C, void* ws,
                 size_t ws_bytes, fl_stream_t stream) {
  NORM_CHECKS("fl_chan_sums");
  int rc = run_partial(dy ? 1 : 0, x, dy, nullptr, nullptr, 0, 0, 0.f, partial, B, HW, C, sl, st);
  if (rc) return rc;
  hipLaunchKernelGGL(chan_sums_finalize, dim3(fl::cdiv((long long)B * C, 256)), dim3(256), 0, st, partial, out1, out2, B, C, sl.S);
  return fl::check_launch("chan_sums_finalize");
}
// |LayerNorm-style affine output| <= mult * max|w| + max|b| + prev: with mult = sqrt(#elements normalised together) this
// bounds LayerNorm2d / LayerNorm outputs (|x - mean| * rstd <= sqrt(n)); Swish and gates in (0,1) only shrink it.
int fl_affine_bound(const float* w, const float* b, int C, float mult, const float* prev, float* out, fl_stream_t stream) {
  FL_REQUIRE(out && C > 0 && mult >= 0.f, "fl_affine_bound: bad arguments");
  hipLaunchKernelGGL(affine_bound_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, C, mult, prev, out);
  return fl::check_launch("affine_bound_kernel");
}
}  // extern "C"
