// Elementwise / gather / diffusion-update / RNG kernels (all HBM- or latency-bound).
#include "fl_common.h"

#define FL_MULTI_MAX 96

namespace {

__device__ __forceinline__ float act_grad(float u, int act, float p) {
  switch (act) {
    case FL_ACT_SWISH: return fl::swish_grad(u);
    case FL_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case FL_ACT_LRELU: return u > 0.f ? 1.f : p;
    case FL_ACT_SIGMOID: { float s = fl::sigmoid_acc(u); return s * (1.f - s); }
    default: return 1.f;
  }
}

#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

static int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- layout ------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW, int Cpad) {
  const long long total = (long long)B * HW * Cpad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const long long r = i / Cpad;
    const int p = (int)(r % HW);
    const int b = (int)(r / HW);
    y[i] = c < C ? x[((long long)b * C + c) * HW + p] : 0.f;
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW, int Cpad) {
  const long long total = (long long)B * C * HW;
  GRID_STRIDE(i, total) {
    const int p = (int)(i % HW);
    const long long r = i / HW;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    y[i] = x[((long long)b * HW + p) * Cpad + c];
  }
}

// ---- activations ---------------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int act, float p) {
  GRID_STRIDE(i, n) y[i] = fl::apply_act(x[i], act, p);
}
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                               long long n, int act, float p) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * act_grad(x[i], act, p);
}
__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, const float* __restrict__ y,
                             float* __restrict__ out, long long n) {
  GRID_STRIDE(i, n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
// Many small copies / accumulations in ONE launch (per-parameter gradients into the flat gradient buffer: the VAE-GAN
// step did 196 of them, 4.9 us each = launch floor). The table travels in the kernel arguments.
struct MultiAxpby {
  const float* src[FL_MULTI_MAX];
  float* dst[FL_MULTI_MAX];
  int blk0[FL_MULTI_MAX + 1];  // first block of each entry (1024 elements per block)
  long long n[FL_MULTI_MAX];
};
__global__ void multi_axpby_kernel(const MultiAxpby t, int entries, float a, float b) {
  int lo = 0, hi = entries - 1;  // last entry whose first block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long base = (long long)(blockIdx.x - t.blk0[lo]) * 1024;
  const float* __restrict__ x = t.src[lo];
  float* __restrict__ y = t.dst[lo];
  const long long n = t.n[lo];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) y[i] = b != 0.f ? a * x[i] + b * y[i] : a * x[i];
  }
}
// Column sums of many [M][N] matrices in ONE launch (the bias gradients of every nn.Linear of the denoiser's backward:
// 27 launches of ~5 us each otherwise, and each of them a link in the step's dependent chain). One block = 64 columns
// of one matrix: 4 row groups x 64 lanes, combined through LDS in a fixed order.
constexpr int FL_MCS_MAX = 64;
struct MultiColsum {
  const float* x[FL_MCS_MAX];
  float* out[FL_MCS_MAX];
  int blk0[FL_MCS_MAX + 1];
  int M[FL_MCS_MAX], N[FL_MCS_MAX];
  float alpha[FL_MCS_MAX];
};
__global__ __launch_bounds__(256) void multi_colsum_kernel(const MultiColsum t, int entries) {
  __shared__ float part[4][64];
  int lo = 0, hi = entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int N = t.N[lo], M = t.M[lo];
  const int c = (blockIdx.x - t.blk0[lo]) * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const float* __restrict__ x = t.x[lo];
  float s = 0.f;
  if (c < N)
    for (int m = rg; m < M; m += 4) s += x[(long long)m * N + c];
  part[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < N) t.out[lo][c] = t.alpha[lo] * (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]);
}
// out[b, c] = v[b, c] * hm[b, c / (D/heads)]   (attention-dropout on a length-1 sequence, v2:514)
__global__ void scale_heads_kernel(const float* __restrict__ v, const float* __restrict__ hm, float* __restrict__ out,
                                   int B, int D, int heads) {
  const long long n = (long long)B * D;
  const int hd = D / heads;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % D);
    const int b = (int)(i / D);
    out[i] = v[i] * hm[b * heads + c / hd];
  }
}
// out[b, :] = x[b, :] + y[b or 0, :]  (broadcast a single row over the batch: shared-t sampling)
__global__ void add_rows_kernel(const float* __restrict__ x, const float* __restrict__ y, int y_rows,
                                float* __restrict__ out, int B, int D) {
  const long long n = (long long)B * D;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % D);
    const int b = (int)(i / D);
    out[i] = x[i] + y[(long long)(y_rows == 1 ? 0 : b) * D + c];
  }
}

// ---- embeddings --------------------------------------------------------------
// sinusoidal time features (v2:410-414): [sin(t*f_i), cos(t*f_i)]. The constant table
// f_i = exp(-i*ln(1e4)/(half-1)) is an input (computed once by the host exactly as v2:411-412):
// a 1-ulp difference in f_i would be amplified ~1000x by t.
__global__ void time_sinusoid_kernel(const long long* __restrict__ t, const float* __restrict__ freqs,
                                     float* __restrict__ out, int Bt, int n_ch) {
  const int half = n_ch / 2;
  const long long n = (long long)Bt * half;
  GRID_STRIDE(i, n) {
    const int j = (int)(i % half);
    const int b = (int)(i / half);
    const float a = (float)t[b] * freqs[j];
    out[(long long)b * n_ch + j] = sinf(a);
    out[(long long)b * n_ch + half + j] = cosf(a);
  }
}
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const long long* __restrict__ idx,
                                     float* __restrict__ out, int n, int D) {
  const long long total = (long long)n * D;
  GRID_STRIDE(i, total) {
    const int d = (int)(i % D);
    const int r = (int)(i / D);
    out[i] = table[idx[r] * D + d];
  }
}
// deterministic scatter-add: one thread per (table row, d) scans the batch
__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ idx,
                                     float* __restrict__ dtable, int n, int D, int rows) {
  const long long total = (long long)rows * D;
  GRID_STRIDE(i, total) {
    const int d = (int)(i % D);
    const int r = (int)(i / D);
    float a = 0.f;
    for (int j = 0; j < n; ++j)
      if (idx[j] == r) a += dout[(long long)j * D + d];
    dtable[i] = a;
  }
}

// ---- diffusion process (v2:574-592) ------------------------------------------------
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                const long long* __restrict__ t, const float* __restrict__ alpha_bar,
                                float* __restrict__ xt, int B, int D) {
  const long long n = (long long)B * D;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / D);
    const float ab = alpha_bar[t[b]];
    xt[i] = sqrtf(ab) * x0[i] + sqrtf(1.f - ab) * eps[i];
  }
}
// x_{t-1} = (x_t - (1-a_t)/sqrt(1-ab_t) * eps_theta)/sqrt(a_t) + [t>0] sqrt(beta_t) * noise
// t comes from device memory (t_ptr[0], int32) so the step is hipGraph-replayable.
__global__ void p_sample_kernel(const float* __restrict__ xt, const float* __restrict__ eps_theta,
                                const float* __restrict__ noise, const int* __restrict__ t_ptr,
                                const float* __restrict__ alpha, const float* __restrict__ alpha_bar,
                                const float* __restrict__ beta, float* __restrict__ out, long long n) {
  const int t = t_ptr[0];
  const float a = alpha[t], ab = alpha_bar[t];
  const float coef = (1.f - a) / sqrtf(1.f - ab);
  const float sa = sqrtf(a), sg = sqrtf(beta[t]);
  GRID_STRIDE(i, n) {
    float m = (xt[i] - coef * eps_theta[i]) / sa;
    if (t > 0 && noise) m += sg * noise[i];
    out[i] = m;
  }
}
__global__ void step_counter_kernel(int* t_ptr, long long* t64_ptr, int delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int v = t_ptr[0] + delta;
    t_ptr[0] = v;
    if (t64_ptr) t64_ptr[0] = v;
  }
}
// z = mu + eps * exp(0.5 * clamp(logvar, -2, 10))   (v2:339-343)
__global__ void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                               const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ lv_clamped,
                               long long n) {
  GRID_STRIDE(i, n) {
    const float lv = fminf(fmaxf(logvar[i], -2.0f), 10.0f);
    if (lv_clamped) lv_clamped[i] = lv;
    if (z) z[i] = mu[i] + eps[i] * expf(0.5f * lv);
  }
}

// ---- Philox4x32-10 counter RNG (graph-safe: state = (seed, offset) scalars / device offset) ------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// 4 outputs per counter. mode 0: N(0,1) (Box-Muller), 1: dropout scale {0, 1/(1-p)}, 2: U[0,1)
__global__ void philox_fill_kernel(float* __restrict__ out, long long n, uint64_t seed, uint64_t offset,
                                   const long long* __restrict__ dev_offset, int mode, float p) {
  const uint64_t base = offset + (dev_offset ? (uint64_t)dev_offset[0] : 0ull);
  const long long n4 = (n + 3) / 4;
  GRID_STRIDE(i, n4) {
    uint32_t r[4];
    philox4(seed, base + (uint64_t)i, r);
    float v[4];
    if (mode == 0) {
      const float u1 = u01(r[0]), u2 = u01(r[1]), u3 = u01(r[2]), u4 = u01(r[3]);
      const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
      float s1, c1, s2, c2;
      sincosf(6.283185307179586f * u2, &s1, &c1);
      sincosf(6.283185307179586f * u4, &s2, &c2);
      v[0] = ra * c1; v[1] = ra * s1; v[2] = rb * c2; v[3] = rb * s2;
    } else if (mode == 1) {
      const float keep = 1.f / (1.f - p);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = u01(r[j]) < p ? 0.f : keep;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = u01(r[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i * 4 + j < n) out[i * 4 + j] = v[j];
  }
}
__global__ void philox_randint_kernel(long long* __restrict__ out, long long n, uint64_t seed, uint64_t offset,
                                      const long long* __restrict__ dev_offset, int hi) {
  const uint64_t base = offset + (dev_offset ? (uint64_t)dev_offset[0] : 0ull);
  GRID_STRIDE(i, n) {
    uint32_t r[4];
    philox4(seed, base + (uint64_t)i, r);
    out[i] = (long long)(((uint64_t)r[0] * (uint64_t)hi) >> 32);
  }
}
__global__ void advance_offset_kernel(long long* dev_offset, long long delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dev_offset[0] += delta;
}

// out = a + sigmoid(w[0]) * b   (learned residual gate, v2:561), w on the device
__global__ void gated_add_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                                 float* __restrict__ out, long long n) {
  const float s = fl::sigmoid_acc(w[0]);
  GRID_STRIDE(i, n) out[i] = (a ? a[i] : 0.f) + s * b[i];
}
// out[0] = s(1-s) * sum x*y, s = sigmoid(w[0])  (gradient of the gate); single block
__global__ void gate_grad_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
                                 float* __restrict__ out, long long n) {
  __shared__ float red[16];
  float a = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) a += x[i] * y[i];
  a = fl::block_sum(a, red);
  if (threadIdx.x == 0) {
    const float s = fl::sigmoid_acc(w[0]);
    out[0] = s * (1.f - s) * a;
  }
}

// column sums: out[n] = sum_m x[m][n]  (bias gradients) — two-stage deterministic.
// Both stages use 64 columns x several row lanes per block with an LDS tree: the first version walked each slab, then
// all slabs, with ONE thread per column (64-thread blocks): 39 + 36 us per call for conv-sized dy, 10 % of the VAE-GAN step.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int M,
                                                              int N, long long ldx, int rows_per_slab) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_slab;
  int r1 = r0 + rows_per_slab;
  if (r1 > M) r1 = M;
  float a0 = 0.f, a1 = 0.f;
  if (c < N) {
    int r = r0 + rl;
    for (; r + 4 < r1; r += 8) {  // two independent chains per thread
      a0 += x[(long long)r * ldx + c];
      a1 += x[(long long)(r + 4) * ldx + c];
    }
    if (r < r1) a0 += x[(long long)r * ldx + c];
  }
  red[rl][cl] = a0 + a1;
  __syncthreads();
  if (rl == 0 && c < N) partial[(long long)blockIdx.y * N + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// short matrices (M <= 1024: the denoiser's bias gradients, M = batch): the whole reduction in ONE launch
__global__ __launch_bounds__(1024) void colsum_small_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N,
                                                             long long ldx, float alpha, float beta) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < N)
    for (int r = rl; r < M; r += 16) a += x[(long long)r * ldx + c];
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < N) {
#pragma unroll
    for (int k = 1; k < 16; ++k) a += red[k][cl];
    out[c] = beta != 0.f ? beta * out[c] + alpha * a : alpha * a;
  }
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int slabs, int N, float alpha, float beta) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < N)
    for (int s = rl; s < slabs; s += 16) a += partial[(long long)s * N + c];
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < N) {
#pragma unroll
    for (int k = 1; k < 16; ++k) a += red[k][cl];
    out[c] = beta != 0.f ? beta * out[c] + alpha * a : alpha * a;
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int fl_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int Cpad, fl_stream_t stream) {
  FL_REQUIRE(x && y && Cpad >= C, "fl_nchw_to_nhwc: bad args");
  const long long total = (long long)B * HW * Cpad;
  if (!total) return FL_OK;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), x, y, B, C, HW, Cpad);
  return fl::check_launch("nchw_to_nhwc");
}
int fl_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, int Cpad, fl_stream_t stream) {
  FL_REQUIRE(x && y && Cpad >= C, "fl_nhwc_to_nchw: bad args");
  const long long total = (long long)B * HW * C;
  if (!total) return FL_OK;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), x, y, B, C, HW, Cpad);
  return fl::check_launch("nhwc_to_nchw");
}
int fl_act_fwd(const float* x, float* y, int64_t n, int act, float p, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), x, y, (long long)n, act, p);
  return fl::check_launch("act_fwd");
}
int fl_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, float p, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), dy, x, dx, (long long)n, act, p);
  return fl::check_launch("act_bwd");
}
int fl_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), a, x, b, y, out, (long long)n);
  return fl::check_launch("axpby");
}
int fl_multi_axpby(const void* const* src, void* const* dst, const int64_t* n, int entries, float a, float b,
                   fl_stream_t stream) {
  FL_REQUIRE(entries >= 0 && (entries == 0 || (src && dst && n)), "fl_multi_axpby: bad args");
  for (int e0 = 0; e0 < entries; e0 += FL_MULTI_MAX) {
    MultiAxpby t;
    const int cnt = entries - e0 < FL_MULTI_MAX ? entries - e0 : FL_MULTI_MAX;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      FL_REQUIRE(src[e0 + i] && dst[e0 + i] && n[e0 + i] >= 0, "fl_multi_axpby: null entry %d", e0 + i);
      t.src[i] = reinterpret_cast<const float*>(src[e0 + i]);
      t.dst[i] = reinterpret_cast<float*>(dst[e0 + i]);
      t.n[i] = n[e0 + i];
      t.blk0[i] = blocks;
      blocks += (int)((n[e0 + i] + 1023) / 1024);
    }
    t.blk0[cnt] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(multi_axpby_kernel, dim3(blocks), dim3(256), 0, ST(stream), t, cnt, a, b);
  }
  return fl::check_launch("multi_axpby");
}
int fl_multi_colsum(const void* const* x, void* const* out, const int* M, const int* N, const float* alpha, int entries,
                    fl_stream_t stream) {
  FL_REQUIRE(entries >= 0 && (entries == 0 || (x && out && M && N && alpha)), "fl_multi_colsum: bad args");
  for (int e0 = 0; e0 < entries; e0 += FL_MCS_MAX) {
    MultiColsum t;
    const int cnt = entries - e0 < FL_MCS_MAX ? entries - e0 : FL_MCS_MAX;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      FL_REQUIRE(x[e0 + i] && out[e0 + i] && M[e0 + i] >= 0 && N[e0 + i] >= 0, "fl_multi_colsum: bad entry %d", e0 + i);
      t.x[i] = reinterpret_cast<const float*>(x[e0 + i]);
      t.out[i] = reinterpret_cast<float*>(out[e0 + i]);
      t.M[i] = M[e0 + i];
      t.N[i] = N[e0 + i];
      t.alpha[i] = alpha[e0 + i];
      t.blk0[i] = blocks;
      blocks += (N[e0 + i] + 63) / 64;
    }
    t.blk0[cnt] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(multi_colsum_kernel, dim3(blocks), dim3(256), 0, ST(stream), t, cnt);
  }
  return fl::check_launch("multi_colsum");
}
int fl_scale_heads(const float* v, const float* head_mask, float* out, int B, int D, int heads, fl_stream_t stream) {
  FL_REQUIRE(heads > 0 && D % heads == 0, "fl_scale_heads: bad heads");
  const long long n = (long long)B * D;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(scale_heads_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), v, head_mask, out, B, D, heads);
  return fl::check_launch("scale_heads");
}
int fl_add_rows(const float* x, const float* y, int y_rows, float* out, int B, int D, fl_stream_t stream) {
  FL_REQUIRE(y_rows == 1 || y_rows == B, "fl_add_rows: y_rows must be 1 or B");
  const long long n = (long long)B * D;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(add_rows_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), x, y, y_rows, out, B, D);
  return fl::check_launch("add_rows");
}
int fl_time_sinusoid(const int64_t* t, const float* freqs, float* out, int Bt, int n_ch, fl_stream_t stream) {
  FL_REQUIRE(n_ch % 2 == 0 && n_ch >= 4 && freqs, "fl_time_sinusoid: n_ch must be even, freqs non-null");
  const long long n = (long long)Bt * (n_ch / 2);
  if (!n) return FL_OK;
  hipLaunchKernelGGL(time_sinusoid_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), (const long long*)t, freqs, out, Bt, n_ch);
  return fl::check_launch("time_sinusoid");
}
int fl_embedding_fwd(const float* table, const int64_t* idx, float* out, int n, int D, fl_stream_t stream) {
  const long long total = (long long)n * D;
  if (!total) return FL_OK;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), table, (const long long*)idx, out, n, D);
  return fl::check_launch("embedding_fwd");
}
int fl_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, int n, int D, int rows, fl_stream_t stream) {
  const long long total = (long long)rows * D;
  if (!total) return FL_OK;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), dout, (const long long*)idx, dtable, n, D, rows);
  return fl::check_launch("embedding_bwd");
}
int fl_q_sample(const float* x0, const float* eps, const int64_t* t, const float* alpha_bar, float* xt, int B, int D,
                fl_stream_t stream) {
  const long long n = (long long)B * D;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(q_sample_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), x0, eps, (const long long*)t, alpha_bar, xt, B, D);
  return fl::check_launch("q_sample");
}
int fl_p_sample_update(const float* xt, const float* eps_theta, const float* noise, const int32_t* t_ptr,
                       const float* alpha, const float* alpha_bar, const float* beta, float* out, int64_t n,
                       fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(p_sample_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), xt, eps_theta, noise, t_ptr, alpha, alpha_bar, beta, out, (long long)n);
  return fl::check_launch("p_sample");
}
int fl_step_counter_add(int32_t* t_ptr, int64_t* t64_ptr, int delta, fl_stream_t stream) {
  hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(64), 0, ST(stream), t_ptr, (long long*)t64_ptr, delta);
  return fl::check_launch("step_counter");
}
int fl_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, float* logvar_clamped,
                      int64_t n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(reparam_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), mu, logvar, eps, z, logvar_clamped, (long long)n);
  return fl::check_launch("reparam");
}
int fl_philox_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, const int64_t* dev_offset, int mode, float p,
                   fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(philox_fill_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, ST(stream), out, (long long)n, seed, offset, (const long long*)dev_offset, mode, p);
  return fl::check_launch("philox_fill");
}
int fl_philox_randint(int64_t* out, int64_t n, uint64_t seed, uint64_t offset, const int64_t* dev_offset, int hi,
                      fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(philox_randint_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), (long long*)out, (long long)n, seed, offset, (const long long*)dev_offset, hi);
  return fl::check_launch("philox_randint");
}
int fl_advance_offset(int64_t* dev_offset, int64_t delta, fl_stream_t stream) {
  hipLaunchKernelGGL(advance_offset_kernel, dim3(1), dim3(64), 0, ST(stream), (long long*)dev_offset, (long long)delta);
  return fl::check_launch("advance_offset");
}
int fl_gated_add(const float* a, const float* b, const float* w, float* out, int64_t n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(gated_add_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), a, b, w, out, (long long)n);
  return fl::check_launch("gated_add");
}
int fl_gate_grad(const float* x, const float* y, const float* w, float* out, int64_t n, fl_stream_t stream) {
  hipLaunchKernelGGL(gate_grad_kernel, dim3(1), dim3(1024), 0, ST(stream), x, y, w, out, (long long)n);
  return fl::check_launch("gate_grad");
}
// rows per slab: at least 32, and at most 1024 slabs (conv bias gradients reduce over B*H*W rows: enough blocks to
// stream dy at HBM rate, few enough that the 16-lane final pass stays short)
static int colsum_rps(int M) {
  int rps = fl::cdiv(M, 1024);
  return rps < 32 ? 32 : rps;
}
size_t fl_colsum_workspace_bytes(int M, int N) { return (size_t)fl::cdiv(M, colsum_rps(M)) * N * sizeof(float) + 64; }
int fl_colsum(const float* x, int64_t ldx, float* out, int M, int N, float alpha, float beta, void* ws, size_t ws_bytes,
              fl_stream_t stream) {
  FL_REQUIRE(ws && ws_bytes >= fl_colsum_workspace_bytes(M, N), "fl_colsum: workspace too small");
  if (!N) return FL_OK;
  if (M <= 1024) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3(fl::cdiv(N, 64)), dim3(1024), 0, ST(stream), x, out, M, N, (long long)ldx, alpha, beta);
    return fl::check_launch("colsum_small");
  }
  const int rps = colsum_rps(M), slabs = fl::cdiv(M, rps);
  float* partial = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(fl::cdiv(N, 64), slabs > 0 ? slabs : 1), dim3(256), 0, ST(stream), x, partial, M, N, (long long)ldx, rps);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(fl::cdiv(N, 64)), dim3(1024), 0, ST(stream), partial, out, slabs, N, alpha, beta);
  return fl::check_launch("colsum");
}

}  // extern "C"
