// The latency-bound chain between the frozen encoder's big head GEMM and the denoiser (v2:1257-1262) as TWO launches instead of
// fourteen (r03 timeline: LayerNorm, GEMM, split-K reduce per head, three Philox fills, two counter bumps, reparameterize,
// randint, q_sample = 76 us of ~5 us launches at batch 256):
//   * encoder_heads_kernel: both heads' LayerNorm -> Swish -> Linear(K -> L) (v2:210-221) on exact-fp32 MFMAs, and - when asked -
//     reparameterize (v2:339-343) with the N(0,1) draws made in the epilogue;
//   * diffuse_prep_kernel: t ~ randint, noise ~ N(0,1), x_t = q_sample(z, t, noise) (v2:574-578, v2:1260-1261) and the step's
//     dropout-scale draws, element-wise.
// Every draw is the number fl_philox_fill / fl_philox_randint would have produced at the same (seed, counter): the fused and the
// multi-launch paths are interchangeable draw for draw (tests/test_latent_prep_gpu.py).
#include "fl_common.h"

namespace {

// ---- Philox4x32-10, as in elementwise.hip (same counter -> same numbers)
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
// the four N(0,1) values of counter ctr (philox_fill_kernel mode 0: Box-Muller on (r0, r1) and (r2, r3))
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t ctr, float (&v)[4]) {
  uint32_t r[4];
  philox4(seed, ctr, r);
  const float u1 = u01(r[0]), u2 = u01(r[1]), u3 = u01(r[2]), u4 = u01(r[3]);
  const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
  float s1, c1, s2, c2;
  sincosf(6.283185307179586f * u2, &s1, &c1);
  sincosf(6.283185307179586f * u4, &s2, &c2);
  v[0] = ra * c1; v[1] = ra * s1; v[2] = rb * c2; v[3] = rb * s2;
}

// ---------------------------------------------------------------------------------------------------------------------
// encoder_heads_kernel. Block = 32 batch rows x 32 latent columns of BOTH heads, 512 threads; wave w works for head w >> 2:
//   0. it requests its weight fragments first (k quarter w & 3 of the block's 32 weight rows, 16 bytes per lane and 8-k step,
//      straight from global memory into the MFMA B operand registers): their latency hides behind step 1;
//   1. LayerNorm -> Swish of the block's 2 x 32 rows into LDS ([head][row][K + 4], the MFMA A operand): a row per QUARTER wave
//      (16 lanes x K / 16 values, four xor-shuffles per sum instead of six), the wave's eight rows requested before the first
//      reduction, two-pass mean / variance in registers, Swish on the fast exp / rcp pipe like the conv kernels' fused pre-op;
//   2. v_mfma_f32_32x32x2_f32 (exact fp32) over the wave's k quarter: K / 8 dependent MFMAs;
//   3. the four k quarters of a head meet in LDS; mu, logvar (+ bias) and z = mu + eps * exp(0.5 clamp(logvar, -2, 10)) leave as
//      float4 rows, eps drawn on the spot.
// The last block to finish adds the draws' counters to the device-resident Philox counter (every block read it first).
struct HeadsArgs {
  const float* g;          // [2][B][K]
  const float* ln_w[2];
  const float* ln_b[2];
  const float* w2[2];      // [L][K]
  const float* b2[2];
  float* mu;
  float* logvar;
  float* z;
  unsigned long long seed;
  long long* state;
  unsigned* done;
  float ln_eps;
  int B, K, L;
};

template <int KV>  // K = 256 * KV
__global__ __launch_bounds__(512) void encoder_heads_kernel(const HeadsArgs a) {
  constexpr int K = 256 * KV, KP = K + 4, KF = K / 64, NS = K / 32;  // float4 per lane and row; 8-k steps per wave
  extern __shared__ __attribute__((aligned(16))) float hs[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int ntn = a.L >> 5;
  const int rt = blockIdx.x / ntn, n0 = (blockIdx.x - rt * ntn) * 32;
  const unsigned long long base = (a.z && a.state) ? (unsigned long long)a.state[0] : 0ull;
  const bool hd = w >= 4;  // (selects, not indexed loads: an argument array picked by a computed index lives in scratch)
  const float* gh = hd ? a.g + (size_t)a.B * K : a.g;
  const float* lnw = hd ? a.ln_w[1] : a.ln_w[0];
  const float* lnb = hd ? a.ln_b[1] : a.ln_b[0];
  const float* w2 = hd ? a.w2[1] : a.w2[0];

  // ---- 0. weight fragments. MFMA e of step j pairs k = 8 j + e (lanes 0-31) with k = 8 j + 4 + e (lanes 32-63) - the same
  // pairing on both operands, so any order of the k index is the same sum
  const int kb = (w & 3) * (K / 4) + 4 * lh;
  const float* Wp = w2 + (size_t)(n0 + lr) * K + kb;
  f32x4v bw[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) bw[j] = *reinterpret_cast<const f32x4v*>(Wp + 8 * j);

  // ---- 1. LayerNorm + Swish -> LDS
  {
    const int l16 = lane & 15, qr = lane >> 4;
    f32x4v v[2][KF];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int row = 8 * (w & 3) + 4 * bt + qr;
      const float* src = gh + (size_t)min(rt * 32 + row, a.B - 1) * K;
#pragma unroll
      for (int i = 0; i < KF; ++i) v[bt][i] = *reinterpret_cast<const f32x4v*>(src + 64 * i + 4 * l16);
    }
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int row = 8 * (w & 3) + 4 * bt + qr;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < KF; ++i) s += (v[bt][i][0] + v[bt][i][1]) + (v[bt][i][2] + v[bt][i][3]);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float m = s * (1.0f / K);
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < KF; ++i) {
        const float d0 = v[bt][i][0] - m, d1 = v[bt][i][1] - m, d2 = v[bt][i][2] - m, d3 = v[bt][i][3] - m;
        qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
      const float r = 1.0f / sqrtf(qq * (1.0f / K) + a.ln_eps);
      float* dst = hs + ((size_t)(hd ? 32 : 0) + row) * KP + 4 * l16;
#pragma unroll
      for (int i = 0; i < KF; ++i) {
        const f32x4v ww = *reinterpret_cast<const f32x4v*>(lnw + 64 * i + 4 * l16);
        const f32x4v bb = *reinterpret_cast<const f32x4v*>(lnb + 64 * i + 4 * l16);
        f32x4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fl::swish_fast((v[bt][i][e] - m) * r * ww[e] + bb[e]);
        *reinterpret_cast<f32x4v*>(dst + 64 * i) = o;
      }
    }
  }
  __syncthreads();

  // ---- 2. the wave's k quarter of its head's contraction
  const float* Ap = hs + ((size_t)(hd ? 32 : 0) + lr) * KP + kb;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const f32x4v af = *reinterpret_cast<const f32x4v*>(Ap + 8 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bw[j][e], acc, 0, 0, 0);
  }
  __syncthreads();  // every wave has read its A rows: the LDS is free

  // ---- 3. meet: comb[wave][row][33]; C/D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* comb = hs;
#pragma unroll
  for (int r = 0; r < 16; ++r) comb[(w * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + lr] = acc[r];
  __syncthreads();
  const int row = tid >> 3, c4 = (tid & 7) * 4, gr = rt * 32 + row;
  if (tid < 256 && gr < a.B) {
    float mu[4], lv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* c = comb + row * 33 + c4 + e;
      mu[e] = ((c[0 * 32 * 33] + c[1 * 32 * 33]) + (c[2 * 32 * 33] + c[3 * 32 * 33])) + a.b2[0][n0 + c4 + e];
      lv[e] = ((c[4 * 32 * 33] + c[5 * 32 * 33]) + (c[6 * 32 * 33] + c[7 * 32 * 33])) + a.b2[1][n0 + c4 + e];
    }
    const size_t o = (size_t)gr * a.L + n0 + c4;
    if (a.mu) *reinterpret_cast<float4*>(a.mu + o) = make_float4(mu[0], mu[1], mu[2], mu[3]);
    if (a.logvar) *reinterpret_cast<float4*>(a.logvar + o) = make_float4(lv[0], lv[1], lv[2], lv[3]);
    if (a.z) {
      float ev[4], zz[4];
      philox_normal4(a.seed, base + (unsigned long long)(o >> 2), ev);
#pragma unroll
      for (int e = 0; e < 4; ++e) zz[e] = mu[e] + ev[e] * expf(0.5f * fminf(fmaxf(lv[e], -2.0f), 10.0f));  // reparam_kernel's form
      *reinterpret_cast<float4*>(a.z + o) = make_float4(zz[0], zz[1], zz[2], zz[3]);
    }
  }
  if (a.z && a.state && tid == 0) {
    // every block read the counter before anything else; the last one to get here moves it on
    const unsigned prev = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      a.state[0] += ((long long)a.B * a.L + 3) / 4;
      __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// diffuse_prep_kernel: one thread per four latent elements (grid-stride), then the dropout-scale draws.
//   t[b]   = fl_philox_randint at counter base + b
//   noise  = fl_philox_fill(mode 0) at counter base + B + i / 4
//   x_t    = sqrt(ab_t) z + sqrt(1 - ab_t) noise                      (q_sample_kernel's expression)
//   masks  = fl_philox_fill(mode 1, p) of the denoiser's own stream
__global__ __launch_bounds__(256) void diffuse_prep_kernel(const float* __restrict__ z, const float* __restrict__ alpha_bar, int n_steps,
                                                          unsigned long long seed, unsigned long long offset,
                                                          const long long* __restrict__ state, long long* __restrict__ t_out,
                                                          float* __restrict__ noise, float* __restrict__ xt, int B, int L,
                                                          float* __restrict__ masks, long long n_masks, unsigned long long mseed,
                                                          unsigned long long moffset, const long long* __restrict__ mstate, float p) {
  const unsigned long long base = offset + (state ? (unsigned long long)state[0] : 0ull);
  const int L4 = L >> 2;
  const long long nA = (long long)B * L4, nB = masks ? (n_masks + 3) / 4 : 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB; i += (long long)gridDim.x * blockDim.x) {
    if (i < nA) {
      const int b = (int)(i / L4);
      uint32_t r[4];
      philox4(seed, base + (unsigned long long)b, r);
      const long long t = (long long)(((uint64_t)r[0] * (uint64_t)n_steps) >> 32);
      if (i - (long long)b * L4 == 0) t_out[b] = t;
      const float ab = alpha_bar[t];
      float ev[4];
      philox_normal4(seed, base + (unsigned long long)B + (unsigned long long)i, ev);
      const float4 zz = *reinterpret_cast<const float4*>(z + i * 4);
      const float sa = sqrtf(ab), sb = sqrtf(1.f - ab);
      *reinterpret_cast<float4*>(noise + i * 4) = make_float4(ev[0], ev[1], ev[2], ev[3]);
      *reinterpret_cast<float4*>(xt + i * 4) = make_float4(sa * zz.x + sb * ev[0], sa * zz.y + sb * ev[1], sa * zz.z + sb * ev[2], sa * zz.w + sb * ev[3]);
    } else {
      const long long j = i - nA;
      const unsigned long long mb = moffset + (mstate ? (unsigned long long)mstate[0] : 0ull);
      uint32_t r[4];
      philox4(mseed, mb + (unsigned long long)j, r);
      const float keep = 1.f / (1.f - p);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j * 4 + e < n_masks) masks[j * 4 + e] = u01(r[e]) < p ? 0.f : keep;
    }
  }
}

}  // namespace

extern "C" {

int fl_encoder_heads_applicable(int B, int K, int L) { return B > 0 && (K == 256 || K == 512) && L > 0 && L % 32 == 0; }

int fl_encoder_heads_fwd(const float* g, const float* ln_w_mu, const float* ln_b_mu, const float* ln_w_logvar,
                         const float* ln_b_logvar, float ln_eps, const float* w_mu, const float* b_mu, const float* w_logvar,
                         const float* b_logvar, float* mu, float* logvar, float* z, uint64_t seed, int64_t* dev_offset,
                         void* done_counter, int B, int K, int L, fl_stream_t stream) {
  FL_REQUIRE(fl_encoder_heads_applicable(B, K, L), "fl_encoder_heads_fwd: unsupported shape B=%d K=%d (256 or 512) L=%d (multiple of 32)", B, K, L);
  FL_REQUIRE(g && ln_w_mu && ln_b_mu && ln_w_logvar && ln_b_logvar && w_mu && b_mu && w_logvar && b_logvar, "fl_encoder_heads_fwd: null operand");
  FL_REQUIRE(mu || logvar || z, "fl_encoder_heads_fwd: no output requested");
  FL_REQUIRE(!z || !dev_offset || done_counter, "fl_encoder_heads_fwd: a device-resident counter needs the done_counter word");
  HeadsArgs a;
  a.g = g;
  a.ln_w[0] = ln_w_mu; a.ln_b[0] = ln_b_mu; a.ln_w[1] = ln_w_logvar; a.ln_b[1] = ln_b_logvar;
  a.w2[0] = w_mu; a.b2[0] = b_mu; a.w2[1] = w_logvar; a.b2[1] = b_logvar;
  a.mu = mu; a.logvar = logvar; a.z = z;
  a.seed = seed;
  a.state = reinterpret_cast<long long*>(dev_offset);
  a.done = reinterpret_cast<unsigned*>(done_counter);
  a.ln_eps = ln_eps;
  a.B = B; a.K = K; a.L = L;
  const int blocks = fl::cdiv(B, 32) * (L / 32);
  const size_t lds = (size_t)2 * 32 * (K + 4) * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (K == 512) {
    static bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_heads_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)attr;
    hipLaunchKernelGGL(encoder_heads_kernel<2>, dim3(blocks), dim3(512), lds, s, a);
  } else {
    static bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_heads_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)attr;
    hipLaunchKernelGGL(encoder_heads_kernel<1>, dim3(blocks), dim3(512), lds, s, a);
  }
  return fl::check_launch("encoder_heads_kernel");
}

int fl_diffuse_prep(const float* z, const float* alpha_bar, int n_steps, uint64_t seed, uint64_t offset, const int64_t* dev_offset,
                    int64_t* t, float* noise, float* xt, int B, int L, float* masks, int64_t n_masks, uint64_t mask_seed,
                    uint64_t mask_offset, const int64_t* mask_dev_offset, float p, fl_stream_t stream) {
  FL_REQUIRE(z && alpha_bar && t && noise && xt && B > 0 && L > 0 && L % 4 == 0 && n_steps > 0, "fl_diffuse_prep: bad args (B=%d, L=%d)", B, L);
  FL_REQUIRE(!masks || (n_masks > 0 && p >= 0.f && p < 1.f), "fl_diffuse_prep: bad dropout draw (n=%lld, p=%g)", (long long)n_masks, (double)p);
  const long long work = (long long)B * (L / 4) + (masks ? (n_masks + 3) / 4 : 0);
  int blocks = fl::cdiv(work, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(diffuse_prep_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z, alpha_bar, n_steps,
                     (unsigned long long)seed, (unsigned long long)offset, reinterpret_cast<const long long*>(dev_offset),
                     reinterpret_cast<long long*>(t), noise, xt, B, L, masks, (long long)n_masks, (unsigned long long)mask_seed,
                     (unsigned long long)mask_offset, reinterpret_cast<const long long*>(mask_dev_offset), p);
  return fl::check_launch("diffuse_prep_kernel");
}

}  // extern "C"
