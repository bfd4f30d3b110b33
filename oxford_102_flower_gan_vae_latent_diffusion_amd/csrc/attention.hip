// Fused multi-head self-attention over the h*w tokens of a feature map: the attention core of v2's UNetAttentionBlock
// (v2:446-457) - softmax(scale * Q K^T) V per (image, head) in ONE launch, no (HW x HW) score tensor in HBM.
//
// qkv: (B, HW, 3C) fp32, channel = which*C + head*d + i (the 1x1 qkv conv's NHWC output, reshape(b,3,heads,d,hw) of
// v2:447). out: (B, HW, C) with channel = i*heads + head - the d-major re-ordering `out.permute(0,3,1,2).reshape(b,c,h,w)`
// of v2:456-457 produces. P (optional, training): (B, heads, HW, HW) normalised attention weights for the backward.
//
// One wave owns 32 queries. Scores are built TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x2_f32: exact fp32) so that a query
// is a lane and its keys sit in that lane's 16 accumulator registers: the row softmax is in-lane arithmetic plus ONE
// cross-half shuffle per statistic, and the accumulator tile is - as it stands, no lane movement, no LDS - the B operand of
// the P V contraction (O^T = V^T P^T; register r of lane l holds key (r&3)+8(r>>2)+4(l>>5) of query l&31, exactly the
// (k slot, column) a 32x32x2 MFMA wants). Two passes over the keys: (1) running max / sum, (2) exact normalised
// probabilities -> P V (and the optional P store): Q K^T is d/2 MFMAs per tile, recomputing it is cheaper than rescaling.
#include "fl_common.h"

namespace {

struct AttnArgs {
  const float* qkv;
  float* out;
  float* P;
  float* stats;  // optional (B, heads, HW, 2): row max of scale * s and sum of exp - what the backward recomputes P from
  int B, HW, C, nh, d;
  float scale;
};

template <int DH>  // head dim padded to 16 or 32 (DH / 2 MFMAs per score tile)
__global__ __launch_bounds__(256) void attention_fwd_kernel(const AttnArgs g) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane & 31, ks = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  if (q0 >= g.HW) return;  // whole wave
  const int h = blockIdx.y, b = blockIdx.z;
  const int C3 = 3 * g.C, d = g.d;
  const float* base = g.qkv + (size_t)b * g.HW * C3 + h * d;
  const float* Qr = base + (size_t)(q0 + lq) * C3;  // this lane's query row
  // Q fragments: B operand of S^T = K Q^T: lane (col q, k slot ks) supplies Q[q][2j + ks]
  float qf[DH / 2];
#pragma unroll
  for (int j = 0; j < DH / 2; ++j) qf[j] = (2 * j + ks < d) ? Qr[2 * j + ks] : 0.f;

  const int ntile = g.HW / 32;
  auto scores = [&](int kt, f32x16& s) {
    const float* Kr = base + g.C + (size_t)(kt * 32 + lq) * C3;  // A operand: lane (row key, k slot ks) supplies K[key][2j + ks]
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int j = 0; j < DH / 2; ++j) {
      const float kv = (2 * j + ks < d) ? Kr[2 * j + ks] : 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qf[j], s, 0, 0, 0);
    }
  };

  // ---- pass 1: row max and sum of exp (per query = per lane pair (l, l ^ 32))
  float m = -INFINITY, l = 0.f;
  for (int kt = 0; kt < ntile; ++kt) {
    f32x16 s;
    scores(kt, s);
    float mt = s[0] * g.scale;
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e] * g.scale);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += expf(s[e] * g.scale - mn);
    acc += __shfl_xor(acc, 32, 64);
    l = l * expf(m - mn) + acc;
    m = mn;
  }
  const float inv_l = 1.0f / l;
  if (g.stats && ks == 0) {
    float* st = g.stats + (((size_t)b * g.nh + h) * g.HW + q0 + lq) * 2;
    st[0] = m;
    st[1] = l;
  }

  // ---- pass 2: P = exp(scale s - m) / l, O^T += V^T P^T
  f32x16 o;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  const float* Vb = base + 2 * g.C;
  for (int kt = 0; kt < ntile; ++kt) {
    f32x16 s;
    scores(kt, s);
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = expf(s[e] * g.scale - m) * inv_l;
    if (g.P) {
      float* Pq = g.P + (((size_t)b * g.nh + h) * g.HW + (q0 + lq)) * g.HW + kt * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) Pq[(e & 3) + 8 * (e >> 2) + 4 * ks] = s[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * ks;   // the key register e of this lane half holds
      const float vv = (lq < d) ? Vb[(size_t)key * C3 + lq] : 0.f;  // A operand: lane (row i = lq, k slot ks) = V[key][i]
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[e], o, 0, 0, 0);
    }
  }
  // ---- O^T tile: row i = (e&3) + 8 (e>>2) + 4 ks, column = query lq -> out[b, q, i * heads + h]
  float* Oq = g.out + ((size_t)b * g.HW + q0 + lq) * g.C + h;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * ks;
    if (i < d) Oq[(size_t)i * g.nh] = o[e];
  }
}


// ---------------------------------------------------------------------------------------------------------------- forward, one pass
// The same contraction in ONE pass over the keys (online softmax), on v_mfma_f32_16x16x4_f32 and with K / V staged in LDS once per
// block (round 2's kernel above: two passes, every wave fetching its own K / V fragments from global memory, V^T padded to 32
// rows - at d = 16 half of its P V MFMAs multiplied zeros - and the accurate expf: 0.377 ms = 14.5 % of the fp32 MFMA peak).
//   * block = 512 threads = 8 waves x 32 queries of one (image, head); the keys go through LDS in chunks of 256: K as it lies
//     ([key][DH]), V TRANSPOSED ([dim][256 + 4]) so that every MFMA operand below is one conflict-free ds_read_b128;
//   * 16 x 16 tiles: S^T = K Q^T puts query lane & 15 in a lane and keys 4 (lane >> 4) + r in its four accumulator registers;
//     register r - after exp - IS the B operand of the P V step that contracts keys {4 g + r}: no lane movement, no LDS, and
//     V^T has exactly d rows (no padding work for d = 16). The contraction index of S is walked as dim = 16 jj + 4 g + e so that
//     a lane's four steps read ONE float4 of K and of Q;
//   * softmax per 64 keys: running max (two xor-shuffles among the 4 lanes of a query), p = exp2(s' - m') on the fast
//     transcendental pipe with scale * log2(e) folded into Q, accumulators rescaled by exp2(m_old - m_new), the row sums kept
//     per lane and added up once at the end. stats = (m' ln 2, l): what fl_attention_bwd recomputes P from.
template <int DH>
__global__ __launch_bounds__(512, (DH <= 32 ? 4 : 2)) void attention_fwd_onepass_kernel(const AttnArgs g) {
  constexpr int KC = 256, NJ = DH / 16, PV = KC + 4;
  constexpr int KT = DH <= 32 ? 4 : 2;  // key tiles (of 16) per softmax group: bounded by the fragment registers
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float at_smem[];
  float* Ks = at_smem;
  float* Vt = at_smem + KC * DH;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g4 = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 256 + 32 * w;
  const bool active = q0 < g.HW;  // wave-uniform; HW % 32 == 0: all 32 queries exist
  const int C3 = 3 * g.C, d = g.d;
  const float* base = g.qkv + (size_t)b * g.HW * C3 + h * d;
  const float qs = g.scale * 1.4426950408889634f;

  f32x4a qf[2][NJ];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int dim = 16 * jj + 4 * g4;
      f32x4a v = {0.f, 0.f, 0.f, 0.f};
      if (active && dim < d) v = *reinterpret_cast<const f32x4a*>(base + (size_t)(q0 + 16 * qt + ln) * C3 + dim);
      qf[qt][jj] = v * qs;
    }
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f32x4a o[2][NJ];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int mt = 0; mt < NJ; ++mt) o[qt][mt] = f32x4a{0.f, 0.f, 0.f, 0.f};

  for (int kc = 0; kc < g.HW; kc += KC) {
    const int nk = g.HW - kc < KC ? g.HW - kc : KC;
    __syncthreads();  // the previous chunk has been consumed
    for (int idx = tid; idx < nk * (DH / 4); idx += 512) {
      const int key = idx / (DH / 4), c4 = idx - key * (DH / 4);
      f32x4a kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (4 * c4 < d) {
        const float* src = base + (size_t)(kc + key) * C3 + 4 * c4;
        kv = *reinterpret_cast<const f32x4a*>(src + g.C);
        vv = *reinterpret_cast<const f32x4a*>(src + 2 * g.C);
      }
      *reinterpret_cast<f32x4a*>(Ks + key * DH + 4 * c4) = kv;
#pragma unroll
      for (int e = 0; e < 4; ++e) Vt[(4 * c4 + e) * PV + key] = vv[e];
    }
    __syncthreads();
    if (!active) continue;
    for (int kg = 0; kg < nk; kg += 16 * KT) {
      f32x4a kf[KT][NJ], vf[KT][NJ];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          kf[kt][jj] = *reinterpret_cast<const f32x4a*>(Ks + (kg + 16 * kt + ln) * DH + 16 * jj + 4 * g4);
          vf[kt][jj] = *reinterpret_cast<const f32x4a*>(Vt + (16 * jj + ln) * PV + kg + 16 * kt + 4 * g4);
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4a sc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          sc[kt] = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][jj][e], qf[qt][jj][e], sc[kt], 0, 0, 0);
        }
        float mx = sc[0][0];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);
        m[qt] = mn;
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sc[kt][r] = __builtin_amdgcn_exp2f(sc[kt][r] - mn);
            ps += sc[kt][r];
          }
        l[qt] = l[qt] * alpha + ps;
#pragma unroll
        for (int mt = 0; mt < NJ; ++mt) o[qt][mt] = o[qt][mt] * alpha;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < NJ; ++mt) o[qt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][mt][r], sc[kt][r], o[qt][mt], 0, 0, 0);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    const int q = q0 + 16 * qt + ln;
    if (g.stats && g4 == 0) {
      float* st = g.stats + (((size_t)b * g.nh + h) * g.HW + q) * 2;
      st[0] = m[qt] * 0.6931471805599453f;
      st[1] = lt;
    }
    // O^T tile: row i = 16 mt + 4 g4 + r, column = query ln -> out[b, q, i * heads + h]
    float* Oq = g.out + ((size_t)b * g.HW + q) * g.C + h;
#pragma unroll
    for (int mt = 0; mt < NJ; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * mt + 4 * g4 + r;
        if (i < d) Oq[(size_t)i * g.nh] = o[qt][mt][r] * inv;
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------- backward
// dQ, dK, dV of the attention core in two launches, no (HW x HW) tensor in HBM and no atomics: the probabilities are
// recomputed from Q, K and the forward's row statistics (m = max of scale*s, l = sum of exp).
//   dV = P^T dO;  dP = dO V^T;  D_q = sum_k P_qk dP_qk = <dO_q, O_q>;  dS = scale * P o (dP - D);  dQ = dS K;  dK = dS^T Q
// dmerged / merged use the forward's d-major channel order (channel = i * heads + head), dqkv the qkv layout.
struct AttnBwdArgs {
  const float* qkv;
  const float* dmerged;
  const float* merged;
  const float* stats;
  float* dqkv;
  float* D;  // (B, heads, HW): written by the dQ kernel, read by the dK / dV kernel
  int B, HW, C, nh, d;
  float scale;
};

// One wave = 32 queries (a lane is a query, as in the forward): S^T and dP^T tiles (keys x queries) per key tile, dS^T in
// lane, and the accumulator tile is the B operand of dQ^T += K^T dS^T as it stands.
template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(const AttnBwdArgs g) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane & 31, ks = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wid) * 32;
  if (q0 >= g.HW) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int C3 = 3 * g.C, d = g.d, nh = g.nh;
  const float* base = g.qkv + (size_t)b * g.HW * C3 + h * d;
  const float* Qr = base + (size_t)(q0 + lq) * C3;
  const float* dOr = g.dmerged + ((size_t)b * g.HW + q0 + lq) * g.C + h;  // dO[q][i] = dOr[i * nh]
  const float* Or = g.merged + ((size_t)b * g.HW + q0 + lq) * g.C + h;
  float qf[DH / 2], dof[DH / 2];
  float Dq = 0.f;
#pragma unroll
  for (int j = 0; j < DH / 2; ++j) {
    const int i = 2 * j + ks;
    qf[j] = i < d ? Qr[i] : 0.f;
    dof[j] = i < d ? dOr[(size_t)i * nh] : 0.f;
    Dq += i < d ? dof[j] * Or[(size_t)i * nh] : 0.f;
  }
  Dq += __shfl_xor(Dq, 32, 64);
  const float* st = g.stats + (((size_t)b * nh + h) * g.HW + q0 + lq) * 2;
  const float m = st[0], inv_l = 1.0f / st[1];
  if (ks == 0) g.D[((size_t)b * nh + h) * g.HW + q0 + lq] = Dq;
  f32x16 o;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  const float* Kb = base + g.C;
  const float* Vb = base + 2 * g.C;
  const int ntile = g.HW / 32;
  for (int kt = 0; kt < ntile; ++kt) {
    const float* Kr = Kb + (size_t)(kt * 32 + lq) * C3;
    const float* Vr = Vb + (size_t)(kt * 32 + lq) * C3;
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
#pragma unroll
    for (int j = 0; j < DH / 2; ++j) {
      const int i = 2 * j + ks;
      const float kv = i < d ? Kr[i] : 0.f, vv = i < d ? Vr[i] : 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qf[j], s, 0, 0, 0);     // S^T[key][q]
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, dof[j], dp, 0, 0, 0);  // dP^T[key][q] = <V_key, dO_q>
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = expf(s[e] * g.scale - m) * inv_l * (dp[e] - Dq) * g.scale;  // dS^T
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * ks;
      const float kk = lq < d ? Kb[(size_t)key * C3 + lq] : 0.f;  // A operand: lane (row i = lq, k slot ks) = K[key][i]
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(kk, s[e], o, 0, 0, 0);
    }
  }
  float* dQ = g.dqkv + ((size_t)b * g.HW + q0 + lq) * C3 + h * d;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * ks;
    if (i < d) dQ[i] = o[e];
  }
}

// One wave = 32 keys (a lane is a key): S and dP tiles (queries x keys) per query tile; P and dS tiles are the B operands of
// dV^T += dO^T P and dK^T += Q^T dS as they stand.
template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(const AttnBwdArgs g) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane & 31, ks = lane >> 5;
  const int k0 = (blockIdx.x * 4 + wid) * 32;
  if (k0 >= g.HW) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int C3 = 3 * g.C, d = g.d, nh = g.nh;
  const float* base = g.qkv + (size_t)b * g.HW * C3 + h * d;
  const float* Kr = base + g.C + (size_t)(k0 + lq) * C3;
  const float* Vr = base + 2 * g.C + (size_t)(k0 + lq) * C3;
  float kf[DH / 2], vf[DH / 2];
#pragma unroll
  for (int j = 0; j < DH / 2; ++j) {
    const int i = 2 * j + ks;
    kf[j] = i < d ? Kr[i] : 0.f;
    vf[j] = i < d ? Vr[i] : 0.f;
  }
  f32x16 dk, dv;
#pragma unroll
  for (int e = 0; e < 16; ++e) dk[e] = dv[e] = 0.f;
  const float* dOb = g.dmerged + (size_t)b * g.HW * g.C + h;
  const float* stb = g.stats + ((size_t)b * nh + h) * g.HW * 2;
  const float* Db = g.D + ((size_t)b * nh + h) * g.HW;
  const int ntile = g.HW / 32;
  for (int qt = 0; qt < ntile; ++qt) {
    const float* Qr = base + (size_t)(qt * 32 + lq) * C3;          // A operand rows: this lane's query of the tile
    const float* dOr = dOb + (size_t)(qt * 32 + lq) * g.C;
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
#pragma unroll
    for (int j = 0; j < DH / 2; ++j) {
      const int i = 2 * j + ks;
      const float qv = i < d ? Qr[i] : 0.f, gv = i < d ? dOr[(size_t)i * nh] : 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, kf[j], s, 0, 0, 0);    // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, vf[j], dp, 0, 0, 0);  // dP[q][key]
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = qt * 32 + (e & 3) + 8 * (e >> 2) + 4 * ks;  // the query row register e of this lane half holds
      const float p = expf(s[e] * g.scale - stb[2 * q]) / stb[2 * q + 1];
      dp[e] = p * (dp[e] - Db[q]) * g.scale;  // dS[q][key]
      s[e] = p;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = qt * 32 + (e & 3) + 8 * (e >> 2) + 4 * ks;
      const float gq = lq < d ? dOb[(size_t)q * g.C + (size_t)lq * nh] : 0.f;  // A operand: lane (row i = lq, k slot ks) = dO[q][i]
      const float qq = lq < d ? base[(size_t)q * C3 + lq] : 0.f;               //                                        = Q[q][i]
      dv = __builtin_amdgcn_mfma_f32_32x32x2f32(gq, s[e], dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qq, dp[e], dk, 0, 0, 0);
    }
  }
  float* dK = g.dqkv + ((size_t)b * g.HW + k0 + lq) * C3 + g.C + h * d;
  float* dV = dK + g.C;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * ks;
    if (i < d) {
      dK[i] = dk[e];
      dV[i] = dv[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- backward, LDS-staged
// The two backward kernels rebuilt like attention_fwd_onepass_kernel: v_mfma_f32_16x16x4_f32, every operand a ds_read_b128 out
// of LDS images staged once per block (row-major images with a 4-float pad: conflict-free; the transposed ones [dim][256 + 4]),
// exp2 on the transcendental pipe, and every accumulator tile used AS IT STANDS as the B operand of the next contraction.
//   dQ kernel (a lane = a query, 8 waves x 32 queries of one (image, head), keys through LDS in chunks of 256):
//     S^T = K Q^T,  P = exp2(s' - m') / l,  dP^T = V dO^T,  dS^T = scale P (dP^T - D),  dQ^T += K^T dS^T;  also writes D = <dO, O>
//   dK / dV kernel (a lane = a key, queries through LDS):
//     S = Q K^T,  P,  dP = dO V^T,  dS;  dV^T += dO^T P,  dK^T += Q^T dS
template <int DH>
__global__ __launch_bounds__(512, 2) void attention_bwd_dq_lds_kernel(const AttnBwdArgs g) {
  constexpr int KC = 256, NJ = DH / 16, PR = DH + 4, PT = KC + 4, KT = 2;
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float at_smem[];
  float* Ks = at_smem;              // [key][PR]
  float* Vs = Ks + KC * PR;         // [key][PR]
  float* Kt = Vs + KC * PR;         // [dim][PT]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g4 = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, nh = g.nh;
  const int q0 = blockIdx.x * 256 + 32 * w;
  const bool active = q0 < g.HW;
  const int C3 = 3 * g.C, d = g.d;
  const float* base = g.qkv + (size_t)b * g.HW * C3 + h * d;
  const float qs = g.scale * 1.4426950408889634f;

  f32x4a qf[2][NJ], dof[2][NJ], dq[2][NJ];
  float mq[2], il[2], Dq[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + 16 * qt + ln;
    float part = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int dim = 16 * jj + 4 * g4;
      f32x4a v = {0.f, 0.f, 0.f, 0.f}, go = {0.f, 0.f, 0.f, 0.f};
      if (active && dim < d) {
        v = *reinterpret_cast<const f32x4a*>(base + (size_t)q * C3 + dim);
        const float* dOr = g.dmerged + ((size_t)b * g.HW + q) * g.C + h;
        const float* Or = g.merged + ((size_t)b * g.HW + q) * g.C + h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          go[e] = dOr[(size_t)(dim + e) * nh];
          part += go[e] * Or[(size_t)(dim + e) * nh];
        }
      }
      qf[qt][jj] = v * qs;
      dof[qt][jj] = go;
      dq[qt][jj] = f32x4a{0.f, 0.f, 0.f, 0.f};
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    Dq[qt] = part;
    mq[qt] = 0.f;
    il[qt] = 0.f;
    if (active) {
      const float* st = g.stats + (((size_t)b * nh + h) * g.HW + q) * 2;
      mq[qt] = st[0] * 1.4426950408889634f;
      il[qt] = 1.0f / st[1];
      if (g4 == 0) g.D[((size_t)b * nh + h) * g.HW + q] = part;
    }
  }

  for (int kc = 0; kc < g.HW; kc += KC) {
    const int nk = g.HW - kc < KC ? g.HW - kc : KC;
    __syncthreads();
    for (int idx = tid; idx < nk * (DH / 4); idx += 512) {
      const int key = idx / (DH / 4), c4 = idx - key * (DH / 4);
      f32x4a kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (4 * c4 < d) {
        const float* src = base + (size_t)(kc + key) * C3 + 4 * c4;
        kv = *reinterpret_cast<const f32x4a*>(src + g.C);
        vv = *reinterpret_cast<const f32x4a*>(src + 2 * g.C);
      }
      *reinterpret_cast<f32x4a*>(Ks + key * PR + 4 * c4) = kv;
      *reinterpret_cast<f32x4a*>(Vs + key * PR + 4 * c4) = vv;
#pragma unroll
      for (int e = 0; e < 4; ++e) Kt[(4 * c4 + e) * PT + key] = kv[e];
    }
    __syncthreads();
    if (!active) continue;
    for (int kg = 0; kg < nk; kg += 16 * KT) {
      f32x4a kf[KT][NJ], vf[KT][NJ], tf[KT][NJ];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          kf[kt][jj] = *reinterpret_cast<const f32x4a*>(Ks + (kg + 16 * kt + ln) * PR + 16 * jj + 4 * g4);
          vf[kt][jj] = *reinterpret_cast<const f32x4a*>(Vs + (kg + 16 * kt + ln) * PR + 16 * jj + 4 * g4);
          tf[kt][jj] = *reinterpret_cast<const f32x4a*>(Kt + (16 * jj + ln) * PT + kg + 16 * kt + 4 * g4);
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          f32x4a sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][jj][e], qf[qt][jj][e], sc, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][jj][e], dof[qt][jj][e], dp, 0, 0, 0);
            }
          f32x4a ds;
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(sc[r] - mq[qt]) * il[qt] * (dp[r] - Dq[qt]) * g.scale;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < NJ; ++mt) dq[qt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(tf[kt][mt][r], ds[r], dq[qt][mt], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float* dQ = g.dqkv + ((size_t)b * g.HW + q0 + 16 * qt + ln) * C3 + h * d;
#pragma unroll
    for (int mt = 0; mt < NJ; ++mt) {
      const int i = 16 * mt + 4 * g4;
      if (i < d) *reinterpret_cast<f32x4a*>(dQ + i) = dq[qt][mt];  // d % 4 == 0: a lane's four rows exist together
    }
  }
}

template <int DH>
__global__ __launch_bounds__(512, 2) void attention_bwd_dkv_lds_kernel(const AttnBwdArgs g) {
  constexpr int QC = 256, NJ = DH / 16, PR = DH + 4, PT = QC + 4;
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float at_smem[];
  float* Qs = at_smem;               // [query][PR], scaled by scale * log2(e)
  float* Gs = Qs + QC * PR;          // dO [query][PR]
  float* Qt = Gs + QC * PR;          // Q^T [dim][PT] (unscaled)
  float* Gt = Qt + DH * PT;          // dO^T [dim][PT]
  float* Ms = Gt + DH * PT;          // m' | 1 / l | D, 256 each
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g4 = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, nh = g.nh;
  const int k0 = blockIdx.x * 256 + 32 * w;
  const bool active = k0 < g.HW;
  const int C3 = 3 * g.C, d = g.d;
  const float* base = g.qkv + (size_t)b * g.HW * C3 + h * d;
  const float qs = g.scale * 1.4426950408889634f;

  f32x4a kf[2][NJ], vf[2][NJ], dk[2][NJ], dv[2][NJ];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int dim = 16 * jj + 4 * g4;
      f32x4a kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (active && dim < d) {
        const float* src = base + (size_t)(k0 + 16 * kt + ln) * C3 + dim;
        kv = *reinterpret_cast<const f32x4a*>(src + g.C);
        vv = *reinterpret_cast<const f32x4a*>(src + 2 * g.C);
      }
      kf[kt][jj] = kv;
      vf[kt][jj] = vv;
      dk[kt][jj] = f32x4a{0.f, 0.f, 0.f, 0.f};
      dv[kt][jj] = f32x4a{0.f, 0.f, 0.f, 0.f};
    }

  for (int qc = 0; qc < g.HW; qc += QC) {
    const int nq = g.HW - qc < QC ? g.HW - qc : QC;
    __syncthreads();
    for (int idx = tid; idx < nq * (DH / 4); idx += 512) {
      const int q = idx / (DH / 4), c4 = idx - q * (DH / 4);
      f32x4a qv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
      if (4 * c4 < d) {
        qv = *reinterpret_cast<const f32x4a*>(base + (size_t)(qc + q) * C3 + 4 * c4);
        const float* dOr = g.dmerged + ((size_t)b * g.HW + qc + q) * g.C + h;
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = dOr[(size_t)(4 * c4 + e) * nh];
      }
      *reinterpret_cast<f32x4a*>(Qs + q * PR + 4 * c4) = qv * qs;
      *reinterpret_cast<f32x4a*>(Gs + q * PR + 4 * c4) = gv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Qt[(4 * c4 + e) * PT + q] = qv[e];
        Gt[(4 * c4 + e) * PT + q] = gv[e];
      }
    }
    for (int q = tid; q < nq; q += 512) {
      const float* st = g.stats + (((size_t)b * nh + h) * g.HW + qc + q) * 2;
      Ms[q] = st[0] * 1.4426950408889634f;
      Ms[256 + q] = 1.0f / st[1];
      Ms[512 + q] = g.D[((size_t)b * nh + h) * g.HW + qc + q];
    }
    __syncthreads();
    if (!active) continue;
    for (int qb = 0; qb < nq; qb += 16) {
      f32x4a qa[NJ], ga[NJ], qtf[NJ], gtf[NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        qa[jj] = *reinterpret_cast<const f32x4a*>(Qs + (qb + ln) * PR + 16 * jj + 4 * g4);
        ga[jj] = *reinterpret_cast<const f32x4a*>(Gs + (qb + ln) * PR + 16 * jj + 4 * g4);
        qtf[jj] = *reinterpret_cast<const f32x4a*>(Qt + (16 * jj + ln) * PT + qb + 4 * g4);
        gtf[jj] = *reinterpret_cast<const f32x4a*>(Gt + (16 * jj + ln) * PT + qb + 4 * g4);
      }
      const f32x4a m4 = *reinterpret_cast<const f32x4a*>(Ms + qb + 4 * g4);          // statistics of queries 4 g4 + r
      const f32x4a l4 = *reinterpret_cast<const f32x4a*>(Ms + 256 + qb + 4 * g4);
      const f32x4a D4 = *reinterpret_cast<const f32x4a*>(Ms + 512 + qb + 4 * g4);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x4a sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};   // [query 4 g4 + r][key ln]
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[jj][e], kf[kt][jj][e], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[jj][e], vf[kt][jj][e], dp, 0, 0, 0);
          }
        f32x4a pp, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pp[r] = __builtin_amdgcn_exp2f(sc[r] - m4[r]) * l4[r];
          ds[r] = pp[r] * (dp[r] - D4[r]) * g.scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mt = 0; mt < NJ; ++mt) {
            dv[kt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gtf[mt][r], pp[r], dv[kt][mt], 0, 0, 0);
            dk[kt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qtf[mt][r], ds[r], dk[kt][mt], 0, 0, 0);
          }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    float* dK = g.dqkv + ((size_t)b * g.HW + k0 + 16 * kt + ln) * C3 + g.C + h * d;
#pragma unroll
    for (int mt = 0; mt < NJ; ++mt) {
      const int i = 16 * mt + 4 * g4;
      if (i < d) {
        *reinterpret_cast<f32x4a*>(dK + i) = dk[kt][mt];
        *reinterpret_cast<f32x4a*>(dK + g.C + i) = dv[kt][mt];
      }
    }
  }
}

}  // namespace

extern "C" {

int fl_attention_fwd(const float* qkv, float* out, float* P, float* stats, int B, int HW, int C, int heads, float scale,
                     fl_stream_t stream) {
  FL_REQUIRE(qkv && out && B >= 0 && HW > 0 && C > 0 && heads > 0, "fl_attention_fwd: bad arguments");
  FL_REQUIRE(C % heads == 0 && HW % 32 == 0, "fl_attention_fwd: C %% heads == 0 and HW %% 32 == 0 required (C=%d heads=%d HW=%d)", C, heads, HW);
  const int d = C / heads;
  // one-pass kernel: no P output, head dim a multiple of 4 up to 64, key groups of 64; anything else: the two-pass kernel (d <= 32)
  const bool onepass = !P && d % 4 == 0 && d <= 64 && HW % 64 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0;
  FL_REQUIRE(onepass || d <= 32, "fl_attention_fwd: head dim %d > 32 needs HW %% 64 == 0, d %% 4 == 0, d <= 64 and no P output", d);
  if (B == 0) return FL_OK;
  AttnArgs g;
  g.qkv = qkv; g.out = out; g.P = P; g.stats = stats; g.B = B; g.HW = HW; g.C = C; g.nh = heads; g.d = d; g.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (onepass) {
    dim3 grid1((unsigned)fl::cdiv(HW, 256), (unsigned)heads, (unsigned)B);
    const int DH = d <= 16 ? 16 : (d <= 32 ? 32 : 64);
    const size_t lds = (size_t)(256 * DH + DH * 260) * sizeof(float);
    if (DH == 16) hipLaunchKernelGGL(attention_fwd_onepass_kernel<16>, grid1, dim3(512), lds, s, g);
    else if (DH == 32) {
      static bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_onepass_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
      (void)attr;
      hipLaunchKernelGGL(attention_fwd_onepass_kernel<32>, grid1, dim3(512), lds, s, g);
    } else {
      static bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_onepass_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
      (void)attr;
      hipLaunchKernelGGL(attention_fwd_onepass_kernel<64>, grid1, dim3(512), lds, s, g);
    }
    return fl::check_launch("attention_fwd_onepass_kernel");
  }
  dim3 grid((unsigned)fl::cdiv(HW, 128), (unsigned)heads, (unsigned)B);
  if (d <= 16) hipLaunchKernelGGL(attention_fwd_kernel<16>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(attention_fwd_kernel<32>, grid, dim3(256), 0, s, g);
  return fl::check_launch("attention_fwd_kernel");
}

size_t fl_attention_bwd_workspace_bytes(int B, int HW, int heads) { return (size_t)B * heads * HW * sizeof(float); }

int fl_attention_bwd(const float* qkv, const float* merged, const float* dmerged, const float* stats, float* dqkv, int B, int HW,
                     int C, int heads, float scale, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(qkv && merged && dmerged && stats && dqkv && B >= 0 && HW > 0 && C > 0 && heads > 0, "fl_attention_bwd: bad arguments");
  FL_REQUIRE(C % heads == 0 && HW % 32 == 0 && C / heads <= 32, "fl_attention_bwd: C %% heads == 0, HW %% 32 == 0, C / heads <= 32 (C=%d heads=%d HW=%d)",
             C, heads, HW);
  if (B == 0) return FL_OK;
  FL_REQUIRE(ws && ws_bytes >= fl_attention_bwd_workspace_bytes(B, HW, heads), "fl_attention_bwd: workspace too small");
  AttnBwdArgs g;
  g.qkv = qkv; g.dmerged = dmerged; g.merged = merged; g.stats = stats; g.dqkv = dqkv; g.D = reinterpret_cast<float*>(ws);
  g.B = B; g.HW = HW; g.C = C; g.nh = heads; g.d = C / heads; g.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (g.d % 4 == 0 && HW % 64 == 0 && ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dqkv)) & 15) == 0) {
    // LDS-staged 16x16x4 kernels (the same contractions; what the shape rules of the one-pass forward allow)
    dim3 grid1((unsigned)fl::cdiv(HW, 256), (unsigned)heads, (unsigned)B);
    const int DH = g.d <= 16 ? 16 : 32;
    const size_t lds_q = (size_t)(2 * 256 * (DH + 4) + DH * 260) * sizeof(float);
    const size_t lds_k = (size_t)(2 * 256 * (DH + 4) + 2 * DH * 260 + 768) * sizeof(float);
    static bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dq_lds_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dkv_lds_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dkv_lds_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)attr;
    if (DH == 16) {
      hipLaunchKernelGGL(attention_bwd_dq_lds_kernel<16>, grid1, dim3(512), lds_q, s, g);
      hipLaunchKernelGGL(attention_bwd_dkv_lds_kernel<16>, grid1, dim3(512), lds_k, s, g);
    } else {
      hipLaunchKernelGGL(attention_bwd_dq_lds_kernel<32>, grid1, dim3(512), lds_q, s, g);
      hipLaunchKernelGGL(attention_bwd_dkv_lds_kernel<32>, grid1, dim3(512), lds_k, s, g);
    }
    return fl::check_launch("attention_bwd_lds_kernels");
  }
  dim3 grid((unsigned)fl::cdiv(HW, 128), (unsigned)heads, (unsigned)B);
  if (g.d <= 16) {
    hipLaunchKernelGGL(attention_bwd_dq_kernel<16>, grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<16>, grid, dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL(attention_bwd_dq_kernel<32>, grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<32>, grid, dim3(256), 0, s, g);
  }
  return fl::check_launch("attention_bwd_kernels");
}

}  // extern "C"
