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

}  // namespace

extern "C" {

int fl_attention_fwd(const float* qkv, float* out, float* P, int B, int HW, int C, int heads, float scale, fl_stream_t stream) {
  FL_REQUIRE(qkv && out && B >= 0 && HW > 0 && C > 0 && heads > 0, "fl_attention_fwd: bad arguments");
  FL_REQUIRE(C % heads == 0 && HW % 32 == 0, "fl_attention_fwd: C %% heads == 0 and HW %% 32 == 0 required (C=%d heads=%d HW=%d)", C, heads, HW);
  const int d = C / heads;
  FL_REQUIRE(d <= 32, "fl_attention_fwd: head dim %d > 32", d);
  if (B == 0) return FL_OK;
  AttnArgs g;
  g.qkv = qkv; g.out = out; g.P = P; g.B = B; g.HW = HW; g.C = C; g.nh = heads; g.d = d; g.scale = scale;
  dim3 grid((unsigned)fl::cdiv(HW, 128), (unsigned)heads, (unsigned)B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d <= 16) hipLaunchKernelGGL(attention_fwd_kernel<16>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(attention_fwd_kernel<32>, grid, dim3(256), 0, s, g);
  return fl::check_launch("attention_fwd_kernel");
}

}  // extern "C"
