// First convolution of the encoder / VGG16: nn.Conv2d(3, 64, 3, padding=1) on 64x64 images (v2:186, v2:965 features[0]).
//
// The implicit-GEMM kernel is the wrong tool here: K = 9 taps x 4 (padded) channels = 36, its 32-wide K chunks turn
// that into 64 (44 % padding), and the 128x64 tiles spend their time in prologue/epilogue: 313 us for a layer whose
// floor is the 268 MB of output (about 55 us at 5 TB/s). This kernel keeps the whole 36x64 weight matrix in
// registers as MFMA B fragments (36 VGPRs per lane) and streams pixels: one wave owns 256 consecutive pixels of one
// image (8 groups of 32), per group 9 float4 tap loads per lane feed 18 k-steps x 2 channel tiles of
// v_mfma_f32_32x32x2_f32 with K = 36 exactly, the accumulators go straight to HBM (each store instruction writes
// two full 128-byte lines), and the LayerNorm2d statistics of the output (v2:152-153) are summed in registers and
// written as one slab per wave in the layout conv_inorm_finalize (gemm.hip) already consumes.
#include <cstdlib>

#include "fl_common.h"

namespace {

constexpr int GPW = 8;  // 32-pixel groups per wave

// ACT: activation fixed at compile time (a runtime switch inside the unrolled groups spills)
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

// H2: f16x2 split math (the scheme of conv_h2.hip) instead of the fp32 MFMA - the 36 MFMAs of 64 cycles per 32 pixels were
// 39 us of this 97 us kernel and did not overlap its 268 MB of stores; 18 MFMAs of 32 cycles are not worth mentioning. The
// weights are scaled by the power of two that puts their largest magnitude in [2^14, 2^15) (found by the wave itself) and
// split once per wave; the taps are scaled by 2^6 (|x| < 1023: images, VGG-normalised images; larger inputs give NaN, loudly)
// and split per pixel group. k = 4 tap + channel: group g of 16 k = taps 4g..4g+3 (lane half lh: taps 4g+2lh, 4g+2lh+1).
#ifndef FL_STEM_OCC
#define FL_STEM_OCC 2  // waves per SIMD the f16x2 variant is compiled for (A/B knob, tools/build_variant_lib.sh)
#endif
template <bool STATS, int ACT, bool H2>
__global__ __launch_bounds__(256, H2 ? FL_STEM_OCC : 3) void conv3x3_c4_o64_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             float* __restrict__ partial, int H, int W, float act_param) {
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  // everything about "which pixels" is wave-uniform: keep it in scalar registers (scalar bases + 32-bit lane offsets)
  const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // global wave id: (image, slab)
  const int HW = H * W, S = HW / (32 * GPW);              // GPW groups of 32 pixels per wave
  const int b = wave / S, slab = wave - b * S;
  const int gpr = W >> 5;                                 // 32-pixel groups per image row
  typedef __attribute__((address_space(1))) const f32x4v gld4_t;
  typedef __attribute__((address_space(1))) float gst1_t;

  // B fragments: step s (k = 2s + lh), channel tile j (co = 32j + lr);  wp = [64][3][3][4] -> w[k][co] = wp[co*36 + k]
  float wf[18][2];
  if constexpr (!H2) {
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      wf[s][0] = wp[lr * 36 + 2 * s + lh];
      wf[s][1] = wp[(32 + lr) * 36 + 2 * s + lh];
    }
  }
  h8_t wh[3][2], wl[3][2];
  float out_scale = 1.f;
  constexpr float XS = 64.f;  // 2^6
  if constexpr (H2) {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 36; ++k) m = fmaxf(m, fmaxf(fabsf(wp[lr * 36 + k]), fabsf(wp[(32 + lr) * 36 + k])));
    m = fl::wave_max(m);
    const int E = (int)((__float_as_uint(m) >> 23) & 0xffu);
    int e = E ? 141 - E : 0;  // m * 2^e in [2^14, 2^15)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    const float sw = __uint_as_float((unsigned)(e + 127) << 23);
    out_scale = __uint_as_float((unsigned)(127 - e) << 23) * (1.0f / XS);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = 16 * g + 8 * lh + q;
          const float X = k < 36 ? wp[(32 * j + lr) * 36 + k] * sw : 0.f;
          const _Float16 hh = (_Float16)X;
          wh[g][j][q] = hh;
          wl[g][j][q] = (_Float16)(X - (float)hh);
        }
  }
  const float b0 = bias ? bias[lr] : 0.f, b1 = bias ? bias[32 + lr] : 0.f;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  const float* xi = x + (size_t)b * HW * 4;
  float* yi = y + (size_t)b * HW * 64;

  // taps of group g: this lane's pixel for the A fragment (both lane halves read the same 16 bytes). The load is
  // unconditional from a clamped address and the border zero is a select (a predicated load would branch 72 times).
#define FL_LOAD_TAPS(g_, buf_)                                                                             \
  do {                                                                                                     \
    const int gi_ = slab * GPW + (g_);                                                                     \
    const int row_ = gi_ / gpr, c0_ = (gi_ - row_ * gpr) << 5;                                             \
    _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) {                                                     \
      const int iy_ = row_ + t_ / 3 - 1, ix_ = c0_ + lr + t_ % 3 - 1;                                      \
      const bool rok_ = (unsigned)iy_ < (unsigned)H, cok_ = (unsigned)ix_ < (unsigned)W;                   \
      const unsigned off_ = ((unsigned)(rok_ ? iy_ : 0) * (unsigned)W + (unsigned)(cok_ ? ix_ : 0)) * 4u;  \
      f32x4v v_ = *(gld4_t*)(xi + off_);                                                                   \
      asm volatile("" : "+v"(v_));                                                                         \
      const f32x4v z_ = {0.f, 0.f, 0.f, 0.f};                                                              \
      tp[buf_][t_] = (rok_ && cok_) ? v_ : z_;                                                             \
    }                                                                                                      \
  } while (0)
  f32x4v tp[2][9];
  FL_LOAD_TAPS(0, 0);
  // fully unrolled: the taps of group g+1 are requested BEFORE the 32 stores of group g (memory operations retire in
  // order per wave - a load issued behind the stores would wait for their write acknowledgements), and the waits stay
  // exact counts in straight-line code
  const unsigned vo = (unsigned)lr + (unsigned)lh * 256u;  // lane part of the output offset: channel, +4 pixels for the upper half
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    if (g + 1 < GPW) FL_LOAD_TAPS(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);  // keeps the unrolled groups from being hoisted on top of each other (VGPRs)
    const int gi = slab * GPW + g;
    const int row = gi / gpr, col0 = (gi - row * gpr) << 5;
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
    if constexpr (H2) {
#pragma unroll
      for (int kg = 0; kg < 3; ++kg) {
        // this lane's 8 k values of group kg: taps 4 kg + 2 lh and 4 kg + 2 lh + 1 (tap 9.. = padding: zeros)
        const f32x4v z = {0.f, 0.f, 0.f, 0.f};
        const f32x4v ta = lh ? (4 * kg + 2 < 9 ? tp[g & 1][4 * kg + 2 < 9 ? 4 * kg + 2 : 0] : z) : tp[g & 1][4 * kg < 9 ? 4 * kg : 0];
        const f32x4v tb = lh ? (4 * kg + 3 < 9 ? tp[g & 1][4 * kg + 3 < 9 ? 4 * kg + 3 : 0] : z)
                             : (4 * kg + 1 < 9 ? tp[g & 1][4 * kg + 1 < 9 ? 4 * kg + 1 : 0] : z);
        h8_t ah, al;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float X = (q < 4 ? ta[q & 3] : tb[q & 3]) * XS;
          const _Float16 hh = (_Float16)X;
          ah[q] = hh;
          al[q] = (_Float16)(X - (float)hh);
        }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[kg][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[kg][1], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[kg][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[kg][1], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[kg][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[kg][1], a1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float lo = lh ? tp[g & 1][t][1] : tp[g & 1][t][0];  // k = 4t + lh      (channels 0/1)
        const float hi = lh ? tp[g & 1][t][3] : tp[g & 1][t][2];  // k = 4t + 2 + lh  (channels 2/3)
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, wf[2 * t][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, wf[2 * t][1], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(hi, wf[2 * t + 1][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(hi, wf[2 * t + 1][1], a1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // C/D layout: channel = lane&31 (+32 for a1), pixel = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* yg = yi + ((size_t)row * W + col0) * 64;  // scalar base of the group's 32 x 64 outputs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned cst = (unsigned)(((r & 3) + 8 * (r >> 2)) * 64);
      float v0 = (H2 ? a0[r] * out_scale : a0[r]) + b0, v1 = (H2 ? a1[r] * out_scale : a1[r]) + b1;
      if (STATS) {
        s1[0] += v0; s2[0] += v0 * v0;
        s1[1] += v1; s2[1] += v1 * v1;
      }
      v0 = fl::apply_act(v0, ACT, act_param);
      v1 = fl::apply_act(v1, ACT, act_param);
      *(gst1_t*)(yg + (vo + cst)) = v0;
      *(gst1_t*)(yg + (vo + cst + 32u)) = v1;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef FL_LOAD_TAPS
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    if (lh == 0) {
      float* p = partial + ((size_t)(b * S + slab) * 2) * 64;
      p[lr] = s1[0];
      p[32 + lr] = s1[1];
      p[64 + lr] = s2[0];
      p[96 + lr] = s2[1];
    }
  }
}

}  // namespace

namespace fl {

bool conv3x3_c4_o64_applicable(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int act) {
  // H*W/(32*GPW) waves per image, 4 waves per block: B*H*W must split into whole blocks
  return (act == FL_ACT_NONE || act == FL_ACT_RELU) && Cin == 4 && Cout == 64 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && W % 32 == 0 &&
         (H * W) % (32 * GPW) == 0 && ((long long)B * H * W) % (4 * 32 * GPW) == 0 && B > 0;
}

int conv3x3_c4_o64_slabs(int H, int W) { return (H * W) / (32 * GPW); }

// partial (nullable): [B][S][2][64] sums / sums of squares of the biased, pre-activation output, S = H*W/(32*GPW)
int launch_conv3x3_c4_o64(const float* x, const float* wp, const float* bias, float* y, float* partial, int B, int H, int W,
                          int act, float act_param, hipStream_t s, int math) {
  const int blocks = (int)(((long long)B * H * W) / (4 * 32 * GPW));
  static const bool force_fp32 = getenv("FL_STEM_FP32") != nullptr;  // timing knob: the fp32-MFMA variant in split mode too
  if (force_fp32) math = 0;
#define FL_STEM(ST, AC)                                                                                                          \
  do {                                                                                                                           \
    if (math) hipLaunchKernelGGL((conv3x3_c4_o64_kernel<ST, AC, true>), dim3(blocks), dim3(256), 0, s, x, wp, bias, y, partial, H, W, act_param); \
    else hipLaunchKernelGGL((conv3x3_c4_o64_kernel<ST, AC, false>), dim3(blocks), dim3(256), 0, s, x, wp, bias, y, partial, H, W, act_param);    \
  } while (0)
  if (partial) {
    if (act != FL_ACT_NONE) {
      set_error("conv3x3_c4_o64: fused statistics are taken before the activation; act must be NONE");
      return FL_ERR_BAD_ARG;
    }
    FL_STEM(true, FL_ACT_NONE);
  } else if (act == FL_ACT_NONE) {
    FL_STEM(false, FL_ACT_NONE);
  } else if (act == FL_ACT_RELU) {
    FL_STEM(false, FL_ACT_RELU);
  } else {
    set_error("conv3x3_c4_o64: activation %d not instantiated", act);
    return FL_ERR_UNSUPPORTED;
  }
#undef FL_STEM
  return check_launch("conv3x3_c4_o64_kernel");
}

}  // namespace fl
