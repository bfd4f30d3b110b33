// Common device/host helpers for the flowers_hip library (gfx950 / CDNA4 only).
// Wavefront = 64 lanes everywhere in this code base.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/flowers_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace fl {

// ---- error plumbing (never throws across the C ABI) -------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> status

#define FL_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      fl::set_error(__VA_ARGS__);             \
      return FL_ERR_BAD_ARG;                  \
    }                                         \
  } while (0)

// ---- activations -------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// accurate sigmoid (expf, not the fast intrinsic) – used where parity matters
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swish_acc(float x) { return x * sigmoid_acc(x); }
// x * sigmoid(x) on the fast transcendental pipe (v_exp_f32 + v_rcp_f32: a few ulp, far inside the parity tolerance): the
// inference-side streaming kernels, where the accurate expf makes a bandwidth-bound pass ALU-bound
__device__ __forceinline__ float swish_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// d/dx [x*sigmoid(x)] = s*(1 + x*(1-s))
__device__ __forceinline__ float swish_grad(float x) {
  float s = sigmoid_acc(x);
  return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float apply_act(float v, int act, float p) {
  switch (act) {
    case FL_ACT_SWISH: return swish_acc(v);
    case FL_ACT_RELU: return v > 0.f ? v : 0.f;
    case FL_ACT_LRELU: return v > 0.f ? v : v * p;
    case FL_ACT_SIGMOID: return sigmoid_acc(v);
    default: return v;
  }
}

// ---- wave / block reductions (64-lane waves) ------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum; `red` is >= (blockDim.x/64) floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// conv_first.hip: dedicated nn.Conv2d(3->64, 3x3, pad 1) kernel (input NHWC padded to 4 channels), used by the
// fl_conv2d_fwd / fl_conv2d_fwd_instnorm entry points of gemm.hip when the shape matches
bool conv3x3_c4_o64_applicable(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int act);
int conv3x3_c4_o64_slabs(int H, int W);
// math: 0 = fp32 MFMA, 1 = f16x2 split math (inputs |x| < 1023)
int launch_conv3x3_c4_o64(const float* x, const float* wp, const float* bias, float* y, float* partial, int B, int H, int W,
                          int act, float act_param, hipStream_t s, int math);

// 3x3 / stride 1 / pad 1 conv with pre-split weights, halo staged once per channel chunk (conv_halo.hip)
bool conv3x3_halo_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int launch_conv3x3_halo(const float* x, const void* wsplit, const float* bias, float* y, float* stats, int B, int H, int W,
                        int Cin, int Cout, int act, float act_param, hipStream_t s);
int launch_conv_halo(const float* x, const void* wsplit, const float* bias, float* y, float* stats, int B, int H, int W,
                     int Cin, int Cout, int ksize, int act, float act_param, int slab_rows, hipStream_t s);
int conv_halo_slab_rows(int ksize, int mode);
int launch_conv_inorm_finalize(const float* partial, const float* ln_w, const float* ln_b, float eps, float* scale, float* shift,
                               float* mean, float* rstd, int B, int HW, int C, int slabs, hipStream_t s);

// f16x2 split math (conv_h2.hip): 3x3 s1 p1 conv, weights packed once (header + fp16 h|l fragments)
bool conv3x3_h2_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
bool conv4x4s2_h2_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
// LayerNorm2d finish (scale / shift from the fused statistics) done inside the conv kernel where its tiles hold whole images;
// *finished = 0 when the caller still has to run launch_conv_inorm_finalize on `stats`
struct H2Finish {
  const float* ln_w;
  const float* ln_b;
  float eps;
  float* scale;
  float* shift;
  float* mean;
  float* rstd;
};
int launch_conv_h2(const float* x, const void* wpack, const float* bias, float* y, float* stats, const unsigned* a_amax,
                   int a_exp, const float* pre_scale, const float* pre_shift, int B, int H, int W, int Cin, int Cout, int ksize,
                   int act, float act_param, const H2Finish* fin, int* finished, hipStream_t s);
int conv_h2_lds_bytes(int ksize, int Win, int Cin, bool pre);
size_t pack_weight_f16x2_bytes(int N, int K);
int pack_weight_f16x2(const float* w, void* out, int N, int K, int layout, hipStream_t s);

}  // namespace fl
