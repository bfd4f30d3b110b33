// Matrix-pipe ceiling probe (gfx950): a register-only loop of MFMAs, no memory traffic, to see what fraction of the
// data-sheet dense peak the chip sustains (clock under matrix load, issue rate) - the practical ceiling for the split
// GEMM (v_mfma_f32_32x32x16_bf16) and the fp32 GEMM (v_mfma_f32_32x32x2_f32).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/build/mfma_peak && tools/build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float s = (float)(threadIdx.x & 7) * 0.125f;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(s + j); b[j] = (__bf16)(1.0f - s); }
  float fa = s, fb = 1.0f - s;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// MFMA + NV independent VALU ops per MFMA from the same wave: do the vector and the matrix pipe overlap (time = max) or
// serialise (time = sum)? The VALU chain is the operand-split pattern (and, sub, perm) on registers the MFMAs never read.
template <int NV>
__global__ __launch_bounds__(256) void mix_loop(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float s = (float)(threadIdx.x & 7) * 0.125f;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(s + j); b[j] = (__bf16)(1.0f - s); }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = s + 1.5f * j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        float& x = v[(i * NV + k) & 7];
        if (k & 1) x = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u) * 0.5f;
        else x = __uint_as_float(__float_as_uint(x) & 0xfffff000u) + 1.0f;
      }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[i][j];
#pragma unroll
  for (int j = 0; j < 8; ++j) t += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// The regime the conv kernels live in: NI SINGLE VALU instructions per MFMA (inline asm v_add_f32 on
// registers the MFMAs never touch, so the count is exact: no and+fma pairs, no fusion), same wave.
template <int NI>
__global__ __launch_bounds__(256) void mix1_loop(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float s = (float)(threadIdx.x & 7) * 0.125f;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(s + j); b[j] = (__bf16)(1.0f - s); }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = s + 1.5f * j;
  const float c = 1.0f + s;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NI; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i * NI + k) & 7]) : "v"(c));
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[i][j];
#pragma unroll
  for (int j = 0; j < 8; ++j) t += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NI>
static void run_mix1(int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mix1_loop<NI><<<blocks, 256>>>(out, iters / 8);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    mix1_loop<NI><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
  printf("bf16 MFMA + %2d single VALU instr/MFMA  blocks=%5d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (matrix work only)\n", NI, blocks,
         blocks / 256, best, flops / best * 1e-9);
  hipFree(out);
}

template <int NV>
static void run_mix(int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mix_loop<NV><<<blocks, 256>>>(out, iters / 8);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    mix_loop<NV><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
  printf("bf16 MFMA + %2d VALU/MFMA      blocks=%5d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (matrix work only)\n", NV, blocks, blocks / 256, best,
         flops / best * 1e-9);
  hipFree(out);
}

template <int MODE, int NACC>
static void run(const char* name, int blocks, int iters, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<MODE, NACC><<<blocks, 256>>>(out, iters / 8);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    mfma_loop<MODE, NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * NACC * flop_per_mfma;
  printf("%-28s blocks=%5d waves/SIMD=%d acc=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, blocks / 256, NACC, best, flops / best * 1e-9);
  hipFree(out);
}

int main() {
  const int it = 20000;
  run<0, 4>("bf16 32x32x16", 256, it, 32768.0);
  run<0, 4>("bf16 32x32x16", 512, it, 32768.0);
  run<0, 4>("bf16 32x32x16", 1024, it, 32768.0);
  run<0, 1>("bf16 32x32x16 dependent", 256, it, 32768.0);
  run<0, 2>("bf16 32x32x16", 512, it, 32768.0);
  run<1, 4>("fp32 32x32x2", 256, it, 4096.0);
  run<1, 4>("fp32 32x32x2", 512, it, 4096.0);
  run<0, 4>("bf16 32x32x16 sustained", 512, 400000, 32768.0);  // ~45 ms per launch: clock under sustained matrix load
  run_mix<0>(512, it); run_mix<4>(512, it); run_mix<8>(512, it); run_mix<12>(512, it); run_mix<16>(512, it);
  run_mix<8>(256, it); run_mix<8>(1024, it);
  run_mix1<0>(512, it); run_mix1<1>(512, it); run_mix1<2>(512, it); run_mix1<3>(512, it); run_mix1<4>(512, it);
  run_mix1<6>(512, it); run_mix1<8>(512, it); run_mix1<12>(512, it); run_mix1<4>(256, it); run_mix1<6>(256, it);
  return 0;
}
