// Which f16 MFMA shape sustains more FLOP/s under DVFS when the operands are RANDOM and come out of LDS (the conv kernels' regime)?
// v_mfma_f32_32x32x16_f16 (the shape conv_h2.hip issues) against v_mfma_f32_16x16x32_f16, same 64 x 64 output per wave, same LDS
// bytes per flop (16 ds_read_b128 per 32 k and wave either way), 8 waves per block, one block per CU, ~2 s of back-to-back
// launches before the timed ones (MI355X_MICROARCH.md, DVFS give-back (7)). Zero operands as the control (no clock give-back).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// LDS: 64 KB of f16 (random in [-1, 1) or zero); every iteration reads 16 KB-strided windows so that nothing is hoisted
template <int SHAPE>
__global__ __launch_bounds__(512) void shape_loop(float* out, int iters, int zero) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) {
    const unsigned h = hash(i * 2654435761u + blockIdx.x);
    _Float16 a = zero ? (_Float16)0.f : (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
    _Float16 b = zero ? (_Float16)0.f : (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
    reinterpret_cast<_Float16*>(lds)[2 * i] = a;
    reinterpret_cast<_Float16*>(lds)[2 * i + 1] = b;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float t = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int p0 = it * 8192 + w * 1024 + lane * 16;
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // 32 k = two 16-k groups: 8 reads + 12 MFMAs each (the conv kernel's tap)
        f16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            fa[i][q] = *reinterpret_cast<const f16x8*>(lds + ((p0 + (half * 8 + i * 2 + q) * 1024) & 0xffff));
            fb[i][q] = *reinterpret_cast<const f16x8*>(lds + ((p0 + (half * 8 + 4 + i * 2 + q) * 1024) & 0xffff));
          }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][pc == 0], fb[j][pc == 1], acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[a][b][e];
  } else {
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      const int p0 = it * 8192 + w * 1024 + lane * 16;
      f16x8 fa[4][2], fb[4][2];  // 32 k: 16 reads + 48 MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          fa[i][q] = *reinterpret_cast<const f16x8*>(lds + ((p0 + (i * 2 + q) * 1024) & 0xffff));
          fb[i][q] = *reinterpret_cast<const f16x8*>(lds + ((p0 + (8 + i * 2 + q) * 1024) & 0xffff));
        }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][pc == 0], fb[j][pc == 1], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) t += acc[a][b][e];
  }
  out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int SHAPE>
static void run(const char* name, int zero, float* out) {
  const int iters = 20000, blocks = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(shape_loop<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f, warm = 0.f;
  while (warm < 2000.f) {  // ~2 s of load before anything is timed
    hipEventRecord(e0);
    hipLaunchKernelGGL(shape_loop<SHAPE>, dim3(blocks), dim3(512), 65536, 0, out, iters, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    warm += ms;
  }
  float best = 0.f, sum = 0.f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(shape_loop<SHAPE>, dim3(blocks), dim3(512), 65536, 0, out, iters, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    sum += ms;
    best = best == 0.f || ms < best ? ms : best;
  }
  // per wave and iteration: 64 x 64 outputs x 32 k x 3 piece products
  const double flop = 2.0 * 64 * 64 * 32 * 3 * (double)iters * 8 * blocks;
  printf("%-34s %s  %8.3f ms (mean of 5)  %7.1f TFLOP/s issued f16\n", name, zero ? "zero operands  " : "random operands", sum / 5, flop / (sum / 5) / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<32>("v_mfma_f32_32x32x16_f16, LDS-fed", 0, out);
  run<16>("v_mfma_f32_16x16x32_f16, LDS-fed", 0, out);
  run<32>("v_mfma_f32_32x32x16_f16, LDS-fed", 1, out);
  run<16>("v_mfma_f32_16x16x32_f16, LDS-fed", 1, out);
  hipFree(out);
  return 0;
}
