// What does one CU pull global -> LDS when its working set sits in L2? Sizes the operand streams of program.hip's GEMM tiles.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_probe.hip -o tools/build/dma_rate_probe && tools/build/dma_rate_probe
// One 512-thread block per CU (LDS-limited). Every iteration moves one 32 KB "chunk" (32 x 1 KB pieces) into a 4-slot LDS ring,
// three chunks in flight, one raw s_barrier per chunk (the tile loop's shape, without the MFMAs). Variants:
//   0  4 loader waves, LDS-DMA (global_load_lds_dwordx4), block-private 256 KB source (L2 resident), 8 pieces per wave and chunk
//   1  8 loader waves, LDS-DMA, 4 pieces per wave and chunk
//   2  4 loader waves, global_load_dwordx4 to registers + ds_write_b128
//   3  as 0, the 32 blocks of an XCD read the SAME 256 KB (the A operand of an XCD-local phase)
//   4  as 0, half of every chunk from the shared region, half private (A + W of a tile)
//   5  as 1 with the half/half sources of 4
//   6  as 0 with 16 pieces in flight per wave at most (two chunks ahead instead of three)
//   7  the W operand as program.hip reads it from a row-major [n][k] weight with an 8 KB row pitch: the block's 256 KB are 32 rows,
//      a piece is 2 rows x 512 B (lane >> 5 picks the row), chunk c is the rows' bytes [512 c, 512 c + 512); other half of the
//      chunk from the shared region as in 4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) f32x4v lf4_t;
typedef __attribute__((address_space(1))) const f32x4v gf4_t;

constexpr int CHUNK = 8192;          // floats per chunk (32 KB)
constexpr int REGION = 8 * CHUNK;    // floats per source region (256 KB)

template <int V>
__global__ __launch_bounds__(512, 1) void probe(const float* src, int iters, unsigned long long* times, float* sink, int cmask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int xcc = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & 7;
  constexpr int NL = (V == 1 || V == 5) ? 8 : 4;    // loader waves
  constexpr int PPW = 32 / NL;                      // pieces per wave and chunk
  const float* priv = src + (size_t)(8 + blockIdx.x) * REGION;
  const float* shared = src + (size_t)xcc * REGION;
  const bool loader = w8 >= 8 - NL;
  const int lw = w8 - (8 - NL);
  const unsigned long long t0 = wall_clock64();
  if (loader) {
    auto issue = [&](int c) {
      float* dst = smem + (c & 3) * CHUNK;
      const int cc = c & cmask;  // chunk of the region (cmask 1: 64 KB per block, L2 resident; 7: 256 KB per block, beyond the XCD's 4 MB L2)
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int piece = lw * PPW + q;  // 0..31
        const float* base = V == 3 ? shared : ((V == 4 || V == 5 || V == 7) && piece < 16) ? shared : priv;
        const float* g = base + cc * CHUNK + piece * 256 + lane * 4;
        if (V == 7 && piece >= 16)  // rows 2 (piece - 16) + (lane >> 5) of 32 rows x 2048 floats; 16 chunks of 128 floats per row
          g = priv + (2 * (piece - 16) + (lane >> 5)) * 2048 + (c & 15) * 128 + (lane & 31) * 4;
        if (V == 2) {
          const f32x4v v = *(gf4_t*)g;
          *(lf4_t*)(dst + piece * 256 + lane * 4) = v;
        } else {
          __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + piece * 256), 16, 0, 0);
        }
      }
    };
    constexpr int AHEAD = V == 6 ? 2 : 3;
    for (int c = 0; c < AHEAD; ++c) issue(c);
    for (int c = 0; c < iters; ++c) {
      if (V != 2) {
        if (AHEAD == 3) {
          if (PPW == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
      }
      __builtin_amdgcn_s_barrier();
      issue(c + AHEAD);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    float acc = 0.f;
    for (int c = 0; c < iters; ++c) {
      __builtin_amdgcn_s_barrier();
      acc += smem[(c & 3) * CHUNK + threadIdx.x];  // a token read
    }
    if (acc == 123.456f) sink[threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) times[blockIdx.x] = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  float *src, *sink;
  unsigned long long* times;
  const size_t n = (size_t)(8 + G) * REGION;
  CHECK(hipMalloc(&src, n * 4));
  CHECK(hipMemset(src, 0, n * 4));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMalloc(&times, G * 8));
  unsigned long long* ht = (unsigned long long*)malloc(G * 8);
  const int lds = 4 * CHUNK * 4;
  printf("device %s, %d CUs, %d chunks of 32 KB per block\n", prop.name, G, iters);
  for (int cmask : {1, 7})
  for (int grid : {G, G / 4}) {
    for (int v = 0; v < 8; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
#define LAUNCH(V)                                                                                                       \
  case V:                                                                                                               \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(512), lds, 0, src, iters, times, sink, cmask);                               \
    break;
        switch (v) { LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) }
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ht, times, grid * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        unsigned long long mx = 0;
        for (int i = 0; i < grid; ++i) { sum += ht[i]; mx = ht[i] > mx ? ht[i] : mx; }
        const double us = sum / grid * 0.01;
        if (rep == 1)
          printf("private region %3d KB, grid %3d variant %d: %.3f us per 32 KB chunk (slowest block %.3f) = %.1f GB/s per CU\n", 32 * (cmask + 1), grid, v, us / iters,
                 mx * 0.01 / iters, 32768.0 * iters / us / 1e3);
      }
    }
  }
  return 0;
}
