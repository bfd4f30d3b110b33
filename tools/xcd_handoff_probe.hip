// XCD-local phase hand-off probe (gfx950): what does one dependent phase cost when the blocks of ONE XCD hand 4 KB
// slices to each other through that XCD's L2, by barrier flavour? Decides the barrier of program.hip's local phases.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_handoff_probe.hip -o tools/build/xcd_handoff_probe && tools/build/xcd_handoff_probe
//
// Every phase: each block stores a 4 KB slice (plain stores; write-through in the device-wide variant), waits for its
// stores, passes the barrier, then loads two OTHER blocks' slices of the same phase and checks every word. Buffers are
// fresh per phase (as in the programs: every intermediate is written once per launch), so no L1 line can be stale.
// Variants:
//   0  device-wide barrier, agent-scope atomics on 8 spread counters + sc1 polls, write-through stores   (program.hip today)
//   1  XCD barrier, agent-scope atomic add + agent-scope (sc1) load poll                                 (FL_PROG_LOCAL=1 today)
//      (hipcc emits the same global_atomic_add for workgroup and agent scope and turns an idempotent RMW into an sc0 load:
//      there is no "L2 atomic" to be had on this memory, so the other variants use flags)
//   2  XCD barrier by FLAGS: each block stores its phase count into its own word of the XCD's 128-byte flag line (plain
//      store: stays in this XCD's L2), lanes 0..nx-1 of wave 0 poll the line with sc1 loads (L1 bypassed, L2 served)
//   3  as 2, every wave polls for itself (no LDS relay, no second workgroup barrier)
//   4  as 2 with write-through (sc0 sc1) flag stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int CTR_STRIDE = 1024;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr int SLICE = 1024;  // floats per block and phase (4 KB)

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const f32x4v gf4_t;
typedef __attribute__((address_space(1))) f32x4v gf4w_t;

__device__ __forceinline__ float expect(int p, int xcc, int slot, int i) { return (float)(((p * 131 + xcc * 17 + slot) * 1024 + i) & 0xFFFFF); }

// sync layout: [g * CTR_STRIDE] device-wide arrivals (g < 8); [x * CTR_STRIDE + 64] arrivals of XCD x;
// [8 * CTR_STRIDE + 16] abort; [8 * CTR_STRIDE + 128 + 32 x] registration of XCD x
template <int V>
__device__ __forceinline__ bool barrier(unsigned* sync, int xcc, int slot, int nx, unsigned phase, unsigned target_global, int* s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (V >= 2) {
    // flag line of this XCD: sync[xcc * CTR_STRIDE + 256 + slot]
    unsigned* line = &sync[xcc * CTR_STRIDE + 256];
    if (threadIdx.x == 0) {
      if (V == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(line + slot), "v"(phase) : "memory");
      else *(__attribute__((address_space(1))) unsigned*)(line + slot) = phase;
    }
    const int lane = threadIdx.x & 63;
    if (V == 3 || threadIdx.x < 64) {
      unsigned spins = 0;
      int ok = 1;
      for (;;) {
        unsigned v = phase;
        if (lane < nx) v = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64(v < phase) == 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { ok = 0; break; }
      }
      if (!ok && lane == 0) __hip_atomic_store(&sync[8 * CTR_STRIDE + 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (V == 3) return ok != 0;  // (a dead run: the other waves of the block time out by themselves)
      if (lane == 0) *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
  }
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    int ok = 1;
    if (V == 0) {
      __hip_atomic_fetch_add(&sync[(blockIdx.x & 7) * CTR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        unsigned tot = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) tot += __hip_atomic_load(&sync[g * CTR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tot >= target_global) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { ok = 0; break; }
      }
    } else {
      unsigned* ctr = &sync[xcc * CTR_STRIDE + 64];
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase * (unsigned)nx) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { ok = 0; break; }
      }
    }
    if (!ok) __hip_atomic_store(&sync[8 * CTR_STRIDE + 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

template <int V>
__global__ __launch_bounds__(256, 1) void probe(float* buf, unsigned* sync, int phases, unsigned long long* times, unsigned* errors,
                                                int* census) {
  extern __shared__ float smem[];  // sized by the host so that one block fits per CU
  __shared__ int s_ok, s_slot, s_nx;
  const int xcc = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & 7;
  if (threadIdx.x == 0) s_slot = (int)__hip_atomic_fetch_add(&sync[8 * CTR_STRIDE + 128 + 32 * xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool alive = barrier<0>(sync, xcc, 0, 1, 0, gridDim.x, &s_ok);
  if (threadIdx.x == 0) s_nx = (int)__hip_atomic_load(&sync[8 * CTR_STRIDE + 128 + 32 * xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot), nx = __builtin_amdgcn_readfirstlane(s_nx);
  if (slot == 0 && threadIdx.x == 0) census[xcc] = nx;
  unsigned err = 0;
  const unsigned long long t0 = wall_clock64();
  for (int p = 0; p < phases && alive; ++p) {
    float* base = buf + (size_t)p * 8 * 32 * SLICE + (size_t)xcc * 32 * SLICE;
    f32x4v v;
    const int i4 = threadIdx.x * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = expect(p, xcc, slot, i4 + e);
    if (V == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(base + slot * SLICE + i4), "v"(v) : "memory");
    else *(gf4w_t*)(base + slot * SLICE + i4) = v;
    alive = barrier<V>(sync, xcc, slot, nx, (unsigned)(p + 1), (unsigned)(p + 2) * gridDim.x, &s_ok);
    if (!alive) break;
    const int s1 = (slot + 1 + p) % nx, s2 = (slot + 7 + 3 * p) % nx;
    const f32x4v a = *(gf4_t*)(base + s1 * SLICE + i4), b = *(gf4_t*)(base + s2 * SLICE + i4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      err += a[e] != expect(p, xcc, s1, i4 + e);
      err += b[e] != expect(p, xcc, s2, i4 + e);
    }
    smem[threadIdx.x] = a[0] + b[0];  // keep the loads
  }
  const unsigned long long t1 = wall_clock64();
  if (err) atomicAdd(errors, err);
  if (!alive && threadIdx.x == 0) atomicAdd(errors + 1, 1u);
  if (threadIdx.x == 0) times[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 64;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  float* buf;
  unsigned *sync, *errors;
  unsigned long long* times;
  int* census;
  const size_t nbuf = (size_t)phases * 8 * 32 * SLICE;
  CHECK(hipMalloc(&buf, nbuf * 4));
  CHECK(hipMalloc(&sync, 9 * CTR_STRIDE * 4));
  CHECK(hipMalloc(&errors, 8));
  CHECK(hipMalloc(&times, G * 8));
  CHECK(hipMalloc(&census, 32));
  const int lds = 100 * 1024;
  unsigned long long* ht = (unsigned long long*)malloc(G * 8);
  printf("device %s, %d CUs, %d phases, wall clock 100 MHz\n", prop.name, G, phases);
  for (int v = 0; v < 5; ++v) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(buf, 0xFF, nbuf * 4));
      CHECK(hipMemset(sync, 0, 9 * CTR_STRIDE * 4));
      CHECK(hipMemset(errors, 0, 8));
      CHECK(hipMemset(census, 0, 32));
      CHECK(hipDeviceSynchronize());
#define LAUNCH(V)                                                                                                        \
  case V:                                                                                                                \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(probe<V>, dim3(G), dim3(256), lds, 0, buf, sync, phases, times, errors, census);                  \
    break;
      switch (v) {
        LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4)
      }
      CHECK(hipGetLastError());
      CHECK(hipDeviceSynchronize());
      unsigned he[2];
      int hc[8];
      CHECK(hipMemcpy(he, errors, 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(ht, times, G * 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hc, census, 32, hipMemcpyDeviceToHost));
      unsigned long long mx = 0, sum = 0;
      for (int i = 0; i < G; ++i) {
        mx = ht[i] > mx ? ht[i] : mx;
        sum += ht[i];
      }
      printf("variant %d rep %d: %.3f us/phase (slowest block), %.3f mean; wrong words %u, dead blocks %u; census %d %d %d %d %d %d %d %d\n",
             v, rep, mx * 0.01 / phases, sum * 0.01 / phases / G, he[0], he[1], hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
    }
  }
  return 0;
}
