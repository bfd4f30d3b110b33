// Persistent phase-program executor (include/flowers_hip.h, "Persistent phase-program executor").
//
// Why: the live denoiser of v2 (ConditionalUNet, v2:501-561) is a chain of ~35 dependent GEMMs with M = batch (256)
// and N, K in {256, 512, 1024}, plus row LayerNorms. One such GEMM is 3.4 us of MFMA work at most, while a kernel
// node of a hipGraph costs >= 4.5 us on MI355X whatever it does: the multi-kernel forward is launch-bound (71 nodes,
// 422 us, profiles/r01_unet_fwd_timeline.txt). Here ONE grid of (#CU) blocks stays resident and walks the phases;
// dependent phases are separated by a device-wide barrier.
//
// GEMM tile = 32x32 outputs per block. K is walked in 256-wide chunks: all 256 threads fetch the A and W chunk rows
// with fully coalesced dwordx4 loads (a wave covers one 1 KB row slice), park them in LDS ([row][256+4] images,
// conflict-free for ds_read_b128), and each of the four waves contracts its own 64-wide k slice with
// v_mfma_f32_32x32x2_f32 (exact fp32); the four partial tiles meet in LDS and all threads run the epilogue. (A first
// version loaded MFMA fragments straight from global memory: every lane then touches its own cache line and the
// texture addresser needs ~57 cycles per load instruction - 5 us just to ISSUE one tile's loads.)
// Two register stages + two LDS stages keep two chunks (2 x 2048 MFMA cycles) of loads in flight.
//
// Barrier: arrivals are spread over 8 counters 4 KB apart (memory-side atomics on one address serialise at ~25 ns
// each: 6.4 us for 256 blocks), pollers sum the 8 with relaxed sc1 loads; outputs are stored write-through so no L2
// write-back is needed; the invalidating acquire only where the program says a buffer is re-read after being re-written.
// The program itself is copied to LDS once (a cold descriptor fetch costs 1-2.5 us per phase otherwise).
#include <cstddef>
#include <cstring>
#include <mutex>

#include "fl_common.h"

namespace {

constexpr int TS = 32;             // output tile edge
constexpr int KC = 256;            // k values per block chunk (64 per wave)
constexpr int LDK = KC + 4;        // LDS row stride of a chunk image (floats)
constexpr int STAGE = 2 * TS * LDK;  // floats per LDS stage: 32 A rows then 32 W rows
constexpr int SLD = 36;            // LDS row stride of a partial output tile (floats)
constexpr int PROG_LDS_BYTES = 30 * 1024 - 512;  // LDS copy of the program (phases + jobs); 512 bytes left for the statics
constexpr int LDS_BYTES = 2 * STAGE * 4 + PROG_LDS_BYTES;
// LDS-DMA GEMM tile (FL_PG_DMA): 4-stage ring of 128-wide k chunks, 16 KB per operand and stage, unpadded images whose
// 16-byte column blocks are XOR-swizzled by the row (the swizzle is applied on the GLOBAL side of the DMA: lane L lands at
// LDS byte 16 L whatever it fetched)
constexpr int DKC = 128;                 // k values per chunk (32 per wave)
constexpr int DIMG = TS * DKC;           // floats per operand image
constexpr int DSTAGE = 2 * DIMG;         // floats per ring stage (A image, then W image)
constexpr int DSTAGES = 4;
static_assert(DSTAGES * DSTAGE <= 2 * STAGE, "the DMA ring lives in the two register-staged stages' LDS");
constexpr int NCTR = 8;            // arrival counters
constexpr int CTR_STRIDE = 1024;   // ... this many uints apart (4 KB)
constexpr int SYNC_UINTS = (NCTR + 1) * CTR_STRIDE;
constexpr unsigned SPIN_LIMIT = 1u << 21;  // barrier polls before a run is declared dead (~ seconds)

// -DFL_PROG_TRACE: timing-only build; io slot 15 is a uint64 buffer [block][phase][12] of 100 MHz wall-clock stamps
#ifndef FL_PROG_TRACE
#define FL_PROG_TRACE 0
#endif
#if FL_PROG_TRACE
__shared__ unsigned long long* g_trace;  // only thread 0 of the block touches these
__shared__ int g_trace_phases, g_trace_p;
#define FL_STAMP(k)                                                                                        \
  do {                                                                                                     \
    if (threadIdx.x == 0 && g_trace) g_trace[((size_t)blockIdx.x * g_trace_phases + g_trace_p) * 12 + (k)] = wall_clock64(); \
  } while (0)
#else
#define FL_STAMP(k) do {} while (0)
#endif

struct ProgIO {
  const void* p[FL_PROG_MAX_IO];
  fl_program_post post;
  unsigned* abort;  // the abort word inside `sync` (1 = a barrier timed out, 2 = a checked gather index was out of range)
};

// io slots are bound when the program is copied to LDS (program_kernel): descriptors hold plain addresses here
__device__ __forceinline__ const float* rp(uint64_t v, const ProgIO&) { return reinterpret_cast<const float*>(v); }
__device__ __forceinline__ float* wp(uint64_t v, const ProgIO& io) { return const_cast<float*>(rp(v, io)); }

// every operand lives in global memory: say so (pointers rebuilt from integers would otherwise be FLAT accesses,
// which tick both counters and make every wait a full drain)
typedef __attribute__((address_space(1))) const f32x4v gf4_t;  // native vector type: no struct copy through a generic ref
typedef __attribute__((address_space(1))) f32x4v gf4w_t;
typedef __attribute__((address_space(3))) f32x4v lf4_t;
typedef __attribute__((address_space(3))) float lf1_t;
__device__ __forceinline__ float4 ld4g(const float* p) {
  const f32x4v t = *(gf4_t*)(p);
  return make_float4(t[0], t[1], t[2], t[3]);
}
// Outputs are stored write-through (sc0 sc1: at memory, visible device-wide, once acknowledged - vmcnt), so the barrier
// needs no L2 write-back (buffer_wbl2 costs 1.2-2.3 us per block and phase). -DFL_PROG_WT_STORES=0 restores plain stores
// + a release fence per block.
// XCD-local programs (see program_kernel<true>) hand data over through the XCD's own L2 only: plain stores.
__device__ __forceinline__ f32x4v ld4gv(const float* p) { return *(gf4_t*)(p); }
__device__ __forceinline__ float ld1g(const float* p) { return *(__attribute__((address_space(1))) const float*)(p); }
__device__ __forceinline__ void st1g(bool wt, float* p, float v) {
  if (wt) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else *(__attribute__((address_space(1))) float*)(p) = v;
}
// Row scalars (LayerNorm statistics, loss norms, sum-of-squares partials) are 4-8 byte items that DIFFERENT blocks write into
// one cache line: a block whose XCD wrote part of such a line may keep a copy of it in its L2 whose other bytes are stale.
// Consumers read them with agent-scope loads (sc1: served coherently, never from a stale L2 copy), like the barrier counters.
__device__ __forceinline__ float ld1c(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long ldi64g(const long long* p) { return *(__attribute__((address_space(1))) const long long*)(p); }
__device__ __forceinline__ void st4g(bool wt, float* p, const float4& v) {
  f32x4v t;
  t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  // (a VMEM store of more than 64 bits reads its data registers AFTER issue: a VALU write to them needs 2 wait states on
  // gfx940+. hipcc's hazard recognizer inserts those for its own stores but cannot see inside inline asm: s_nop 1.)
  if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  else *(gf4w_t*)(p) = t;
}

// A job descriptor is read from the LDS copy into scalar registers (every field is wave-uniform); unused fields fold away.
// (field by field, typed: a punned word copy keeps the struct in scratch memory)
// ONE ds_read: lane i fetches word i of the 60-word descriptor, fields are then picked with v_readlane.
typedef const __attribute__((address_space(3))) int* lint_p;
typedef int jobwords_t;
__device__ __forceinline__ int ldsi(jobwords_t p, int word) { return __builtin_amdgcn_readlane(p, word); }
__device__ __forceinline__ float ldsf(jobwords_t p, int word) { return __int_as_float(ldsi(p, word)); }
__device__ __forceinline__ uint64_t ldsp(jobwords_t p, int word) {
  return (uint64_t)(unsigned)ldsi(p, word) | ((uint64_t)(unsigned)ldsi(p, word + 1) << 32);
}
#define FL_JW(f) ((int)(offsetof(fl_pjob, f) / 4))
__device__ __forceinline__ void load_job(const int* src /* LDS */, fl_pjob& d) {
  static_assert(sizeof(fl_pjob) / 4 <= 64, "descriptor must fit one word per lane");
  const int ln = threadIdx.x & 63;
  const jobwords_t p = ((lint_p)src)[ln < (int)(sizeof(fl_pjob) / 4) ? ln : 0];
#pragma unroll
  for (int s = 0; s < FL_PROG_MAX_SEG; ++s) {
    d.A[s] = ldsp(p, FL_JW(A) + 2 * s);
    d.W[s] = ldsp(p, FL_JW(W) + 2 * s);
    d.bias[s] = ldsp(p, FL_JW(bias) + 2 * s);
    d.K[s] = ldsi(p, FL_JW(K) + s);
    d.lda[s] = ldsi(p, FL_JW(lda) + s);
    d.ldw[s] = ldsi(p, FL_JW(ldw) + s);
    d.bias_scale[s] = ldsf(p, FL_JW(bias_scale) + s);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) d.aux[s] = ldsp(p, FL_JW(aux) + 2 * s);
  d.R = ldsp(p, FL_JW(R));
  d.C = ldsp(p, FL_JW(C));
  d.mask = ldsp(p, FL_JW(mask));
  d.gate = ldsp(p, FL_JW(gate));
  d.kind = ldsi(p, FL_JW(kind));
  d.M = ldsi(p, FL_JW(M));
  d.N = ldsi(p, FL_JW(N));
  d.nseg = ldsi(p, FL_JW(nseg));
  d.ldr = ldsi(p, FL_JW(ldr));
  d.ldc = ldsi(p, FL_JW(ldc));
  d.act = ldsi(p, FL_JW(act));
  d.mask_group = ldsi(p, FL_JW(mask_group));
  d.tile_begin = ldsi(p, FL_JW(tile_begin));
  d.ntiles = ldsi(p, FL_JW(ntiles));
  d.eps = ldsf(p, FL_JW(eps));
  d.eps2 = ldsf(p, FL_JW(eps2));
  d.bias_row_stride = ldsi(p, FL_JW(bias_row_stride));
  d.flags = ldsi(p, FL_JW(flags));
  d.alpha = ldsf(p, FL_JW(alpha));
  d.ld2 = ldsi(p, FL_JW(ld2));
  d.nrt = ldsi(p, FL_JW(nrt));
  d.per = ldsi(p, FL_JW(per));
}

// ---------------------------------------------------------------------------------------------- GEMM tile
struct Stage {
  f32x4v a[8], w[8];
};

__device__ __forceinline__ float act_grad(float u, int act) {
  switch (act) {
    case FL_ACT_SWISH: return fl::swish_grad(u);
    case FL_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case FL_ACT_SIGMOID: { const float sg = fl::sigmoid_acc(u); return sg * (1.f - sg); }
    default: return 1.f;
  }
}

// TA / TW: the operand is stored with the contraction index as its memory row (FL_PG_TA / FL_PG_TW): its chunk image in
// LDS is then [256 k][32] (a wave's loads and ds_write_b128 cover 8 rows x 128 B, fragments are 64 consecutive floats
// per ds_read_b32: both conflict-free) instead of [32][256 + 4].
template <bool TA, bool TW>
__device__ __forceinline__ void gemm_tile(const fl_pjob& jb, int rt, int ct, const ProgIO& io, float* smem, const bool wt) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const int kq4 = lane * 4;  // this thread's k offset inside a chunk row (a wave covers one whole 256-wide row slice)
  const int tr = threadIdx.x >> 3, tc4 = (threadIdx.x & 7) * 4;  // transposed operands: chunk rows tr + 32 i, 4 columns

  // chunk bookkeeping (all scalar): segment s owns chunks [cb_s, cb_{s+1})
  const float* A0 = rp(jb.A[0], io);
  const float* W0 = rp(jb.W[0], io);
  const float *A1 = A0, *A2 = A0, *W1 = W0, *W2 = W0;
  const int c1 = (jb.K[0] + KC - 1) / KC;
  int c2 = c1, nC = c1;
  if (jb.nseg > 1) {
    A1 = rp(jb.A[1], io);
    W1 = rp(jb.W[1], io);
    c2 = nC = c1 + (jb.K[1] + KC - 1) / KC;
  }
  if (jb.nseg > 2) {
    A2 = rp(jb.A[2], io);
    W2 = rp(jb.W[2], io);
    nC = c2 + (jb.K[2] + KC - 1) / KC;
  }
  const int K0 = jb.K[0], K1 = jb.K[1], K2 = jb.K[2];  // scalars: a select between array elements would pin jb in memory
  const int lda0 = jb.lda[0], lda1 = jb.lda[1], lda2 = jb.lda[2];
  const int ldw0 = jb.ldw[0], ldw1 = jb.ldw[1], ldw2 = jb.ldw[2];
  const int r0 = rt * TS + w, n0 = ct * TS + w;  // this thread loads rows r0 + 4i (A) and n0 + 4i (W), i < 8
  const int am0 = rt * TS + tc4, wn0 = ct * TS + tc4;  // ... or these columns of rows k (transposed operands)
  const int Mm1 = jb.M - 1;

  // global -> registers for chunk c;
  // okvar: this thread's 16-byte column lies inside the chunk's valid k range; kvvar: valid k rows of the chunk
#define FL_ISSUE(RS, cidx, okvar, kvvar)                                                           \
  do {                                                                                             \
    int c_ = (cidx);                                                                               \
    c_ = c_ < nC ? c_ : nC - 1;                                                                    \
    const bool s1_ = c_ >= c1, s2_ = c_ >= c2;                                                     \
    const float* Ab_ = s2_ ? A2 : (s1_ ? A1 : A0);                                                 \
    const float* Wb_ = s2_ ? W2 : (s1_ ? W1 : W0);                                                 \
    const int lda_ = s2_ ? lda2 : (s1_ ? lda1 : lda0);                                             \
    const int ldw_ = s2_ ? ldw2 : (s1_ ? ldw1 : ldw0);                                             \
    const int Ks_ = s2_ ? K2 : (s1_ ? K1 : K0);                                                    \
    const int kk_ = (c_ - (s2_ ? c2 : (s1_ ? c1 : 0))) * KC;                                       \
    okvar = kk_ + kq4 < Ks_;                                                                       \
    kvvar = Ks_ - kk_;                                                                             \
    const int kcol_ = okvar ? kk_ + kq4 : kk_;                                                     \
    /* 32-bit element offsets from a scalar base (validated: rows*ld < 2^30): full-rate 24-bit multiplies */       \
    if constexpr (TA) {                                                                            \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                           \
        int kr_ = kk_ + tr + 32 * i_;                                                              \
        kr_ = kr_ < Ks_ ? kr_ : Ks_ - 1;                                                           \
        RS.a[i_] = ld4gv(Ab_ + (size_t)(__umul24((unsigned)kr_, (unsigned)lda_) + (unsigned)am0)); \
      }                                                                                            \
    } else {                                                                                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                           \
        int ar_ = r0 + 4 * i_;                                                                     \
        ar_ = ar_ < Mm1 ? ar_ : Mm1;                                                               \
        RS.a[i_] = ld4gv(Ab_ + (size_t)(__umul24((unsigned)ar_, (unsigned)lda_) + (unsigned)kcol_)); \
      }                                                                                            \
    }                                                                                              \
    if constexpr (TW) {                                                                            \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                           \
        int kr_ = kk_ + tr + 32 * i_;                                                              \
        kr_ = kr_ < Ks_ ? kr_ : Ks_ - 1;                                                           \
        RS.w[i_] = ld4gv(Wb_ + (size_t)(__umul24((unsigned)kr_, (unsigned)ldw_) + (unsigned)wn0)); \
      }                                                                                            \
    } else {                                                                                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                             \
        RS.w[i_] = ld4gv(Wb_ + (size_t)(__umul24((unsigned)(n0 + 4 * i_), (unsigned)ldw_) + (unsigned)kcol_)); \
    }                                                                                              \
  } while (0)

  // registers -> LDS stage (A image, then W image at TS*LDK); out-of-range k become zeros
#define FL_PARK(RS, okvar, kvvar, st)                                                              \
  do {                                                                                             \
    float* Ia_ = smem + (st) * STAGE;                                                              \
    float* Iw_ = Ia_ + TS * LDK;                                                                   \
    const f32x4v z_ = {0.f, 0.f, 0.f, 0.f};                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                             \
      if constexpr (TA) *(lf4_t*)(Ia_ + (tr + 32 * i_) * TS + tc4) = tr + 32 * i_ < kvvar ? RS.a[i_] : z_; \
      else *(lf4_t*)(Ia_ + (w + 4 * i_) * LDK + kq4) = okvar ? RS.a[i_] : z_;                      \
      if constexpr (TW) *(lf4_t*)(Iw_ + (tr + 32 * i_) * TS + tc4) = tr + 32 * i_ < kvvar ? RS.w[i_] : z_; \
      else *(lf4_t*)(Iw_ + (w + 4 * i_) * LDK + kq4) = okvar ? RS.w[i_] : z_;                      \
    }                                                                                              \
  } while (0)

  // each wave contracts its 64-wide k slice of the parked chunk: 16 ds_read_b128 (or 4 x ds_read_b32 each) + 32 MFMA;
  // MFMA (q, e) of lane half lh takes k = 64 w + 32 (q>>2) + 16 lh + 4 (q&3) + e from both operands
#define FL_CONTRACT(st)                                                                            \
  do {                                                                                             \
    const float* Ia_ = smem + (st) * STAGE;                                                        \
    const float* Iw_ = Ia_ + TS * LDK;                                                             \
    f32x4v fa_[8], fw_[8]; /* all fragment reads in flight before the first MFMA */               \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                             \
      const int kb_ = 64 * w + 32 * (q_ >> 2) + 16 * lh + 4 * (q_ & 3);                            \
      if constexpr (TA) {                                                                          \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) fa_[q_][e_] = *(const lf1_t*)(Ia_ + (kb_ + e_) * TS + lr); \
      } else fa_[q_] = *(const lf4_t*)(Ia_ + lr * LDK + kb_);                                      \
      if constexpr (TW) {                                                                          \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) fw_[q_][e_] = *(const lf1_t*)(Iw_ + (kb_ + e_) * TS + lr); \
      } else fw_[q_] = *(const lf4_t*)(Iw_ + lr * LDK + kb_);                                      \
    }                                                                                              \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                             \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                             \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[q_][e_], fw_[q_][e_], acc, 0, 0, 0);        \
    }                                                                                              \
  } while (0)

  // FL_PG_COLSUM (A transposed, column-0 tiles): sums over k of the parked A image, 8 row groups x 32 columns
  const bool colsum = TA && (jb.flags & FL_PG_COLSUM) != 0 && ct == 0;
  float cs = 0.f;
#define FL_COLSUM(st)                                                                              \
  do {                                                                                             \
    if constexpr (TA) {                                                                            \
      if (colsum) {                                                                                \
        const float* Ia_ = smem + (st) * STAGE + (threadIdx.x >> 5) * 32 * TS + (threadIdx.x & 31); \
        _Pragma("unroll 8") for (int j_ = 0; j_ < 32; ++j_) cs += *(const lf1_t*)(Ia_ + j_ * TS);  \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // epilogue operands are requested first: they ride along with the chunk loads instead of costing a memory round
  // trip (about 1 us here) after the contraction
  const int er = threadIdx.x >> 3, ec = (threadIdx.x & 7) * 4;
  const int gr = rt * TS + er, gc = ct * TS + ec;
  const int grc = gr < jb.M ? gr : jb.M - 1;
  float4 e_b[FL_PROG_MAX_SEG], e_r = make_float4(0.f, 0.f, 0.f, 0.f), e_p = make_float4(0.f, 0.f, 0.f, 0.f);
  float e_gate = 0.f, e_m = 1.f;
#pragma unroll
  for (int s = 0; s < FL_PROG_MAX_SEG; ++s) {
    e_b[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (jb.bias[s]) {
      const float* bp = rp(jb.bias[s], io) + gc;
      if (s == 0 && jb.aux[2])  // row of a per-timestep table, picked by a device-resident index
        bp += ldi64g(reinterpret_cast<const long long*>(rp(jb.aux[2], io))) * jb.bias_row_stride;
      e_b[s] = ld4g(bp);
    }
  }
  const bool actbwd = (jb.flags & FL_PG_ACTBWD) != 0;
  if (jb.gate) e_gate = ld1g(rp(jb.gate, io));
  if (jb.mask) e_m = ld1g(rp(jb.mask, io) + (long long)grc * (jb.N / jb.mask_group) + gc / jb.mask_group);
  if (jb.R) e_r = ld4g(rp(jb.R, io) + (long long)grc * jb.ldr + gc);
  if (actbwd) e_p = ld4g(rp(jb.aux[1], io) + (long long)grc * jb.ld2 + gc);
  Stage R0, R1;
  bool ok0 = false, ok1 = false;
  int kv0 = 0, kv1 = 0;
  FL_STAMP(7);
  // (chunks past the last one are not requested: a one-chunk tile - every weight-gradient tile at batch 256 - would
  // otherwise fetch its operands four times over)
  FL_ISSUE(R0, 0, ok0, kv0);
  if (nC > 1) FL_ISSUE(R1, 1, ok1, kv1);
  FL_STAMP(8);
  for (int c = 0; c < nC; c += 2) {
    FL_PARK(R0, ok0, kv0, 0);
    if (c + 2 < nC) FL_ISSUE(R0, c + 2, ok0, kv0);
    __syncthreads();
    FL_CONTRACT(0);
    FL_COLSUM(0);
    const bool more = c + 1 < nC;  // block-uniform
    if (more) FL_PARK(R1, ok1, kv1, 1);
    if (c + 3 < nC) FL_ISSUE(R1, c + 3, ok1, kv1);
    __syncthreads();
    if (more) {
      FL_CONTRACT(1);
      FL_COLSUM(1);
    }
  }
#undef FL_ISSUE
#undef FL_PARK
#undef FL_CONTRACT
#undef FL_COLSUM
  FL_STAMP(1);

  // meet in LDS: sm[w][row][col] (aliases stage 0; the loop's last barrier precedes every wave's last stage-0 read
  // only when nC is even, so fence once more); C/D layout of 32x32x2: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  __syncthreads();
  float* mine = smem + w * (TS * SLD);
#pragma unroll
  for (int r = 0; r < 16; ++r) *(lf1_t*)(mine + ((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + lr) = acc[r];
  if constexpr (TA) {
    if (colsum) *(lf1_t*)(smem + 4 * TS * SLD + threadIdx.x) = cs;  // [8 row groups][32 columns], behind the partial tiles
  }
  __syncthreads();
  const float* s0 = smem + er * SLD + ec;
  f32x4v vv = *(const lf4_t*)(s0);
#pragma unroll
  for (int k = 1; k < 4; ++k) vv += *(const lf4_t*)(s0 + k * (TS * SLD));
  float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
  const float gmul = jb.gate ? 1.f + fl::sigmoid_acc(e_gate) : 1.f;
  if constexpr (TA) {
    if (colsum && threadIdx.x < TS) {  // FL_PG_TA requires M % 32 == 0: every row of the tile exists
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += *(const lf1_t*)(smem + 4 * TS * SLD + g * TS + threadIdx.x);
      st1g(wt, wp(jb.aux[0], io) + rt * TS + threadIdx.x, jb.alpha * gmul * t);
    }
  }
  if (gr < jb.M) {
#pragma unroll
    for (int s = 0; s < FL_PROG_MAX_SEG; ++s) {
      if (jb.bias[s]) {
        const float sc = jb.bias_scale[s] * gmul;
        v.x += sc * e_b[s].x; v.y += sc * e_b[s].y; v.z += sc * e_b[s].z; v.w += sc * e_b[s].w;
      }
    }
    if (jb.aux[3]) st4g(wt, wp(jb.aux[3], io) + (long long)gr * jb.ldc + gc, v);  // pre-activation copy (saved for backward)
    if (actbwd) {
      v.x *= act_grad(e_p.x, jb.act); v.y *= act_grad(e_p.y, jb.act);
      v.z *= act_grad(e_p.z, jb.act); v.w *= act_grad(e_p.w, jb.act);
    } else {
      v.x = fl::apply_act(v.x, jb.act, 0.f);
      v.y = fl::apply_act(v.y, jb.act, 0.f);
      v.z = fl::apply_act(v.z, jb.act, 0.f);
      v.w = fl::apply_act(v.w, jb.act, 0.f);
    }
    if (jb.mask) {
      v.x *= e_m; v.y *= e_m; v.z *= e_m; v.w *= e_m;
    }
    if (jb.R) {
      v.x += e_r.x; v.y += e_r.y; v.z += e_r.z; v.w += e_r.w;
    }
    st4g(wt, wp(jb.C, io) + (long long)gr * jb.ldc + gc, v);
  }
  __syncthreads();  // the partial tiles alias stage 0 of the next tile
}

// FL_PG_T64 (with TA | TW, one segment, no epilogue besides the bias column sums): a weight-gradient tile of 64 x 64 outputs.
// Both operand chunks ([256 k][64] floats, 64 KB each) go global -> LDS by LDS-DMA into ONE stage, every wave owns a 32 x 32
// quadrant over the whole contraction (no partial tiles to merge) and stores it straight from the accumulators. Against
// four 32 x 32 tiles: half the operand traffic, no LDS round trip for the partial sums, 128 MFMAs per wave behind one
// load latency instead of 32 (measured: 4.6 us per 32 x 32 tile -> 7 us per 64 x 64).
__device__ __forceinline__ void wg64_tile(const fl_pjob& jb, int rt, int ct, const ProgIO& io, float* smem, const bool wt) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const int wr = w >> 1, wc = w & 1;
  const float* A0 = rp(jb.A[0], io) + rt * 64 + (lane & 15) * 4;
  const float* W0 = rp(jb.W[0], io) + ct * 64 + (lane & 15) * 4;
  const int K = jb.K[0], lda = jb.lda[0], ldw = jb.ldw[0];
  float* imgA = smem;
  float* imgW = smem + 256 * 64;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool colsum = (jb.flags & FL_PG_COLSUM) != 0 && ct == 0 && wc == 0;
  float cs = 0.f;
  const float e_gate = jb.gate ? ld1g(rp(jb.gate, io)) : 0.f;
  for (int kk = 0; kk < K; kk += 256) {
    const int kv = K - kk < 256 ? K - kk : 256;
    if (kk) __syncthreads();  // every wave has read the previous chunk
    // wave w moves rows [64 w, 64 w + 64) of both images: 16 x (4 rows x 256 B) each; rows past the end re-read row K-1
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row4 = (w * 16 + q) * 4;
      if (row4 < kv) {  // wave-uniform
        int kr = kk + row4 + (lane >> 4);
        kr = kr < K ? kr : K - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(A0 + (size_t)__umul24((unsigned)kr, (unsigned)lda)), (lptr_t)(imgA + row4 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(W0 + (size_t)__umul24((unsigned)kr, (unsigned)ldw)), (lptr_t)(imgW + row4 * 64), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kv & 1) {  // MFMA steps take k in pairs: the partner row of the last one is zero
      if (threadIdx.x < 16) *(lf4_t*)(imgA + kv * 64 + threadIdx.x * 4) = f32x4v{0.f, 0.f, 0.f, 0.f};
      else if (threadIdx.x < 32) *(lf4_t*)(imgW + kv * 64 + (threadIdx.x - 16) * 4) = f32x4v{0.f, 0.f, 0.f, 0.f};
      __syncthreads();
    }
    const int steps = (kv + 1) >> 1;
    const float* pa = imgA + lh * 64 + 32 * wr + lr;
    const float* pw = imgW + lh * 64 + 32 * wc + lr;
    int s = 0;
    for (; s + 16 <= steps; s += 16) {
      float fa[16], fw[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        fa[j] = *(const lf1_t*)(pa + (s + j) * 128);
        fw[j] = *(const lf1_t*)(pw + (s + j) * 128);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fw[j], acc, 0, 0, 0);
        cs += fa[j];
      }
    }
    for (; s < steps; ++s) {
      const float a = *(const lf1_t*)(pa + s * 128), b = *(const lf1_t*)(pw + s * 128);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      cs += a;
    }
  }
  // C/D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* C = wp(jb.C, io) + (long long)(rt * 64 + 32 * wr + 4 * lh) * jb.ldc + ct * 64 + 32 * wc + lr;
#pragma unroll
  for (int r = 0; r < 16; ++r) st1g(wt, C + (long long)((r & 3) + 8 * (r >> 2)) * jb.ldc, acc[r]);
  if (colsum) {
    cs += __shfl_xor(cs, 32, 64);
    const float gmul = jb.gate ? 1.f + fl::sigmoid_acc(e_gate) : 1.f;
    if (lh == 0) st1g(wt, wp(jb.aux[0], io) + rt * 64 + 32 * wr + lr, jb.alpha * gmul * cs);
  }
  __syncthreads();  // the images are the next tile's stage
}

// ---------------------------------------------------------------------------------------------- LDS-DMA GEMM tile
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(3))) int* ldsint_p;

// s_waitcnt vmcnt(n) for a wave-uniform n in {0, 4, 8, 12, 16, 20, 24} (anything else waits for everything)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// The W side of a FL_PG_DMA job as the tile loop needs it: read from the job's LDS copy (a few words, wave-uniform), so the
// NEXT tile's weights can be requested without holding a second descriptor in registers.
// The W side of a FL_PG_DMA job is read from the job's LDS copy into LOCAL scalars (a struct or array whose members are
// then picked by the chunk's segment would be kept in scratch memory: hipcc turns the select into an indexed load).
__device__ __forceinline__ int lds_word(const int* job, int word) { return __builtin_amdgcn_readfirstlane(((ldsint_p)job)[word]); }
__device__ __forceinline__ const float* lds_ptr(const int* job, int word) {
  return reinterpret_cast<const float*>((uint64_t)(unsigned)lds_word(job, word) | ((uint64_t)(unsigned)lds_word(job, word + 1) << 32));
}
#define FL_LOAD_WSRC(P, job)                                                                         \
  const float* P##W0 = lds_ptr(job, FL_JW(W));                                                       \
  const float* P##W1 = lds_ptr(job, FL_JW(W) + 2);                                                   \
  const float* P##W2 = lds_ptr(job, FL_JW(W) + 4);                                                   \
  const int P##ldw0 = lds_word(job, FL_JW(ldw)), P##ldw1 = lds_word(job, FL_JW(ldw) + 1),            \
            P##ldw2 = lds_word(job, FL_JW(ldw) + 2);                                                 \
  const int P##nseg = lds_word(job, FL_JW(nseg));                                                    \
  const int P##c1 = lds_word(job, FL_JW(K)) / DKC;                                                   \
  const int P##c2 = P##nseg > 1 ? P##c1 + lds_word(job, FL_JW(K) + 1) / DKC : P##c1;                 \
  const int P##nC = P##nseg > 2 ? P##c2 + lds_word(job, FL_JW(K) + 2) / DKC : P##c2;                 \
  const bool P##tw = (lds_word(job, FL_JW(flags)) & FL_PG_TW) != 0
#define FL_WSRC_ARGS(P) P##W0, P##W1, P##W2, P##ldw0, P##ldw1, P##ldw2, P##c1, P##c2, P##tw

// W chunk c of column tile ct -> the W image of ring stage c % DSTAGES (4 LDS-DMA instructions per wave).
//   not TW: W[n][k] rows n = 32 ct + 8 w + 2 i + (lane >> 5); lane fetches the 16-byte block j = (lane & 31) ^ (row & 15)
//           of the chunk (row = 8 w + 2 i + (lane >> 5) inside the tile) and lands at block position lane & 31
//   TW:     W[k][n] rows k = 32 w + 8 i + (lane >> 3) of the chunk, 128 B each: the image is [128 k][32 n], as it lies
__device__ __forceinline__ void issue_w_chunk(const float* W0, const float* W1, const float* W2, int ldw0, int ldw1, int ldw2,
                                              int c1, int c2, bool tw, int c, int ct, float* smem) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const bool s1 = c >= c1, s2 = c >= c2;
  const float* Wb = s2 ? W2 : (s1 ? W1 : W0);
  const int ldw = s2 ? ldw2 : (s1 ? ldw1 : ldw0);
  const int kk = (c - (s2 ? c2 : (s1 ? c1 : 0))) * DKC;
  float* img = smem + (c & (DSTAGES - 1)) * DSTAGE + DIMG;
  if (tw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = __umul24((unsigned)(kk + 32 * w + 8 * i + (lane >> 3)), (unsigned)ldw) + (unsigned)(ct * TS + (lane & 7) * 4);
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)off), (lptr_t)(img + (32 * w + 8 * i) * TS), 16, 0, 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * w + 2 * i + (lane >> 5);
      const unsigned off = __umul24((unsigned)(ct * TS + r), (unsigned)ldw) + (unsigned)(kk + 4 * ((lane & 31) ^ (r & 15)));
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)off), (lptr_t)(img + (8 * w + 2 * i) * DKC), 16, 0, 0);
    }
  }
}

// Requests the first (up to 3) W chunks of the block's NEXT tile (pf_job = its descriptor in LDS; FL_PG_DMA | FL_PG_WCONST).
// Ring stages 0..2 are free once every wave has left the current tile's loop; stage 3 holds the partial tiles meanwhile.
// Returns the number of DMA instructions each wave issued (the caller's waits leave exactly these in flight).
__device__ __forceinline__ int prefetch_w(const int* pf_job, int pf_ct, float* smem) {
  FL_LOAD_WSRC(p_, pf_job);
  const int n = p_nC < 3 ? p_nC : 3;
  for (int c = 0; c < n; ++c) issue_w_chunk(FL_WSRC_ARGS(p_), c, pf_ct, smem);
  return 4 * n;
}

// One 32 x 32 output tile of a FL_PG_DMA job. `pre`: DMA instructions per wave of THIS tile's first W chunks already in
// flight (prefetch_w, issued before the preceding barrier / the previous tile's epilogue). pf_job != nullptr: the block's
// next tile is a prefetchable GEMM (column tile pf_ct); returns what prefetch_w issued for it (0 if nothing).
template <bool TW>
__device__ __forceinline__ int gemm_dma_tile(const fl_pjob& jb, int rt, int ct, const ProgIO& io, float* smem, const bool wt,
                                             const int pre, const int* my_job, const int* pf_job, const int pf_ct) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  FL_LOAD_WSRC(ws_, my_job);
  const float* A0 = rp(jb.A[0], io);
  const float *A1 = A0, *A2 = A0;
  if (jb.nseg > 1) A1 = rp(jb.A[1], io);
  if (jb.nseg > 2) A2 = rp(jb.A[2], io);
  const int lda0 = jb.lda[0], lda1 = jb.lda[1], lda2 = jb.lda[2];
  const int nC = ws_nC;
  // A rows of this lane's four DMA instructions (clamped: rows past M repeat the last one, their results are never stored)
  int arow[4], acol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * w + 2 * i + (lane >> 5);
    const int gr_ = rt * TS + r;
    arow[i] = gr_ < jb.M ? gr_ : jb.M - 1;
    acol[i] = 4 * ((lane & 31) ^ (r & 15));
  }
#define FL_ISSUE_A(cidx)                                                                               \
  do {                                                                                                 \
    const int c_ = (cidx);                                                                             \
    const bool s1_ = c_ >= ws_c1, s2_ = c_ >= ws_c2;                                                   \
    const float* Ab_ = s2_ ? A2 : (s1_ ? A1 : A0);                                                     \
    const int lda_ = s2_ ? lda2 : (s1_ ? lda1 : lda0);                                                 \
    const int kk_ = (c_ - (s2_ ? ws_c2 : (s1_ ? ws_c1 : 0))) * DKC;                                    \
    float* img_ = smem + (c_ & (DSTAGES - 1)) * DSTAGE;                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                 \
      const unsigned off_ = __umul24((unsigned)arow[i_], (unsigned)lda_) + (unsigned)(kk_ + acol[i_]); \
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab_ + (size_t)off_), (lptr_t)(img_ + (8 * w + 2 * i_) * DKC), 16, 0, 0); \
    }                                                                                                  \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // epilogue operands are requested first: they ride along with the chunk loads
  const int er = threadIdx.x >> 3, ec = (threadIdx.x & 7) * 4;
  const int gr = rt * TS + er, gc = ct * TS + ec;
  const int grc = gr < jb.M ? gr : jb.M - 1;
  float4 e_b[FL_PROG_MAX_SEG], e_r = make_float4(0.f, 0.f, 0.f, 0.f), e_p = make_float4(0.f, 0.f, 0.f, 0.f);
  float e_gate = 0.f, e_m = 1.f;
#pragma unroll
  for (int s = 0; s < FL_PROG_MAX_SEG; ++s) {
    e_b[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (jb.bias[s]) {
      const float* bp = rp(jb.bias[s], io) + gc;
      if (s == 0 && jb.aux[2])  // row of a per-timestep table, picked by a device-resident index
        bp += ldi64g(reinterpret_cast<const long long*>(rp(jb.aux[2], io))) * jb.bias_row_stride;
      e_b[s] = ld4g(bp);
    }
  }
  const bool actbwd = (jb.flags & FL_PG_ACTBWD) != 0;
  if (jb.gate) e_gate = ld1g(rp(jb.gate, io));
  if (jb.mask) e_m = ld1g(rp(jb.mask, io) + (long long)grc * (jb.N / jb.mask_group) + gc / jb.mask_group);
  if (jb.R) e_r = ld4g(rp(jb.R, io) + (long long)grc * jb.ldr + gc);
  if (actbwd) e_p = ld4g(rp(jb.aux[1], io) + (long long)grc * jb.ld2 + gc);
  FL_STAMP(7);
  // prologue: chunks 0..2. DMA instructions per wave and chunk: 8 (A + W), or 4 where the W half came with the prefetch
  const int npre = pre >> 2;  // W chunks already in flight
  const int n0 = nC < 3 ? nC : 3;
  for (int c = 0; c < n0; ++c) {
    FL_ISSUE_A(c);
    if (c >= npre) issue_w_chunk(ws_W0, ws_W1, ws_W2, ws_ldw0, ws_ldw1, ws_ldw2, ws_c1, ws_c2, TW, c, ct, smem);
  }
  FL_STAMP(8);
  // instructions issued AFTER chunk c's (the wait at the top of iteration c leaves exactly those in flight)
  int after1 = n0 > 1 ? (1 < npre ? 4 : 8) : 0, after2 = n0 > 2 ? (2 < npre ? 4 : 8) : 0;
  for (int c = 0; c < nC; ++c) {
    wait_vm(after1 + after2);
    __syncthreads();  // chunk c has landed for every wave; every wave has left chunk c - 1 (its stage is free)
#if FL_PROG_TRACE
    if (c == 0) FL_STAMP(9);
#endif
    const float* Ia = smem + (c & (DSTAGES - 1)) * DSTAGE;
    const float* Iw = Ia + DIMG;
    f32x4v fa[4], fw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 8 * w + 4 * lh + q;  // 16-byte k block of this lane's fragment: k = 4 j + e
      fa[q] = *(const lf4_t*)(Ia + lr * DKC + 4 * (j ^ (lr & 15)));
      if constexpr (TW) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fw[q][e] = *(const lf1_t*)(Iw + (4 * j + e) * TS + lr);
      } else {
        fw[q] = *(const lf4_t*)(Iw + lr * DKC + 4 * (j ^ (lr & 15)));
      }
    }
    const bool more = c + 3 < nC;  // block-uniform
    if (more) {
      // chunk c + 3 -> the stage chunk c - 1 has just left: one DMA instruction after every second MFMA (the matrix
      // pipe is busy for 64 cycles per MFMA; a DMA issue is ~60-100)
      const int c3 = c + 3;
      const bool s1 = c3 >= ws_c1, s2 = c3 >= ws_c2;
      const float* Ab = s2 ? A2 : (s1 ? A1 : A0);
      const float* Wb = s2 ? ws_W2 : (s1 ? ws_W1 : ws_W0);
      const int lda = s2 ? lda2 : (s1 ? lda1 : lda0), ldw = s2 ? ws_ldw2 : (s1 ? ws_ldw1 : ws_ldw0);
      const int kk = (c3 - (s2 ? ws_c2 : (s1 ? ws_c1 : 0))) * DKC;
      float* imgA = smem + (c3 & (DSTAGES - 1)) * DSTAGE;
      float* imgW = imgA + DIMG;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][e], fw[q][e], acc, 0, 0, 0);
          const int n = 4 * q + e;
          if (n & 1) {
            const int i = (n >> 1) & 3;
            if (n < 8) {
              const unsigned off = __umul24((unsigned)arow[i], (unsigned)lda) + (unsigned)(kk + acol[i]);
              __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)off), (lptr_t)(imgA + (8 * w + 2 * i) * DKC), 16, 0, 0);
            } else if constexpr (TW) {
              const unsigned off = __umul24((unsigned)(kk + 32 * w + 8 * i + (lane >> 3)), (unsigned)ldw) + (unsigned)(ct * TS + (lane & 7) * 4);
              __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)off), (lptr_t)(imgW + (32 * w + 8 * i) * TS), 16, 0, 0);
            } else {
              const int r = 8 * w + 2 * i + (lane >> 5);
              const unsigned off = __umul24((unsigned)(ct * TS + r), (unsigned)ldw) + (unsigned)(kk + 4 * ((lane & 31) ^ (r & 15)));
              __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)off), (lptr_t)(imgW + (8 * w + 2 * i) * DKC), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][e], fw[q][e], acc, 0, 0, 0);
      }
    }
    after1 = after2;
    after2 = more ? 8 : 0;
  }
#undef FL_ISSUE_A
  FL_STAMP(1);

  // meet in LDS: sm[w][row][col] in ring stage 3 (stages 0..2 take the next tile's W chunks meanwhile); C/D layout of
  // 32x32x2: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  __syncthreads();  // every wave has left the loop: no stage is being read
  float* part = smem + 3 * DSTAGE;
  float* mine = part + w * (TS * SLD);
#pragma unroll
  for (int r = 0; r < 16; ++r) *(lf1_t*)(mine + ((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + lr) = acc[r];
  __syncthreads();
  FL_STAMP(10);
  const float* s0 = part + er * SLD + ec;
  f32x4v vv = *(const lf4_t*)(s0);
#pragma unroll
  for (int k = 1; k < 4; ++k) vv += *(const lf4_t*)(s0 + k * (TS * SLD));
  float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
  const float gmul = jb.gate ? 1.f + fl::sigmoid_acc(e_gate) : 1.f;
  if (gr < jb.M) {
#pragma unroll
    for (int s = 0; s < FL_PROG_MAX_SEG; ++s) {
      if (jb.bias[s]) {
        const float sc = jb.bias_scale[s] * gmul;
        v.x += sc * e_b[s].x; v.y += sc * e_b[s].y; v.z += sc * e_b[s].z; v.w += sc * e_b[s].w;
      }
    }
    if (jb.aux[3]) st4g(wt, wp(jb.aux[3], io) + (long long)gr * jb.ldc + gc, v);  // pre-activation copy (saved for backward)
    if (actbwd) {
      v.x *= act_grad(e_p.x, jb.act); v.y *= act_grad(e_p.y, jb.act);
      v.z *= act_grad(e_p.z, jb.act); v.w *= act_grad(e_p.w, jb.act);
    } else {
      v.x = fl::apply_act(v.x, jb.act, 0.f);
      v.y = fl::apply_act(v.y, jb.act, 0.f);
      v.z = fl::apply_act(v.z, jb.act, 0.f);
      v.w = fl::apply_act(v.w, jb.act, 0.f);
    }
    if (jb.mask) {
      v.x *= e_m; v.y *= e_m; v.z *= e_m; v.w *= e_m;
    }
    if (jb.R) {
      v.x += e_r.x; v.y += e_r.y; v.z += e_r.z; v.w += e_r.w;
    }
    st4g(wt, wp(jb.C, io) + (long long)gr * jb.ldc + gc, v);
  }
  // the next tile's first W chunks: behind this tile's stores in issue order, so a counted wait on the stores leaves
  // them in flight across the barrier
  int issued = 0;
  if (pf_job) issued = prefetch_w(pf_job, pf_ct, smem);
  __syncthreads();  // the partial tiles (stage 3) are free again
  return issued;
}

// Row LayerNorm (+mask, act, residual, gated extra) and an optional second LayerNorm of the result; one wave per row.
__device__ __forceinline__ void rownorm_tile(const fl_pjob& jb, int row4, const ProgIO& io, const bool wt) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int row = row4 + w;
  if (row >= jb.M) return;
  const int D = jb.N, nv = D >> 8;  // float4 per lane (D % 256 == 0, D <= 1024)
  const float* x = rp(jb.A[0], io) + (long long)row * jb.lda[0];
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      v[i] = ld4g(x + (i * 64 + lane) * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float m = fl::wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const float a = v[i].x - m, b = v[i].y - m, c = v[i].z - m, d = v[i].w - m;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float r = 1.0f / sqrtf(fl::wave_sum(q) / D + jb.eps);
  const float* w1 = rp(jb.W[0], io);
  const float* b1 = rp(jb.bias[0], io);
  const float* mk = jb.mask ? rp(jb.mask, io) + (long long)row * D : nullptr;
  const float* R = jb.R ? rp(jb.R, io) + (long long)row * jb.ldr : nullptr;
  const float* gx = jb.gate ? rp(jb.aux[1], io) + (long long)row * jb.lda[1] : nullptr;
  const float sg = jb.gate ? fl::sigmoid_acc(ld1g(rp(jb.gate, io))) : 0.f;
  float* y1 = jb.C ? wp(jb.C, io) + (long long)row * jb.ldc : nullptr;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const int o = (i * 64 + lane) * 4;
      const float4 ww = ld4g(w1 + o), bb = ld4g(b1 + o);
      float4 y;
      y.x = (v[i].x - m) * r * ww.x + bb.x;
      y.y = (v[i].y - m) * r * ww.y + bb.y;
      y.z = (v[i].z - m) * r * ww.z + bb.z;
      y.w = (v[i].w - m) * r * ww.w + bb.w;
      if (mk) {
        const float4 k = ld4g(mk + o);
        y.x *= k.x; y.y *= k.y; y.z *= k.z; y.w *= k.w;
      }
      y.x = fl::apply_act(y.x, jb.act, 0.f);
      y.y = fl::apply_act(y.y, jb.act, 0.f);
      y.z = fl::apply_act(y.z, jb.act, 0.f);
      y.w = fl::apply_act(y.w, jb.act, 0.f);
      if (R) {
        const float4 k = ld4g(R + o);
        y.x += k.x; y.y += k.y; y.z += k.z; y.w += k.w;
      }
      if (gx) {
        const float4 k = ld4g(gx + o);
        y.x += sg * k.x; y.y += sg * k.y; y.z += sg * k.z; y.w += sg * k.w;
      }
      if (y1) st4g(wt, y1 + o, y);
      v[i] = y;
      s2 += (y.x + y.y) + (y.z + y.w);
    }
  }
  if (!jb.W[1]) return;
  const float m2 = fl::wave_sum(s2) / D;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const float a = v[i].x - m2, b = v[i].y - m2, c = v[i].z - m2, d = v[i].w - m2;
      q2 += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float r2 = 1.0f / sqrtf(fl::wave_sum(q2) / D + jb.eps2);
  const float* w2 = rp(jb.W[1], io);
  const float* b2 = rp(jb.bias[1], io);
  float* y2 = wp(jb.aux[0], io) + (long long)row * jb.ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const int o = (i * 64 + lane) * 4;
      const float4 ww = ld4g(w2 + o), bb = ld4g(b2 + o);
      float4 y;
      y.x = (v[i].x - m2) * r2 * ww.x + bb.x;
      y.y = (v[i].y - m2) * r2 * ww.y + bb.y;
      y.z = (v[i].z - m2) * r2 * ww.z + bb.z;
      y.w = (v[i].w - m2) * r2 * ww.w + bb.w;
      st4g(wt, y2 + o, y);
    }
  }
}

// elementwise kinds: tile k of the rows [r0, r1)
__device__ __forceinline__ void sinusoid_tile(const fl_pjob& jb, int r0, int r1, int k, const ProgIO& io, const bool wt) {
  const int half = jb.N >> 1;
  const int i = r0 * half + k * 256 + threadIdx.x;
  if (i >= r1 * half) return;
  const int b = i / half, j = i - b * half;
  const long long* t = reinterpret_cast<const long long*>(rp(jb.A[0], io));
  const float a = (float)ldi64g(t + (long long)b * jb.lda[0]) * ld1g(rp(jb.W[0], io) + j);
  float* out = wp(jb.C, io) + (long long)b * jb.ldc;
  st1g(wt, out + j, sinf(a));
  st1g(wt, out + half + j, cosf(a));
}

__device__ __forceinline__ void gather_tile(const fl_pjob& jb, int r0, int r1, int k, const ProgIO& io, const bool wt) {
  const int i = r0 * jb.N + (k * 256 + threadIdx.x) * 4;
  if (i >= r1 * jb.N) return;
  const int b = i / jb.N, d = i - b * jb.N;
  const long long* idx = reinterpret_cast<const long long*>(rp(jb.W[0], io));
  long long ix = ldi64g(idx + b);
  if (jb.K[0] > 0 && (ix < 0 || ix >= jb.K[0])) {  // checked table (K[0] rows): never read outside it, void the run loudly
    __hip_atomic_store(io.abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ix = 0;
  }
  const float4 v = ld4g(rp(jb.A[0], io) + ix * jb.lda[0] + d);
  st4g(wt, wp(jb.C, io) + (long long)b * jb.ldc + d, v);
}

// ---- Philox4x32-10, the same stream as fl_philox_fill (elementwise.hip): 4 normals per counter -------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint64_t ctr) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u1 = u01(c[0]), u2 = u01(c[1]), u3 = u01(c[2]), u4 = u01(c[3]);
  const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
  float s1, c1, s2, c2;
  sincosf(6.283185307179586f * u2, &s1, &c1);
  sincosf(6.283185307179586f * u4, &s2, &c2);
  return make_float4(ra * c1, ra * s1, rb * c2, rb * s2);
}

// DDPM reverse update (v2:585-592; same arithmetic as p_sample_kernel in elementwise.hip), float4 per thread.
__device__ __forceinline__ void psample_tile(const fl_pjob& jb, int r0, int r1, int k, const ProgIO& io, const bool wt) {
  const long long i4 = ((long long)r0 * jb.N >> 2) + (long long)k * 256 + threadIdx.x;  // float4 index == Philox counter offset
  if (i4 * 4 >= (long long)r1 * jb.N) return;
  const int t = (int)ldi64g(reinterpret_cast<const long long*>(rp(jb.aux[2], io)));
  const float a = ld1g(rp(jb.W[0], io) + t), ab = ld1g(rp(jb.W[1], io) + t), be = ld1g(rp(jb.W[2], io) + t);
  const float coef = (1.f - a) / sqrtf(1.f - ab), sa = sqrtf(a), sg = sqrtf(be);
  const float4 x = ld4g(rp(jb.A[0], io) + i4 * 4), e = ld4g(rp(jb.A[1], io) + i4 * 4);
  float4 m;
  m.x = (x.x - coef * e.x) / sa;
  m.y = (x.y - coef * e.y) / sa;
  m.z = (x.z - coef * e.z) / sa;
  m.w = (x.w - coef * e.w) / sa;
  if (t > 0) {
    float4 z;
    if (jb.A[2]) {
      z = ld4g(rp(jb.A[2], io) + i4 * 4);
    } else {
      const uint64_t seed = (uint64_t)(unsigned)jb.K[0] | ((uint64_t)(unsigned)jb.K[1] << 32);
      const uint64_t base = (uint64_t)ldi64g(reinterpret_cast<const long long*>(rp(jb.aux[0], io)));
      z = philox_normal4(seed, base + (uint64_t)i4);
    }
    m.x += sg * z.x; m.y += sg * z.y; m.z += sg * z.z; m.w += sg * z.w;
  }
  st4g(wt, wp(jb.C, io) + i4 * 4, m);
}

// ---------------------------------------------------------------------------------------------- training kinds
// Row LayerNorm backward (same arithmetic as layernorm_bwd_dx_kernel in norm.hip); mean / rstd are recomputed from the
// saved input exactly as rownorm_tile computed them in the forward. One wave per row; v = x row, g = dy row on entry,
// g = dx (without the residual) on return.
__device__ __forceinline__ void ln_bwd_row(float4 (&v)[4], float4 (&g)[4], const float* w1, const float* b1, const float* mk,
                                           int act, float eps, int D, int nv, int lane, float* glo, float& m_out, float& r_out, const bool wt) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float m = fl::wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const float a = v[i].x - m, b = v[i].y - m, c = v[i].z - m, d = v[i].w - m;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float r = 1.0f / sqrtf(fl::wave_sum(q) / D + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const int o = (i * 64 + lane) * 4;
      const float4 ww = ld4g(w1 + o), bb = ld4g(b1 + o);
      float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
      if (mk) k = ld4g(mk + o);
      float4 xh, gl;
      xh.x = (v[i].x - m) * r; xh.y = (v[i].y - m) * r; xh.z = (v[i].z - m) * r; xh.w = (v[i].w - m) * r;
      gl.x = g[i].x * act_grad(k.x * (xh.x * ww.x + bb.x), act) * k.x;
      gl.y = g[i].y * act_grad(k.y * (xh.y * ww.y + bb.y), act) * k.y;
      gl.z = g[i].z * act_grad(k.z * (xh.z * ww.z + bb.z), act) * k.z;
      gl.w = g[i].w * act_grad(k.w * (xh.w * ww.w + bb.w), act) * k.w;
      if (glo) st4g(wt, glo + o, gl);
      g[i].x = gl.x * ww.x; g[i].y = gl.y * ww.y; g[i].z = gl.z * ww.z; g[i].w = gl.w * ww.w;  // d xhat
      v[i] = xh;
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * xh.x + g[i].y * xh.y) + (g[i].z * xh.z + g[i].w * xh.w);
    }
  }
  const float t1 = fl::wave_sum(s1) / D, t2 = fl::wave_sum(s2) / D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      g[i].x = r * (g[i].x - t1 - v[i].x * t2);
      g[i].y = r * (g[i].y - t1 - v[i].y * t2);
      g[i].z = r * (g[i].z - t1 - v[i].z * t2);
      g[i].w = r * (g[i].w - t1 - v[i].w * t2);
    }
  }
  m_out = m;
  r_out = r;
}

// FL_PJOB_LNBWD. With A[2] (a second, inner LayerNorm whose OUTPUT + R-path fed the first one's input, v2:514-516 followed
// by the layer norm of v2:549): stage 1 is the plain (no mask / activation) backward of LN(W[1], bias[1]) on x = A[0] plus
// the residual R, stored to aux[2]; stage 2 continues in registers with x = A[2], W[0], bias[0], mask, act.
__device__ __forceinline__ void lnbwd_tile(const fl_pjob& jb, int row4, const ProgIO& io, const bool wt) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int row = row4 + w;
  if (row >= jb.M) return;
  const int D = jb.N, nv = D >> 8;
  const float* x = rp(jb.A[0], io) + (long long)row * jb.lda[0];
  const float* dy = rp(jb.A[1], io) + (long long)row * jb.lda[1];
  const float* mk = jb.mask ? rp(jb.mask, io) + (long long)row * D : nullptr;
  const float* R = jb.R ? rp(jb.R, io) + (long long)row * jb.ldr : nullptr;
  const bool two = jb.A[2] != 0;
  float4 v[4], g[4], v2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      v[i] = ld4g(x + (i * 64 + lane) * 4);
      g[i] = ld4g(dy + (i * 64 + lane) * 4);
      if (two) v2[i] = ld4g(rp(jb.A[2], io) + (long long)row * jb.lda[2] + (i * 64 + lane) * 4);
    }
  }
  float m, r;
  if (two) {
    ln_bwd_row(v, g, rp(jb.W[1], io), rp(jb.bias[1], io), nullptr, FL_ACT_NONE, jb.eps2, D, nv, lane, nullptr, m, r, wt);
    float* mid = wp(jb.aux[2], io) + (long long)row * jb.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nv) {
        const int o = (i * 64 + lane) * 4;
        if (R) {
          const float4 k = ld4g(R + o);
          g[i].x += k.x; g[i].y += k.y; g[i].z += k.z; g[i].w += k.w;
        }
        st4g(wt, mid + o, g[i]);
        v[i] = v2[i];
      }
    }
    if (jb.aux[3] && lane == 0) {
      float* st = wp(jb.aux[3], io) + 2 * (long long)row;
      st1g(wt, st, m);
      st1g(wt, st + 1, r);
    }
    R = nullptr;
  }
  float* glo = jb.aux[0] ? wp(jb.aux[0], io) + (long long)row * jb.ldc : nullptr;
  ln_bwd_row(v, g, rp(jb.W[0], io), rp(jb.bias[0], io), mk, jb.act, jb.eps, D, nv, lane, glo, m, r, wt);
  float* dx = wp(jb.C, io) + (long long)row * jb.ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const int o = (i * 64 + lane) * 4;
      float4 y = g[i];
      if (R) {
        const float4 k = ld4g(R + o);
        y.x += k.x; y.y += k.y; y.z += k.z; y.w += k.w;
      }
      st4g(wt, dx + o, y);
    }
  }
  if (jb.aux[1] && lane == 0) {
    float* st = wp(jb.aux[1], io) + 2 * (long long)row;
    st1g(wt, st, m);
    st1g(wt, st + 1, r);
  }
}

// LayerNorm weight / bias gradients: 32 columns per tile. A lane owns 4 columns of every 32nd row (a wave's load covers
// 8 rows x 128 B), the 32 row groups meet in an LDS tree. The row statistics are staged in LDS (coherent loads).
__device__ __forceinline__ void lncol_tile(const fl_pjob& jb, int k, const ProgIO& io, float* smem, const bool wt) {
  const int c4 = (threadIdx.x & 7) * 4, rg = threadIdx.x >> 3;
  const float* gl = rp(jb.A[0], io) + k * 32 + c4;
  const float* x = rp(jb.A[1], io) + k * 32 + c4;
  const float* st = rp(jb.aux[1], io);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  for (int base = 0; base < jb.M; base += 1024) {
    const int rows = jb.M - base < 1024 ? jb.M - base : 1024;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * rows; i += 256) *(lf1_t*)(smem + i) = ld1c(st + 2 * base + i);
    __syncthreads();
#pragma unroll 8
    for (int m = rg; m < rows; m += 32) {
      const float4 gv = ld4g(gl + (long long)(base + m) * jb.lda[0]), xv = ld4g(x + (long long)(base + m) * jb.lda[1]);
      const float mu = *(const lf1_t*)(smem + 2 * m), rs = *(const lf1_t*)(smem + 2 * m + 1);
      aw.x += gv.x * ((xv.x - mu) * rs); aw.y += gv.y * ((xv.y - mu) * rs);
      aw.z += gv.z * ((xv.z - mu) * rs); aw.w += gv.w * ((xv.w - mu) * rs);
      ab.x += gv.x; ab.y += gv.y; ab.z += gv.z; ab.w += gv.w;
    }
  }
  __syncthreads();
  // [which][row group 32][32 columns]
  f32x4v t0 = {aw.x, aw.y, aw.z, aw.w}, t1 = {ab.x, ab.y, ab.z, ab.w};
  *(lf4_t*)(smem + rg * 32 + c4) = t0;
  *(lf4_t*)(smem + 1024 + rg * 32 + c4) = t1;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) t += *(const lf1_t*)(smem + which * 1024 + g * 32 + c);
    st1g(wt, wp(which ? jb.aux[0] : jb.C, io) + k * 32 + c, t);
  }
  __syncthreads();
}

// euclidean_distance_loss rows (v2:293-302; euclid_norm_kernel / euclid_bwd_kernel of loss_optim.hip): one wave per row.
__device__ __forceinline__ void loss_tile(const fl_pjob& jb, int row4, const ProgIO& io, const bool wt) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int row = row4 + w;
  if (row >= jb.M) return;
  const int D = jb.N, nv = D >> 8;
  const float* a = rp(jb.A[0], io) + (long long)row * jb.lda[0];
  const float* b = rp(jb.A[1], io) + (long long)row * jb.lda[1];
  const float* rr = jb.A[2] ? rp(jb.A[2], io) + (long long)row * jb.lda[2] : nullptr;
  float4 d[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const float4 p = ld4g(a + (i * 64 + lane) * 4), t = ld4g(b + (i * 64 + lane) * 4);
      d[i] = make_float4(p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w);
      s += (d[i].x * d[i].x + d[i].y * d[i].y) + (d[i].z * d[i].z + d[i].w * d[i].w);
    }
  }
  const float nrm = sqrtf(fl::wave_sum(s) + 1e-8f);
  float dot = 0.f;
  float* out = wp(jb.C, io) + (long long)row * jb.ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const int o = (i * 64 + lane) * 4;
      float4 y;
      y.x = jb.alpha * d[i].x / nrm; y.y = jb.alpha * d[i].y / nrm; y.z = jb.alpha * d[i].z / nrm; y.w = jb.alpha * d[i].w / nrm;
      st4g(wt, out + o, y);
      if (rr) {
        const float4 k = ld4g(rr + o);
        dot += (y.x * k.x + y.y * k.y) + (y.z * k.z + y.w * k.w);
      }
    }
  }
  dot = fl::wave_sum(dot);
  if (lane == 0) {
    st1g(wt, wp(jb.aux[0], io) + row, nrm);
    if (jb.aux[1]) st1g(wt, wp(jb.aux[1], io) + row, dot);
  }
}

__device__ __forceinline__ float block_sum_lds(float v, float* smem) {
  v = fl::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) *(lf1_t*)(smem + (threadIdx.x >> 6)) = v;
  __syncthreads();
  const float t = (*(const lf1_t*)(smem) + *(const lf1_t*)(smem + 1)) + (*(const lf1_t*)(smem + 2) + *(const lf1_t*)(smem + 3));
  __syncthreads();
  return t;
}

// loss = mean of the row norms; gradient of the residual gate = s(1-s) * sum of the row dots (gate_grad_kernel)
__device__ __forceinline__ void fin_tile(const fl_pjob& jb, const ProgIO& io, float* smem, const bool wt) {
  float a = 0.f, b = 0.f;
  const float* n = rp(jb.A[0], io);
  const float* d = jb.A[1] ? rp(jb.A[1], io) : nullptr;
  for (int i = threadIdx.x; i < jb.M; i += 256) {
    a += ld1c(n + i);
    if (d) b += ld1c(d + i);
  }
  a = block_sum_lds(a, smem);
  b = block_sum_lds(b, smem);
  if (threadIdx.x == 0) {
    st1g(wt, wp(jb.C, io), a / jb.M);
    if (d) {
      const float sg = fl::sigmoid_acc(ld1g(rp(jb.gate, io)));
      st1g(wt, wp(jb.aux[0], io), sg * (1.f - sg) * b);
    }
  }
}

__device__ __forceinline__ void ewadd_tile(const fl_pjob& jb, int r0, int r1, int k, const ProgIO& io, const bool wt) {
  const int i = r0 * jb.N + (k * 256 + threadIdx.x) * 4;
  if (i >= r1 * jb.N) return;
  const int b = i / jb.N, d = i - b * jb.N;
  float4 v = ld4g(rp(jb.A[0], io) + (long long)b * jb.lda[0] + d);
  const float4 u = ld4g(rp(jb.A[1], io) + (long long)b * jb.lda[1] + d);
  v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  if (jb.A[2]) {
    const float4 t = ld4g(rp(jb.A[2], io) + (long long)b * jb.lda[2] + d);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  st4g(wt, wp(jb.C, io) + (long long)b * jb.ldc + d, v);
}

// deterministic embedding gradient: tile = table row r; every wave ballots the batch's indices against r (64 at a time) and
// adds the matching rows in batch order
__device__ __forceinline__ void embbwd_tile(const fl_pjob& jb, int r, const ProgIO& io, float* smem, const bool wt) {
  const long long* idx = reinterpret_cast<const long long*>(rp(jb.W[0], io));
  const float* src = rp(jb.A[0], io);
  const int D = jb.N, nv = (D + 1023) >> 10;  // float4 per thread (D <= 4096)
  const int lane = threadIdx.x & 63;
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < jb.M; b0 += 64) {
    const bool hit = b0 + lane < jb.M && ldi64g(idx + (b0 + lane < jb.M ? b0 + lane : 0)) == (long long)r;
    unsigned long long mask = __ballot(hit);
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = (i * 256 + threadIdx.x) * 4;
        if (i < nv && o < D) {
          const float4 t = ld4g(src + (long long)(b0 + j) * jb.lda[0] + o);
          acc[i].x += t.x; acc[i].y += t.y; acc[i].z += t.z; acc[i].w += t.w;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = (i * 256 + threadIdx.x) * 4;
    if (i < nv && o < D) st4g(wt, wp(jb.C, io) + (long long)r * jb.ldc + o, acc[i]);
  }
}

constexpr int FLAT_TILES = 256;  // SQNORM / ADAM: the flat buffer is cut into this many slices
__device__ __forceinline__ void flat_slice(int n, int t, int& lo, int& hi) {
  const int per = ((n + FLAT_TILES - 1) / FLAT_TILES + 1023) & ~1023;  // multiple of 256 threads x float4
  lo = t * per < n ? t * per : n;
  hi = lo + per < n ? lo + per : n;
}

__device__ __forceinline__ void sqnorm_tile(const fl_pjob& jb, int t, const ProgIO& io, float* smem, const bool wt) {
  const float* g = rp(jb.A[0], io);
  int lo, hi;
  flat_slice(jb.K[0], t, lo, hi);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = lo + threadIdx.x * 4;
  for (; i + 3 * 1024 + 3 < hi; i += 4096) {  // four float4 in flight
    const float4 p = ld4g(g + i), q = ld4g(g + i + 1024), r = ld4g(g + i + 2048), u = ld4g(g + i + 3072);
    a0 += (p.x * p.x + p.y * p.y) + (p.z * p.z + p.w * p.w);
    a1 += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
    a2 += (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
    a3 += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
  }
  for (; i < hi; i += 1024) {  // n % 4 == 0 (validated): whole float4 only
    const float4 p = ld4g(g + i);
    a0 += (p.x * p.x + p.y * p.y) + (p.z * p.z + p.w * p.w);
  }
  const float tot = block_sum_lds((a0 + a1) + (a2 + a3), smem);
  if (threadIdx.x == 0) st1g(wt, wp(jb.C, io) + t, tot);
}

// clip_grad_norm_ + Adam / AdamW: adam_kernel of loss_optim.hip with the clip coefficient derived in every tile from
// the SQNORM partials (the same 256 numbers summed in the same order: identical in every tile)
__device__ __forceinline__ void adam_tile(const fl_pjob& jb, int t, const ProgIO& io, float* smem) {
  const float part = ld1c(rp(jb.W[0], io) + threadIdx.x);  // FLAT_TILES == 256 threads
  const float sq = block_sum_lds(part, smem);
  const float gs = jb.alpha;
  const float nrm = gs * sqrtf(sq);
  const float maxn = jb.bias_scale[2];
  float coef = 1.f;
  if (maxn > 0.f) {
    const float cf = maxn / (nrm + 1e-6f);
    coef = cf < 1.f ? cf : 1.f;
  }
  if (t == 0 && threadIdx.x == 0) {
    if (jb.aux[2]) *(__attribute__((address_space(1))) float*)wp(jb.aux[2], io) = coef;
    if (jb.aux[3]) *(__attribute__((address_space(1))) float*)wp(jb.aux[3], io) = nrm;
  }
  const float lr = ld1g(rp(jb.W[1], io));
  const float step = (float)(*(const __attribute__((address_space(1))) int*)rp(jb.W[2], io) + 1);
  const float b1 = jb.bias_scale[0], b2 = jb.bias_scale[1], eps = jb.eps, wd = jb.eps2;
  const float bc1 = 1.f - powf(b1, step);
  const float bc2s = sqrtf(1.f - powf(b2, step));
  const float cs = gs * coef;
  const float step_size = lr / bc1;
  const bool decoupled = (jb.flags & FL_PG_DECOUPLED) != 0;
  const float decay = decoupled ? 1.f - lr * wd : 1.f;
  const float l2 = decoupled ? 0.f : wd;
  int lo, hi;
  flat_slice(jb.K[0], t, lo, hi);
  const float* g = rp(jb.A[0], io);
  float* p = wp(jb.C, io);
  float* m = wp(jb.aux[0], io);
  float* v = wp(jb.aux[1], io);
#define FL_ADAM1(G, P, Mv, Vv)                                   \
  do {                                                           \
    float gi_ = (G) * cs;                                        \
    float pi_ = (P) * decay;                                     \
    gi_ += l2 * (P);                                             \
    const float mi_ = b1 * (Mv) + (1.f - b1) * gi_;              \
    const float vi_ = b2 * (Vv) + (1.f - b2) * gi_ * gi_;        \
    (Mv) = mi_;                                                  \
    (Vv) = vi_;                                                  \
    (P) = pi_ - step_size * (mi_ / (sqrtf(vi_) / bc2s + eps));   \
  } while (0)
  for (int i = lo + threadIdx.x * 4; i < hi; i += 2048) {  // two float4 of each of the four arrays in flight
    const bool two = i + 1024 < hi;
    const int j = two ? i + 1024 : i;
    float4 g0 = ld4g(g + i), p0 = ld4g(p + i), m0 = ld4g(m + i), v0 = ld4g(v + i);
    float4 g1 = ld4g(g + j), p1 = ld4g(p + j), m1 = ld4g(m + j), v1 = ld4g(v + j);
    FL_ADAM1(g0.x, p0.x, m0.x, v0.x); FL_ADAM1(g0.y, p0.y, m0.y, v0.y);
    FL_ADAM1(g0.z, p0.z, m0.z, v0.z); FL_ADAM1(g0.w, p0.w, m0.w, v0.w);
    st4g(false, p + i, p0); st4g(false, m + i, m0); st4g(false, v + i, v0);
    if (two) {
      FL_ADAM1(g1.x, p1.x, m1.x, v1.x); FL_ADAM1(g1.y, p1.y, m1.y, v1.y);
      FL_ADAM1(g1.z, p1.z, m1.z, v1.z); FL_ADAM1(g1.w, p1.w, m1.w, v1.w);
      st4g(false, p + j, p1); st4g(false, m + j, m1); st4g(false, v + j, v1);
    }
  }
#undef FL_ADAM1
}

// ---------------------------------------------------------------------------------------------- barriers
// `sync` layout (uint32 words; fl_program_sync_bytes): arrival counters of the device-wide barrier sync[g * CTR_STRIDE]
// (g = block & 7; monotone over the run), departures sync[DEP_WORD], abort word sync[ABORT_WORD], launch epoch
// sync[EPOCH_WORD], blocks registered on XCD x sync[REG_OFF + 32 x], flag line of XCD x sync[x * CTR_STRIDE + FLAG_OFF ..+32).
constexpr int DEP_WORD = NCTR * CTR_STRIDE;
constexpr int ABORT_WORD = NCTR * CTR_STRIDE + 16;
constexpr int EPOCH_WORD = NCTR * CTR_STRIDE + 32;
constexpr int REG_OFF = NCTR * CTR_STRIDE + 128;
constexpr int FLAG_OFF = 256;
static_assert(ABORT_WORD == FL_PROG_SYNC_ABORT_WORD, "abort word");

// s_waitcnt vmcnt(pending), then the workgroup barrier: every wave's stores have left the CU, `pending` prefetch DMAs
// per wave (issued after the stores) stay in flight
__device__ __forceinline__ void drain_stores(int pending) {
  wait_vm(pending);
  __syncthreads();
}

// Device-wide barrier for a fully resident grid. Returns false if the run is dead (timeout here or in another block):
// every wave then leaves. release: the block's XCD holds dirty lines of plain-stored outputs (local phases) that blocks
// of other XCDs are about to read: write the L2 back first.
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned target, bool acquire, bool release, int pending, int* s_ok) {
  drain_stores(pending);
  if (threadIdx.x == 0) {
    FL_STAMP(3);
    if (release) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop its own wait after buffer_wbl2)
    }
    FL_STAMP(4);
    // Polling is relaxed (plain sc1 loads): an acquire per poll would invalidate the L2 on every iteration.
    __hip_atomic_fetch_add(&sync[(blockIdx.x & (NCTR - 1)) * CTR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FL_STAMP(5);
    unsigned spins = 0;
    int ok = 1;
    for (;;) {
      unsigned tot = 0;
#pragma unroll
      for (int g = 0; g < NCTR; ++g) tot += __hip_atomic_load(&sync[g * CTR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tot >= target) break;
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 255u) == 0) {
        const unsigned word = __hip_atomic_load(&sync[ABORT_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (word != 0 || spins > SPIN_LIMIT) {  // the run is void: keep the first reason
          if (word == 0) __hip_atomic_store(&sync[ABORT_WORD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    // The invalidating acquire is only needed when the next phase re-reads a buffer that was overwritten after an
    // earlier read in this launch (stale L1/L2 lines): the program builder knows; a launch starts with clean caches.
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

// Barrier among the blocks of ONE XCD (between two local phases). There are no L2-side atomics to be had on this memory
// (hipcc emits the same memory-side global_atomic_add for workgroup and agent scope), so the blocks use FLAGS: block
// `slot` stores `value` (launch epoch << 12 | local barrier count: monotone over launches, nothing to reset) into its
// word of the XCD's 128-byte flag line with a PLAIN store - the line lives in this XCD's L2, like the data - and lanes
// 0..nx-1 of wave 0 poll the line with sc1 loads (L1 bypassed, served by that L2). tools/xcd_handoff_probe.hip: a phase
// that stores 4 KB per block, passes this barrier and reads two other blocks' slices takes 1.04-1.09 us (1.6 us with
// memory-side arrival atomics, 2.7 us with the device-wide barrier and write-through stores), no wrong word.
__device__ __forceinline__ bool xcd_barrier(unsigned* sync, int xcc, int slot, int nx, unsigned value, bool acquire, int pending, int* s_ok) {
  drain_stores(pending);
  unsigned* line = &sync[xcc * CTR_STRIDE + FLAG_OFF];
  if (threadIdx.x < 64) {
    FL_STAMP(3);
    FL_STAMP(4);
    if (threadIdx.x == 0) *(__attribute__((address_space(1))) unsigned*)(line + slot) = value;
    FL_STAMP(5);
    const int lane = threadIdx.x;
    unsigned spins = 0;
    int ok = 1;
    for (;;) {
      unsigned v = value;
      if (lane < nx) v = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_ballot_w64((int)(v - value) < 0) == 0) break;
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 255u) == 0) {
        const unsigned word = __hip_atomic_load(&sync[ABORT_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (word != 0 || spins > SPIN_LIMIT) {
          if (word == 0 && lane == 0) __hip_atomic_store(&sync[ABORT_WORD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    if (lane == 0) {
      if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      *s_ok = ok;
    }
  }
  __syncthreads();
  return *s_ok != 0;
}

// ---------------------------------------------------------------------------------------------- tile dealing
// a tile of the current phase as this block sees it
struct Cur {
  int j;     // job index (-1: none)
  int t;     // local phases: tile index among the job's tiles of this XCD; device-wide phases: tile index in the phase
  int base;  // local phases: tiles of this XCD in the phase's earlier jobs
  int cnt;   // local phases: tiles of this XCD in job j
};
struct Deal {
  const int* lph;  // phases, LDS
  const int* ljb;  // jobs, LDS
  int xcc, slot, nx, G;
};
constexpr int JW = sizeof(fl_pjob) / 4;

__device__ __forceinline__ int job_word(const Deal& d, int j, int word) { return lds_word(d.ljb + j * JW, word); }

// local phase: the first tile at or after (job j, XCD-tile t >= 0 given by the slot rule) that belongs to this block
__device__ __forceinline__ Cur local_seek(const Deal& d, int j, int job_end, int base) {
  Cur c;
  for (; j < job_end; ++j) {
    const int nrt = job_word(d, j, FL_JW(nrt)), per = job_word(d, j, FL_JW(per));
    const int mine = nrt > d.xcc ? (nrt - d.xcc + NCTR - 1) / NCTR : 0;
    const int cnt = mine * per;
    // XCD-tile t of this job runs on the block with slot == (base + t) % nx
    int t = (d.slot - base) % d.nx;
    t = t < 0 ? t + d.nx : t;
    if (t < cnt) {
      c.j = j; c.t = t; c.base = base; c.cnt = cnt;
      return c;
    }
    base += cnt;
  }
  c.j = -1; c.t = 0; c.base = base; c.cnt = 0;
  return c;
}
__device__ __forceinline__ Cur global_at(const Deal& d, int p, int t) {
  Cur c;
  c.base = 0; c.cnt = 0; c.t = t;
  const int job_begin = lds_word(d.lph + 4 * p, 0), job_end = lds_word(d.lph + 4 * p, 1), ntiles = lds_word(d.lph + 4 * p, 2);
  if (t >= ntiles) {
    c.j = -1;
    return c;
  }
  int j = job_begin;
  while (j + 1 < job_end && t >= job_word(d, j + 1, FL_JW(tile_begin))) ++j;
  c.j = j;
  return c;
}
__device__ __forceinline__ Cur first_tile(const Deal& d, int p, bool local) {
  if (local) return local_seek(d, lds_word(d.lph + 4 * p, 0), lds_word(d.lph + 4 * p, 1), 0);
  return global_at(d, p, (int)blockIdx.x);
}
__device__ __forceinline__ Cur next_tile(const Deal& d, int p, bool local, const Cur& c) {
  if (!local) return global_at(d, p, c.t + d.G);
  if (c.t + d.nx < c.cnt) {
    Cur n = c;
    n.t = c.t + d.nx;
    return n;
  }
  return local_seek(d, c.j + 1, lds_word(d.lph + 4 * p, 1), c.base + c.cnt);
}
// (row tile, tile inside the row tile) of a dealt tile
__device__ __forceinline__ void tile_coords(const Deal& d, bool local, const Cur& c, int& rt, int& k) {
  const int nrt = job_word(d, c.j, FL_JW(nrt)), per = job_word(d, c.j, FL_JW(per));
  if (local) {
    const int rl = c.t / per;
    k = c.t - rl * per;
    rt = d.xcc + NCTR * rl;
    return;
  }
  const int lt = c.t - job_word(d, c.j, FL_JW(tile_begin));
  if (job_word(d, c.j, FL_JW(kind)) == FL_PJOB_GEMM && (per & 7) == 0) {
    // tile lt runs on XCD lt%8 (round-robin block placement): keep a weight column block on one XCD
    const int x = lt & 7, jj = lt >> 3, jc = jj / nrt;
    k = x + 8 * jc;
    rt = jj - jc * nrt;
  } else {
    rt = lt / per;
    k = lt - rt * per;
  }
}
__device__ __forceinline__ bool prefetchable(const Deal& d, const Cur& c) {
  if (c.j < 0) return false;
  const int need = FL_PG_DMA | FL_PG_WCONST;
  return job_word(d, c.j, FL_JW(kind)) == FL_PJOB_GEMM && (job_word(d, c.j, FL_JW(flags)) & need) == need;
}

// returns the prefetch DMA instructions per wave in flight for the block's next tile
__device__ __forceinline__ int run_tile(const fl_pjob& jb, int rt, int k, const ProgIO& io, float* smem, const bool wt, const int pre,
                                        const int* my_job, const int* pf_job, const int pf_ct) {
  const int r0 = rt * TS, r1 = r0 + TS < jb.M ? r0 + TS : jb.M;
  switch (jb.kind) {
    case FL_PJOB_GEMM:
      if (jb.flags & FL_PG_DMA) {
        if (jb.flags & FL_PG_TW) return gemm_dma_tile<true>(jb, rt, k, io, smem, wt, pre, my_job, pf_job, pf_ct);
        return gemm_dma_tile<false>(jb, rt, k, io, smem, wt, pre, my_job, pf_job, pf_ct);
      }
      switch (jb.flags & (FL_PG_TA | FL_PG_TW)) {
        case 0: gemm_tile<false, false>(jb, rt, k, io, smem, wt); break;
        case FL_PG_TW: gemm_tile<false, true>(jb, rt, k, io, smem, wt); break;
        case FL_PG_TA | FL_PG_TW:
          if (jb.flags & FL_PG_T64) wg64_tile(jb, rt, k, io, smem, wt);
          else gemm_tile<true, true>(jb, rt, k, io, smem, wt);
          break;
        default: break;  // TA alone: rejected by fl_program_job_tiles
      }
      break;
    case FL_PJOB_ROWNORM: rownorm_tile(jb, r0 + 4 * k, io, wt); break;
    case FL_PJOB_SINUSOID: sinusoid_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_GATHER: gather_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_PSAMPLE: psample_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_LNBWD: lnbwd_tile(jb, r0 + 4 * k, io, wt); break;
    case FL_PJOB_LNCOL: lncol_tile(jb, k, io, smem, wt); break;
    case FL_PJOB_LOSS: loss_tile(jb, r0 + 4 * k, io, wt); break;
    case FL_PJOB_FIN: fin_tile(jb, io, smem, wt); break;
    case FL_PJOB_EWADD: ewadd_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_EMBBWD: embbwd_tile(jb, k, io, smem, wt); break;
    case FL_PJOB_SQNORM: sqnorm_tile(jb, k, io, smem, wt); break;
    case FL_PJOB_ADAM: adam_tile(jb, k, io, smem); break;
    default: break;
  }
  return 0;
}

// Device-wide phases: any block may run any tile, outputs are stored write-through, the phase ends in the device-wide
// barrier. Local phases (FL_PPHASE_LOCAL, honoured when the launch asked for FL_PROG_XCD_LOCAL and the grid is spread
// evenly over the XCDs): the 32-row tiles of the batch are dealt to the XCDs (row tile rt belongs to XCD rt % 8, the XCD
// being what the block READS from HW_REG_XCC_ID, not an assumption about placement), a block only ever runs tiles of its
// own XCD's rows, data is handed from phase to phase through that XCD's L2 (plain stores, loads that hit L2 instead of
// memory) and two local phases are separated by xcd_barrier among the 32 blocks of the XCD. Every XCD then streams all the
// weights (8 x the weight traffic, out of the 256 MB Infinity Cache): the first weight chunks of a block's next tile are
// requested BEFORE the barrier (gemm_dma_tile / prefetch_w).
__global__ __launch_bounds__(256, 1) void program_kernel(const fl_pphase* __restrict__ phases, const fl_pjob* __restrict__ jobs,
                                                         int nphases, int njobs, const ProgIO io, unsigned* sync, int allow_local) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_ok, s_slot;
  __shared__ unsigned s_epoch;
  // the program lives in LDS for the whole run
  int* lprog = reinterpret_cast<int*>(smem + 2 * STAGE);
  {
    const int nph_w = nphases * (int)(sizeof(fl_pphase) / 4);
    const int* gp = reinterpret_cast<const int*>(phases);
    for (int i = threadIdx.x; i < nph_w; i += 256) ((__attribute__((address_space(3))) int*)lprog)[i] = gp[i];
    // jobs: 64-bit words; the leading pointer fields that name an io slot (1..FL_PROG_MAX_IO) are bound here, once,
    // so that no tile ever waits on a kernel-argument fetch
    constexpr int JQ = sizeof(fl_pjob) / 8, PTRQ = offsetof(fl_pjob, kind) / 8;
    const unsigned long long* gj = reinterpret_cast<const unsigned long long*>(jobs);
    __attribute__((address_space(3))) unsigned long long* lj = (__attribute__((address_space(3))) unsigned long long*)(lprog + nph_w);
    for (int i = threadIdx.x; i < njobs * JQ; i += 256) {
      unsigned long long v = gj[i];
      if (i % JQ < PTRQ && v >= 1 && v <= FL_PROG_MAX_IO) v = reinterpret_cast<unsigned long long>(io.p[v - 1]);
      lj[i] = v;
    }
  }
  const unsigned G = gridDim.x;
  Deal d;
  d.lph = lprog;
  d.ljb = lprog + nphases * (int)(sizeof(fl_pphase) / 4);
  d.G = (int)G;
  d.xcc = 0;
  d.slot = 0;
  d.nx = 1;
  // a run that timed out left the abort word set and the arrival counters un-armed: every later launch must do NOTHING
  // (its barriers would fall through on the stale counters and the phases would compute garbage) until the host has
  // seen the word, re-armed the counters and raised (Program.check / Program.run)
  if (__hip_atomic_load(&sync[ABORT_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  bool alive = true;
  if (allow_local) {
    // which XCD am I on, and which of its blocks: the XCD is read from the hardware, the slot is the order of registration.
    // A grid of G = 8 nx blocks at one block per CU has nx blocks on every XCD; a block that finds itself past that (or
    // an XCD that never fills up: its first barrier times out) voids the run loudly instead of computing on a wrong deal.
    d.xcc = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (NCTR - 1);  // HW_REG_XCC_ID[3:0]
    d.nx = (int)(G / NCTR);
    if (threadIdx.x == 0) {
      s_slot = (int)__hip_atomic_fetch_add(&sync[REG_OFF + 32 * d.xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_epoch = __hip_atomic_load(&sync[EPOCH_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  unsigned epoch = 0;
  if (allow_local) {
    d.slot = __builtin_amdgcn_readfirstlane(s_slot);
    epoch = __builtin_amdgcn_readfirstlane(s_epoch);
    if (d.slot >= d.nx) {
      if (threadIdx.x == 0) __hip_atomic_store(&sync[ABORT_WORD], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      alive = false;
    }
  }
#if FL_PROG_TRACE
  if (threadIdx.x == 0) g_trace = nullptr;
#endif
  unsigned gbar = 0, lbar = 0;   // device-wide / XCD barriers passed so far
  bool dirty = false;            // plain-stored outputs since the last L2 write-back
  int pre = 0;                   // prefetch DMAs per wave in flight for the next tile to run
  bool have_first = false;       // `first` holds the first tile of the phase about to start
  Cur first;
  first.j = -1; first.t = 0; first.base = 0; first.cnt = 0;
  for (int p = 0; p < nphases && alive; ++p) {
#if FL_PROG_TRACE
    if (threadIdx.x == 0) {
      g_trace = (unsigned long long*)io.p[15];
      g_trace_phases = nphases;
      g_trace_p = p;
    }
#endif
    FL_STAMP(0);
#if FL_PROG_TRACE
    if (threadIdx.x == 0 && g_trace) g_trace[((size_t)blockIdx.x * g_trace_phases + g_trace_p) * 12 + 11] = clock64();
#endif
    const int pflags = lds_word(d.lph + 4 * p, 3);
    const bool local = allow_local && (pflags & FL_PPHASE_LOCAL) != 0;
    const bool wt = !local;
    const bool has_next = p + 1 < nphases;
    const int nflags = has_next ? lds_word(d.lph + 4 * (p + 1), 3) : 0;
    const bool nlocal = has_next && allow_local && (nflags & FL_PPHASE_LOCAL) != 0;
    Cur cur = have_first ? first : first_tile(d, p, local);
    have_first = false;
    while (cur.j >= 0) {
      Cur nxt = next_tile(d, p, local, cur);
      // what this block runs after `cur`: the phase's next tile, or the first tile of the next phase
      Cur pf = nxt;
      bool pf_local = local;
      if (nxt.j < 0 && has_next) {
        first = first_tile(d, p + 1, nlocal);
        have_first = true;
        pf = first;
        pf_local = nlocal;
      }
      const int* pf_job = nullptr;
      int pf_ct = 0;
      if (prefetchable(d, pf)) {
        int prt;
        tile_coords(d, pf_local, pf, prt, pf_ct);
        pf_job = d.ljb + pf.j * JW;
      }
      fl_pjob jb;
      load_job(d.ljb + cur.j * JW, jb);
      int rt, k;
      tile_coords(d, local, cur, rt, k);
      pre = run_tile(jb, rt, k, io, smem, wt, pre, d.ljb + cur.j * JW, pf_job, pf_ct);
      cur = nxt;
    }
    dirty = dirty || local;
    FL_STAMP(2);
#if FL_PROG_TRACE
    if (threadIdx.x == 0 && g_trace) {
      unsigned long long* e = &g_trace[((size_t)blockIdx.x * g_trace_phases + g_trace_p) * 12 + 11];
      *e = clock64() - *e;  // core-clock cycles spent on this phase's tiles
    }
#endif
    if (has_next) {
      const bool acq = (nflags & FL_PPHASE_ACQUIRE) != 0;
      if (local && nlocal) {
        ++lbar;
        alive = xcd_barrier(sync, d.xcc, d.slot, d.nx, (epoch << 12) + lbar, acq, pre, &s_ok);
      } else {
        ++gbar;
        alive = grid_barrier(sync, gbar * G, acq, dirty, pre, &s_ok);
        dirty = false;
      }
    }
    FL_STAMP(6);
  }
  // self-cleaning: the last block to leave re-arms the counters for the next run (not after an abort)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && alive) {
    const unsigned old = __hip_atomic_fetch_add(&sync[DEP_WORD], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == G - 1) {
#pragma unroll
      for (int g = 0; g < NCTR; ++g) {
        __hip_atomic_store(&sync[g * CTR_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[REG_OFF + 32 * g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // the XCD flag lines are never reset: their values carry the launch epoch (every block read it before it got here)
      __hip_atomic_store(&sync[EPOCH_WORD], (epoch + 1u) & 0xFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[DEP_WORD], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      // every block has finished every phase: bump the per-run counters (step counter, Philox offset)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (io.post.ptr64[k]) *(__attribute__((address_space(1))) long long*)(io.post.ptr64[k]) += io.post.delta64[k];
      if (io.post.ptr32) *(__attribute__((address_space(1))) int*)(io.post.ptr32) += io.post.delta32;
    }
  }
}

// tiles a job contributes: nrt row tiles x per tiles each; the kinds that are not laid out by row tiles count as ONE row tile
bool flat_kind(int kind) { return kind == FL_PJOB_LNCOL || (kind >= FL_PJOB_FIN && kind != FL_PJOB_EWADD); }
bool t64(const fl_pjob& jb) { return jb.kind == FL_PJOB_GEMM && (jb.flags & FL_PG_T64) != 0; }
int row_tiles(const fl_pjob& jb) { return flat_kind(jb.kind) ? 1 : t64(jb) ? jb.M / 64 : (jb.M + TS - 1) / TS; }
int tiles_per_rowtile(const fl_pjob& jb) {
  switch (jb.kind) {
    case FL_PJOB_GEMM: return (jb.flags & FL_PG_T64) ? jb.N / 64 : jb.N / TS;
    case FL_PJOB_ROWNORM:
    case FL_PJOB_LNBWD:
    case FL_PJOB_LOSS: return TS / 4;
    case FL_PJOB_SINUSOID: return (TS * (jb.N >> 1) + 255) / 256;
    case FL_PJOB_LNCOL: return jb.N / 32;
    case FL_PJOB_FIN: return 1;
    case FL_PJOB_EMBBWD: return jb.K[0];
    case FL_PJOB_SQNORM:
    case FL_PJOB_ADAM: return FLAT_TILES;
    default: return (TS * (jb.N >> 2) + 255) / 256;  // GATHER, PSAMPLE, EWADD: float4 per thread
  }
}

int device_cus() {
  static int cus = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  });
  return cus;
}

}  // namespace

extern "C" {

int fl_program_job_bytes(void) { return (int)sizeof(fl_pjob); }
int fl_program_phase_bytes(void) { return (int)sizeof(fl_pphase); }

int fl_program_job_tiles(const void* job) {
  FL_REQUIRE(job, "fl_program_job_tiles: null job");
  const fl_pjob& j = *reinterpret_cast<const fl_pjob*>(job);
  FL_REQUIRE(j.M >= 1 && j.N >= 1 && j.C, "program job: M=%d N=%d C=%llu", j.M, j.N, (unsigned long long)j.C);
  switch (j.kind) {
    case FL_PJOB_GEMM: {
      const bool ta = (j.flags & FL_PG_TA) != 0, tw = (j.flags & FL_PG_TW) != 0;
      FL_REQUIRE(j.nseg >= 1 && j.nseg <= FL_PROG_MAX_SEG && j.N % TS == 0 && j.ldc % 4 == 0, "program GEMM: nseg=%d N=%d ldc=%d",
                 j.nseg, j.N, j.ldc);
      FL_REQUIRE(!ta || (tw && j.M % TS == 0), "program GEMM: FL_PG_TA needs FL_PG_TW and M %% 32 == 0 (M=%d)", j.M);
      for (int s = 0; s < j.nseg; ++s) {
        FL_REQUIRE(j.A[s] && j.W[s] && j.K[s] >= 1 && j.lda[s] % 4 == 0 && j.ldw[s] % 4 == 0,
                   "program GEMM segment %d: K=%d lda=%d ldw=%d", s, j.K[s], j.lda[s], j.ldw[s]);
        FL_REQUIRE(ta || (j.K[s] >= 4 && j.K[s] % 4 == 0), "program GEMM segment %d: K=%d", s, j.K[s]);
        FL_REQUIRE(tw ? j.ldw[s] >= j.N : j.ldw[s] >= j.K[s], "program GEMM segment %d: ldw=%d", s, j.ldw[s]);
        FL_REQUIRE(!ta || j.lda[s] >= j.M, "program GEMM segment %d: lda=%d", s, j.lda[s]);
        // operands are addressed with 32-bit element offsets and 24-bit multiplies
        const long long arows = ta ? j.K[s] : j.M, wrows = tw ? j.K[s] : j.N;
        FL_REQUIRE(arows < (1 << 24) && wrows < (1 << 24) && j.lda[s] >= 0 && j.lda[s] < (1 << 24) && j.ldw[s] < (1 << 24) &&
                       arows * j.lda[s] < (1ll << 30) && wrows * j.ldw[s] < (1ll << 30),
                   "program GEMM segment %d: operand too large for 32-bit offsets", s);
      }
      FL_REQUIRE(!j.R || j.ldr % 4 == 0, "program GEMM: ldr=%d", j.ldr);
      FL_REQUIRE(!j.mask || (j.mask_group >= 4 && j.mask_group % 4 == 0 && j.N % j.mask_group == 0), "program GEMM: mask_group=%d",
                 j.mask_group);
      FL_REQUIRE(!(j.flags & FL_PG_COLSUM) || (ta && j.aux[0]), "program GEMM: FL_PG_COLSUM needs FL_PG_TA and aux[0]");
      FL_REQUIRE(!(j.flags & FL_PG_ACTBWD) || (j.aux[1] && j.ld2 % 4 == 0), "program GEMM: FL_PG_ACTBWD needs aux[1] (ld2=%d)", j.ld2);
      if (j.flags & FL_PG_T64) {
        FL_REQUIRE(ta && tw && j.nseg == 1 && j.M % 64 == 0 && j.N % 64 == 0 && !j.bias[0] && !j.R && !j.mask && !j.aux[3] &&
                       !(j.flags & FL_PG_ACTBWD) && j.act == FL_ACT_NONE,
                   "program GEMM: FL_PG_T64 is a plain weight-gradient tile (TA|TW, one segment, M, N %% 64 == 0, no epilogue)");
        return (j.M / 64) * (j.N / 64);
      }
      return ((j.M + TS - 1) / TS) * (j.N / TS);
    }
    case FL_PJOB_ROWNORM:
      FL_REQUIRE(j.A[0] && j.W[0] && j.bias[0] && j.N % 256 == 0 && j.N <= 1024 && j.lda[0] % 4 == 0 && j.ldc % 4 == 0,
                 "program ROWNORM: N=%d lda=%d ldc=%d", j.N, j.lda[0], j.ldc);
      FL_REQUIRE(!j.W[1] || (j.bias[1] && j.aux[0]), "program ROWNORM: second norm needs bias[1] and aux[0]");
      FL_REQUIRE(!j.gate || (j.aux[1] && j.lda[1] % 4 == 0), "program ROWNORM: gate needs aux[1]");
      return ((j.M + TS - 1) / TS) * (TS / 4);
    case FL_PJOB_SINUSOID:
      FL_REQUIRE(j.A[0] && j.W[0] && j.N % 2 == 0, "program SINUSOID: N=%d", j.N);
      return ((j.M + TS - 1) / TS) * ((TS * (j.N / 2) + 255) / 256);
    case FL_PJOB_GATHER:
      FL_REQUIRE(j.A[0] && j.W[0] && j.N % 4 == 0 && j.lda[0] % 4 == 0 && j.ldc % 4 == 0, "program GATHER: N=%d", j.N);
      return ((j.M + TS - 1) / TS) * ((TS * (j.N / 4) + 255) / 256);
    case FL_PJOB_PSAMPLE:
      FL_REQUIRE(j.A[0] && j.A[1] && j.W[0] && j.W[1] && j.W[2] && j.aux[2] && (j.A[2] || j.aux[0]) &&
                     ((long long)j.M * j.N) % 4 == 0,
                 "program PSAMPLE: needs x, eps, alpha/alpha_bar/beta tables, t and noise or a Philox offset (M*N %% 4 == 0)");
      FL_REQUIRE(j.N % 4 == 0, "program PSAMPLE: N %% 4");
      return ((j.M + TS - 1) / TS) * ((TS * (j.N / 4) + 255) / 256);
    case FL_PJOB_LNBWD:
      FL_REQUIRE(j.A[0] && j.A[1] && j.W[0] && j.bias[0] && j.N % 256 == 0 && j.N <= 1024 && j.lda[0] % 4 == 0 &&
                     j.lda[1] % 4 == 0 && j.ldc % 4 == 0 && (!j.R || j.ldr % 4 == 0),
                 "program LNBWD: N=%d lda=%d,%d ldc=%d", j.N, j.lda[0], j.lda[1], j.ldc);
      FL_REQUIRE(!j.A[2] || (j.W[1] && j.bias[1] && j.aux[2] && j.lda[2] % 4 == 0),
                 "program LNBWD: the two-stage form needs W[1], bias[1] and aux[2]");
      return ((j.M + TS - 1) / TS) * (TS / 4);
    case FL_PJOB_LNCOL:
      FL_REQUIRE(j.A[0] && j.A[1] && j.aux[0] && j.aux[1] && j.N % 32 == 0, "program LNCOL: N=%d", j.N);
      return j.N / 32;
    case FL_PJOB_LOSS:
      FL_REQUIRE(j.A[0] && j.A[1] && j.aux[0] && j.N % 256 == 0 && j.N <= 1024 && j.lda[0] % 4 == 0 && j.lda[1] % 4 == 0 &&
                     j.ldc % 4 == 0 && (!j.A[2] || (j.aux[1] && j.lda[2] % 4 == 0)),
                 "program LOSS: N=%d", j.N);
      return ((j.M + TS - 1) / TS) * (TS / 4);
    case FL_PJOB_FIN:
      FL_REQUIRE(j.A[0] && (!j.A[1] || (j.aux[0] && j.gate)), "program FIN: the gate gradient needs aux[0] and gate");
      return 1;
    case FL_PJOB_EWADD:
      FL_REQUIRE(j.A[0] && j.A[1] && j.N % 4 == 0 && j.lda[0] % 4 == 0 && j.lda[1] % 4 == 0 && j.lda[2] % 4 == 0 && j.ldc % 4 == 0,
                 "program EWADD: N=%d", j.N);
      return ((j.M + TS - 1) / TS) * ((TS * (j.N / 4) + 255) / 256);
    case FL_PJOB_EMBBWD:
      FL_REQUIRE(j.A[0] && j.W[0] && j.K[0] >= 1 && j.N % 4 == 0 && j.N <= 4096 && j.lda[0] % 4 == 0 && j.ldc % 4 == 0,
                 "program EMBBWD: rows=%d N=%d", j.K[0], j.N);
      return j.K[0];
    case FL_PJOB_SQNORM:
      FL_REQUIRE(j.A[0] && j.K[0] >= 4 && j.K[0] % 4 == 0 && j.K[0] < (1 << 30), "program SQNORM: n=%d", j.K[0]);
      return FLAT_TILES;
    case FL_PJOB_ADAM:
      FL_REQUIRE(j.A[0] && j.W[0] && j.W[1] && j.W[2] && j.aux[0] && j.aux[1] && j.K[0] >= 4 && j.K[0] % 4 == 0 &&
                     j.K[0] < (1 << 30),
                 "program ADAM: needs g, partials, lr, step, m, v (n=%d)", j.K[0]);
      return FLAT_TILES;
    default:
      fl::set_error("program job: unknown kind %d", j.kind);
      return FL_ERR_BAD_ARG;
  }
}

int fl_program_job_layout(void* job) {
  const int n = fl_program_job_tiles(job);
  if (n < 0) return n;
  fl_pjob& j = *reinterpret_cast<fl_pjob*>(job);
  j.nrt = row_tiles(j);
  j.per = tiles_per_rowtile(j);
  FL_REQUIRE(j.nrt * j.per == n, "program job: tile layout %d x %d != %d", j.nrt, j.per, n);
  j.flags &= ~FL_PG_DMA;
  if (j.kind == FL_PJOB_GEMM && !(j.flags & FL_PG_TA)) {
    bool ok = true;
    for (int s = 0; s < j.nseg; ++s) {
      // (not TW: K_s is the contraction length of both operands; TW: W has K_s rows of N floats)
      ok = ok && j.K[s] % DKC == 0 && (j.A[s] <= FL_PROG_MAX_IO || j.A[s] % 16 == 0) && (j.W[s] <= FL_PROG_MAX_IO || j.W[s] % 16 == 0);
    }
    if (ok) j.flags |= FL_PG_DMA;
  }
  return n;
}

int fl_program_sync_bytes(void) { return SYNC_UINTS * 4; }

int fl_program_run(const void* phases, const void* jobs, int n_phases, int n_jobs, const void* const* io, int n_io,
                   void* sync, int max_blocks, int flags, const void* post, fl_stream_t stream) {
  FL_REQUIRE(phases && jobs && sync && n_phases >= 1 && n_jobs >= 1 && n_io >= 0 && n_io <= FL_PROG_MAX_IO && (n_io == 0 || io),
             "fl_program_run: bad args (n_phases=%d n_jobs=%d n_io=%d)", n_phases, n_jobs, n_io);
  FL_REQUIRE((size_t)n_phases * sizeof(fl_pphase) + (size_t)n_jobs * sizeof(fl_pjob) <= (size_t)PROG_LDS_BYTES,
             "fl_program_run: program too large for its LDS copy (%d phases, %d jobs; limit %d bytes)", n_phases, n_jobs,
             PROG_LDS_BYTES);
  ProgIO pio;
  for (int i = 0; i < FL_PROG_MAX_IO; ++i) pio.p[i] = i < n_io ? io[i] : nullptr;
  memset(&pio.post, 0, sizeof(pio.post));
  if (post) pio.post = *reinterpret_cast<const fl_program_post*>(post);
  pio.abort = reinterpret_cast<unsigned*>(sync) + ABORT_WORD;
  for (int i = 0; i < n_io; ++i)
    FL_REQUIRE((reinterpret_cast<uintptr_t>(io[i]) & 15) == 0, "fl_program_run: io[%d] is not 16-byte aligned", i);
  const int cus = device_cus();
  if (cus <= 0) {
    fl::set_error("fl_program_run: no device");
    return FL_ERR_UNSUPPORTED;
  }
  FL_REQUIRE(n_phases < 4096, "fl_program_run: %d phases (the XCD barrier counts them in 12 bits)", n_phases);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(program_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  // the barrier needs every block resident: one block per CU (157 KB of LDS each: a second block cannot join it)
  int G = cus;
  if (max_blocks > 0 && max_blocks < G) G = max_blocks;
  // local phases need the same number of blocks on every XCD (one block per CU: G / 8 each when G is the CU count)
  const int allow_local = (flags & FL_PROG_XCD_LOCAL) != 0 && G == cus && G % NCTR == 0 && G / NCTR <= 32;
  hipLaunchKernelGGL(program_kernel, dim3(G), dim3(256), LDS_BYTES, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const fl_pphase*>(phases), reinterpret_cast<const fl_pjob*>(jobs), n_phases, n_jobs, pio,
                     reinterpret_cast<unsigned*>(sync), allow_local);
  return fl::check_launch("program_kernel");
}

}  // extern "C"
