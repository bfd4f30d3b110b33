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
// LDS-DMA GEMM tile (FL_PG_DMA): rings of 128-wide k chunks (3 A slots, 5 W slots of 16 KB), unpadded images whose
// 16-byte column blocks are XOR-swizzled by the row (the swizzle is applied on the GLOBAL side of the DMA: lane L lands at
// LDS byte 16 L whatever it fetched)
constexpr int DKC = 128;                 // k values per chunk (32 per wave)
constexpr int DIMG = TS * DKC;           // floats per operand image (16 KB)
constexpr int NA = 3, NW = 5;            // ring slots: A chunk c in slot c % 3, W chunk c in slot 3 + c % 5
static_assert((NA + NW) * DIMG <= 2 * STAGE, "the DMA rings live in the two register-staged stages' LDS");
constexpr int NCTR = 8;            // arrival counters
constexpr int CTR_STRIDE = 1024;   // ... this many uints apart (4 KB)
constexpr int SYNC_UINTS = (NCTR + 1) * CTR_STRIDE;
constexpr unsigned SPIN_LIMIT = 1u << 21;  // barrier polls before a run is declared dead (~ seconds)

// -DFL_PROG_TRACE: timing-only build; io slot 15 is a uint64 buffer [block][phase][16] of 100 MHz wall-clock stamps
#ifndef FL_PROG_TRACE
#define FL_PROG_TRACE 0
#endif
// -DFL_PG_ABLATE=n: timing-only builds of the LDS-DMA GEMM tile (WRONG results): 1 no MFMAs, 2 no W loads behind the prologue,
// 3 no A loads behind the prologue, 4 no fragment reads, 5 = 2 + 3
#ifndef FL_PG_ABLATE
#define FL_PG_ABLATE 0
#endif
#if FL_PROG_TRACE
__shared__ unsigned long long* g_trace;  // only thread 0 of the block touches these
__shared__ int g_trace_phases, g_trace_p;
#define FL_STAMP(k)                                                                                        \
  do {                                                                                                     \
    if (threadIdx.x == 0 && g_trace) g_trace[((size_t)blockIdx.x * g_trace_phases + g_trace_p) * 16 + (k)] = wall_clock64(); \
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
// The block keeps a decoded tile's descriptor as ONE VGPR (`jobwords_t`: lane i = word i, fetch_job) and every tile kind
// unpacks just the fields it uses, where it runs (load_job): two whole descriptors in scalar registers (the running tile's and
// the next one's) cost ~600 SGPR spills and a microsecond per tile.
__device__ __forceinline__ jobwords_t fetch_job(const int* src /* LDS */) {
  static_assert(sizeof(fl_pjob) / 4 <= 64, "descriptor must fit one word per lane");
  const int ln = threadIdx.x & 63;
  return ((lint_p)src)[ln < (int)(sizeof(fl_pjob) / 4) ? ln : 0];
}
__device__ __forceinline__ void load_job(const jobwords_t p, fl_pjob& d) {
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
  const int tidx = threadIdx.x & 255;  // waves 4-7 of the 512-thread block shadow waves 0-3 (same loads, same LDS words, same stores)
  const int w = __builtin_amdgcn_readfirstlane(tidx >> 6);
  const int lane = tidx & 63, lr = lane & 31, lh = lane >> 5;
  const int kq4 = lane * 4;  // this thread's k offset inside a chunk row (a wave covers one whole 256-wide row slice)
  const int tr = tidx >> 3, tc4 = (tidx & 7) * 4;  // transposed operands: chunk rows tr + 32 i, 4 columns

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
        const float* Ia_ = smem + (st) * STAGE + (tidx >> 5) * 32 * TS + (tidx & 31); \
        _Pragma("unroll 8") for (int j_ = 0; j_ < 32; ++j_) cs += *(const lf1_t*)(Ia_ + j_ * TS);  \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // epilogue operands are requested first: they ride along with the chunk loads instead of costing a memory round
  // trip (about 1 us here) after the contraction
  const int er = tidx >> 3, ec = (tidx & 7) * 4;
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
  // ONE register stage, two LDS stages (this tile only serves operands the LDS-DMA tile does not take: TA weight-gradient
  // tiles that are not 64 x 64, K not a multiple of 128): chunk c + 1 is in flight while chunk c is contracted. Parking
  // chunk c into stage c & 1 is safe once the wave has passed iteration c - 1's barrier (every wave has then left chunk c - 2).
  Stage R0;
  bool ok0 = false;
  int kv0 = 0;
  FL_STAMP(7);
  FL_ISSUE(R0, 0, ok0, kv0);
  FL_STAMP(8);
  for (int c = 0; c < nC; ++c) {
    FL_PARK(R0, ok0, kv0, c & 1);
    if (c + 1 < nC) FL_ISSUE(R0, c + 1, ok0, kv0);
    __syncthreads();
    FL_CONTRACT(c & 1);
    FL_COLSUM(c & 1);
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
    if (colsum) *(lf1_t*)(smem + 4 * TS * SLD + tidx) = cs;  // [8 row groups][32 columns], behind the partial tiles
  }
  __syncthreads();
  const float* s0 = smem + er * SLD + ec;
  f32x4v vv = *(const lf4_t*)(s0);
#pragma unroll
  for (int k = 1; k < 4; ++k) vv += *(const lf4_t*)(s0 + k * (TS * SLD));
  float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
  const float gmul = jb.gate ? 1.f + fl::sigmoid_acc(e_gate) : 1.f;
  if constexpr (TA) {
    if (colsum && tidx < TS) {  // FL_PG_TA requires M % 32 == 0: every row of the tile exists
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += *(const lf1_t*)(smem + 4 * TS * SLD + g * TS + tidx);
      st1g(wt, wp(jb.aux[0], io) + rt * TS + tidx, jb.alpha * gmul * t);
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
// the workgroup barrier WITHOUT the memory fence of __syncthreads() (hipcc puts s_waitcnt vmcnt(0) lgkmcnt(0) in front of that
// one: it would drain every DMA in flight). LDS reads of the waiting wave must have been consumed before (they feed MFMAs).
__device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }
// K == 256 (the batch contraction of the headline configuration): the two 128-k halves of the images are two stages, X and Y,
// and the block's NEXT weight-gradient tile (the dealing loop knows it) is streamed in behind the running one - X' while Y is
// being multiplied, Y' under the stores: [wait X | mul X | issue X' | wait Y | mul Y | store | issue Y'] - so a tile costs its
// 128 MFMAs per wave instead of [load 128 KB out of the Infinity Cache, then multiply] (r03: 9.7 us per tile, 3.7 of them
// MFMA). Every wave issues 8 DMA instructions per half, so "all but my youngest 8" is the wait for the older half whatever
// stores lie in between. `pre` == WG64_INFLIGHT: the previous tile left this tile's halves in flight; returns the same for the next.
constexpr int WG64_INFLIGHT = -1;
__device__ __forceinline__ int wg64_tile(const fl_pjob& jb, int rt, int ct, const ProgIO& io, float* smem, const bool wt, const int pre,
                                         const jobwords_t njw, const bool has_next, const int nrt, const int nct) {
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), w = w8 & 3;  // 8 waves load, waves 0-3 compute
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const int wr = w >> 1, wc = w & 1;
  const bool compute = w8 < 4;
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
  if (K == 256) {
    bool nx = false;
    const float* nA0 = A0;
    const float* nW0 = W0;
    int nlda = lda, nldw = ldw;
    if (has_next) {
      const int nfl = ldsi(njw, FL_JW(flags));
      nx = ldsi(njw, FL_JW(kind)) == FL_PJOB_GEMM && (nfl & (FL_PG_T64 | FL_PG_TA | FL_PG_TW)) == (FL_PG_T64 | FL_PG_TA | FL_PG_TW) &&
           ldsi(njw, FL_JW(K)) == 256;
      if (nx) {
        nA0 = rp(ldsp(njw, FL_JW(A)), io) + nrt * 64 + (lane & 15) * 4;
        nW0 = rp(ldsp(njw, FL_JW(W)), io) + nct * 64 + (lane & 15) * 4;
        nlda = ldsi(njw, FL_JW(lda));
        nldw = ldsi(njw, FL_JW(ldw));
      }
    }
    // wave w8 moves rows [16 w8, 16 w8 + 16) of a half of both images: 4 x (4 rows x 256 B) each
    auto issue = [&](const float* Ab, const float* Wb, int lda_, int ldw_, int half) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row4 = half * 128 + (w8 * 4 + q) * 4, kr = row4 + (lane >> 4);
        __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)__umul24((unsigned)kr, (unsigned)lda_)), (lptr_t)(imgA + row4 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)__umul24((unsigned)kr, (unsigned)ldw_)), (lptr_t)(imgW + row4 * 64), 16, 0, 0);
      }
    };
    // a half = 64 MFMA steps (k pairs); the fragments of the next 16 steps are read before the MFMAs of the running 16 are issued
    // (read-then-multiply in one piece cost 6.7 us per tile for 3.7 us of MFMAs)
    auto mul = [&](int half) {
      if (!compute) return;
      const float* pa = imgA + (half * 128 + lh) * 64 + 32 * wr + lr;
      const float* pw = imgW + (half * 128 + lh) * 64 + 32 * wc + lr;
      float fa[2][16], fw[2][16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        fa[0][j] = *(const lf1_t*)(pa + j * 128);
        fw[0][j] = *(const lf1_t*)(pw + j * 128);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b + 1 < 4) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            fa[(b + 1) & 1][j] = *(const lf1_t*)(pa + (16 * (b + 1) + j) * 128);
            fw[(b + 1) & 1][j] = *(const lf1_t*)(pw + (16 * (b + 1) + j) * 128);
          }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[b & 1][j], fw[b & 1][j], acc, 0, 0, 0);
          cs += fa[b & 1][j];
        }
      }
    };
    FL_STAMP(4);
    if (pre != WG64_INFLIGHT) {
      issue(A0, W0, lda, ldw, 0);
      issue(A0, W0, lda, ldw, 1);
    }
    FL_STAMP(7);
    FL_STAMP(8);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // X (and whatever the previous tile stored) is done; Y may still fly
    raw_barrier();
    FL_STAMP(9);
    mul(0);
    raw_barrier();  // every wave has read X
    if (nx) {
      issue(nA0, nW0, nlda, nldw, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    raw_barrier();  // Y has landed
    mul(1);
    raw_barrier();  // every wave has read Y
    FL_STAMP(1);
    float* C = wp(jb.C, io) + (long long)(rt * 64 + 32 * wr + 4 * lh) * jb.ldc + ct * 64 + 32 * wc + lr;
    if (compute) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st1g(wt, C + (long long)((r & 3) + 8 * (r >> 2)) * jb.ldc, acc[r]);
    }
    if (colsum && compute) {
      cs += __shfl_xor(cs, 32, 64);
      const float gmul = jb.gate ? 1.f + fl::sigmoid_acc(ld1g(rp(jb.gate, io))) : 1.f;
      if (lh == 0) st1g(wt, wp(jb.aux[0], io) + rt * 64 + 32 * wr + lr, jb.alpha * gmul * cs);
    }
    if (nx) issue(nA0, nW0, nlda, nldw, 1);
    FL_STAMP(10);
    return nx ? WG64_INFLIGHT : (pre == WG64_INFLIGHT ? 0 : pre);
  }
  const float e_gate = jb.gate ? ld1g(rp(jb.gate, io)) : 0.f;
  for (int kk = 0; kk < K; kk += 256) {
    const int kv = K - kk < 256 ? K - kk : 256;
    if (kk) __syncthreads();  // every wave has read the previous chunk
    // wave w8 moves rows [32 w8, 32 w8 + 32) of both images: 8 x (4 rows x 256 B) each; rows past the end re-read row K-1
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row4 = (w8 * 8 + q) * 4;
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
    const int steps = compute ? (kv + 1) >> 1 : 0;
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
  if (compute) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st1g(wt, C + (long long)((r & 3) + 8 * (r >> 2)) * jb.ldc, acc[r]);
  }
  if (colsum && compute) {
    cs += __shfl_xor(cs, 32, 64);
    const float gmul = jb.gate ? 1.f + fl::sigmoid_acc(e_gate) : 1.f;
    if (lh == 0) st1g(wt, wp(jb.aux[0], io) + rt * 64 + 32 * wr + lr, jb.alpha * gmul * cs);
  }
  __syncthreads();  // the images are the next tile's stage
  return pre == WG64_INFLIGHT ? 0 : pre;
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

// Operand streams of a FL_PG_DMA job: up to three concatenated-K segments per operand, walked in 128-wide chunks.
// (Local scalars, copied out of the descriptor BEFORE any select: a struct or array whose members are picked by a
// computed segment index is kept in scratch memory by hipcc.)
#define FL_DMA_SRC(P, J, IO)                                                                                       \
  const int P##nseg = (J).nseg;                                                                                    \
  const float* P##A0 = rp((J).A[0], IO);                                                                           \
  const float* P##A1 = P##nseg > 1 ? rp((J).A[1], IO) : P##A0;                                                     \
  const float* P##A2 = P##nseg > 2 ? rp((J).A[2], IO) : P##A0;                                                     \
  const float* P##W0 = rp((J).W[0], IO);                                                                           \
  const float* P##W1 = P##nseg > 1 ? rp((J).W[1], IO) : P##W0;                                                     \
  const float* P##W2 = P##nseg > 2 ? rp((J).W[2], IO) : P##W0;                                                     \
  const int P##lda0 = (J).lda[0], P##lda1 = (J).lda[1], P##lda2 = (J).lda[2];                                      \
  const int P##ldw0 = (J).ldw[0], P##ldw1 = (J).ldw[1], P##ldw2 = (J).ldw[2];                                      \
  const int P##c1 = (J).K[0] / DKC;                                                                                \
  const int P##c2 = P##nseg > 1 ? P##c1 + (J).K[1] / DKC : P##c1;                                                  \
  const int P##nC = P##nseg > 2 ? P##c2 + (J).K[2] / DKC : P##c2
// chunk c of stream P: segment base pointers, leading dimensions, k offset inside the segment
#define FL_DMA_CHUNK(P, c)                                                                                         \
  const bool s1_ = (c) >= P##c1, s2_ = (c) >= P##c2;                                                               \
  const float* Ab_ = s2_ ? P##A2 : (s1_ ? P##A1 : P##A0);                                                          \
  const float* Wb_ = s2_ ? P##W2 : (s1_ ? P##W1 : P##W0);                                                          \
  const int lda_ = s2_ ? P##lda2 : (s1_ ? P##lda1 : P##lda0), ldw_ = s2_ ? P##ldw2 : (s1_ ? P##ldw1 : P##ldw0);    \
  const int kk_ = ((c) - (s2_ ? P##c2 : (s1_ ? P##c1 : 0))) * DKC;                                                 \
  const int ncs_ = s2_ ? P##nC - P##c2 : (s1_ ? P##c2 - P##c1 : P##c1); /* chunks of the chunk's segment */           \
  (void)Ab_; (void)lda_; (void)ncs_

// W chunk (base Wb, leading dimension ldw, k offset kk) of column tile ct -> W image `img`; lw = loader wave 0..3.
//   not TW: W[n][k] rows n = 32 ct + 8 lw + 2 i + (lane >> 5); the lane fetches the 16-byte block j = (lane & 31) ^ (row & 15)
//           of the chunk (row = 8 lw + 2 i + (lane >> 5) inside the tile) and lands at block position lane & 31
//   TW:     W[k][n] rows k = 32 lw + 8 i + (lane >> 3) of the chunk, 128 B each: the image is [128 k][32 n], as it lies
__device__ __forceinline__ void dma_w(const float* Wb, int ldw, int kk, int ct, bool tw, bool packed, int ncs, float* img, int lw,
                                      int lane) {
  if (packed) {  // FL_PG_WPACK: the chunk's image lies in memory as it will in LDS: 16 contiguous 1 KB pieces
    const float* src = Wb + ((size_t)ct * ncs + (kk / DKC)) * DIMG + lane * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (8 * lw + 2 * i) * DKC), (lptr_t)(img + (8 * lw + 2 * i) * DKC), 16, 0, 0);
  } else if (tw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = __umul24((unsigned)(kk + 32 * lw + 8 * i + (lane >> 3)), (unsigned)ldw) + (unsigned)(ct * TS + (lane & 7) * 4);
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)off), (lptr_t)(img + (32 * lw + 8 * i) * TS), 16, 0, 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * lw + 2 * i + (lane >> 5);
      const unsigned off = __umul24((unsigned)(ct * TS + r), (unsigned)ldw) + (unsigned)(kk + 4 * ((lane & 31) ^ (r & 15)));
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)off), (lptr_t)(img + (8 * lw + 2 * i) * DKC), 16, 0, 0);
    }
  }
}

// One 32 x 32 output tile of a FL_PG_DMA job, run by the 512-thread block as 4 CONSUMER waves (0-3: fragment reads + MFMAs,
// each its own 32-wide k slice of every chunk, then the partial tiles meet in LDS and the epilogue) and 4 LOADER waves (4-7:
// every LDS-DMA instruction of the tile, 8 per wave and chunk). A DMA instruction costs its wave ~60-180 issue cycles that
// nothing hides inside that wave; on a wave of its own it runs under the MFMAs of the consumer wave on the same SIMD.
// Two rings (tools/dma_rate_probe.hip: a CU pulls 130-150 GB/s out of its XCD's L2 but only ~33 GB/s when every CU streams
// from the Infinity Cache, which is where the weights of an XCD-local phase come from - every XCD reads all of them - while
// the activations sit in L2): the W ring is 5 chunks deep and is filled for the block's NEXT tile as soon as this tile's
// loop is over (the stream keeps running through the epilogue, the barrier and whatever row phase lies between two GEMM
// phases), the A ring is 3 deep and filled behind the barrier that publishes the activations.
// ONE raw s_barrier per chunk: passing barrier b means "chunk b has landed" (each loader waited for its own pieces with a
// counted vmcnt) and "every consumer holds its fragments of chunk b - 1 in registers": its slots are refilled with A chunk
// b + 2 and W chunk b + 4. Issue order of a loader wave (4 instructions per operand and chunk):
//   [W_0..W_3 unless prefetched], A_0, A_1 | barrier 0 | A_2, W_4 | barrier 1 | A_3, W_5 | barrier 2 | ...
// `pre` (loaders): DMA instructions of THIS tile's first W chunks already in flight (issued at the end of the block's
// previous GEMM tile). prefetch: the block's next tile (descriptor image njw) is a prefetchable GEMM (column tile pf_ct).
// Returns the DMA instructions the wave left in flight.
template <bool TW>
__device__ __forceinline__ int gemm_dma_tile(const jobwords_t jw, int rt, int ct, const ProgIO& io, float* smem, const bool wt,
                                             const int pre, const jobwords_t& njw, const bool prefetch, const int pf_ct) {
  fl_pjob jb;
  load_job(jw, jb);
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), w = w8 & 3;
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  FL_DMA_SRC(m_, jb, io);
  FL_STAMP(4);  // the descriptor is unpacked
  const int nC = m_nC;
  float* ringW = smem + NA * DIMG;
  float* part = smem;  // the partial tiles meet in the A ring (free once the loop is over; the W ring takes the next tile's chunks)
  if (w8 >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    int arow[4], acol[4];  // A rows of this lane's four DMA instructions (rows past M repeat the last one: never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * w + 2 * i + (lane >> 5);
      const int gr_ = rt * TS + r;
      arow[i] = gr_ < jb.M ? gr_ : jb.M - 1;
      acol[i] = 4 * ((lane & 31) ^ (r & 15));
    }
#define FL_LOAD_A(c, slot)                                                                               \
  do {                                                                                                   \
    FL_DMA_CHUNK(m_, c);                                                                                 \
    (void)Wb_; (void)ldw_;                                                                               \
    float* img_ = smem + (slot) * DIMG;                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                   \
      const unsigned off_ = __umul24((unsigned)arow[i_], (unsigned)lda_) + (unsigned)(kk_ + acol[i_]);   \
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab_ + (size_t)off_), (lptr_t)(img_ + (8 * w + 2 * i_) * DKC), 16, 0, 0); \
    }                                                                                                    \
  } while (0)
#define FL_LOAD_W(c, slot)                                                                               \
  do {                                                                                                   \
    FL_DMA_CHUNK(m_, c);                                                                                 \
    dma_w(Wb_, ldw_, kk_, ct, TW, wpack, ncs_, ringW + (slot) * DIMG, w, lane);                           \
  } while (0)
    const bool wpack = (jb.flags & FL_PG_WPACK) != 0;
    const int npre = pre >> 2;  // W chunks already in flight (0..5)
    for (int c = npre; c < 4 && c < nC; ++c) FL_LOAD_W(c, c);
    FL_LOAD_A(0, 0);
    if (nC > 1) FL_LOAD_A(1, 1);
    // W_j is issued inside the loop (behind barrier j - 4) iff it exists and did not come with the prefetch
#define FL_W_IN_LOOP(j) ((j) < nC && (j) >= npre && FL_PG_ABLATE != 2 && FL_PG_ABLATE != 5)
#define FL_A_IN_LOOP(j) ((j) < nC && FL_PG_ABLATE != 3 && FL_PG_ABLATE != 5)
    int sa = 2, sw = 4;  // ring slots of A chunk b + 2 / W chunk b + 4
    for (int b = 0; b < nC; ++b) {
      // DMA instructions issued after A_b (W_b is older than A_b): b = 0: A_1; b = 1: A_2, W_4; else W_(b+2), A_(b+1), W_(b+3)
      int after;
      if (b == 0) after = nC > 1 ? 4 : 0;
      else if (b == 1) after = (FL_A_IN_LOOP(2) ? 4 : 0) + (FL_W_IN_LOOP(4) ? 4 : 0);
      else after = (FL_W_IN_LOOP(b + 2) ? 4 : 0) + (FL_A_IN_LOOP(b + 1) ? 4 : 0) + (FL_W_IN_LOOP(b + 3) ? 4 : 0);
      wait_vm(after);
      raw_barrier();  // barrier b
      if (FL_A_IN_LOOP(b + 2)) FL_LOAD_A(b + 2, sa);
      if (FL_W_IN_LOOP(b + 4)) FL_LOAD_W(b + 4, sw);
      sa = sa == NA - 1 ? 0 : sa + 1;
      sw = sw == NW - 1 ? 0 : sw + 1;
    }
#undef FL_W_IN_LOOP
#undef FL_A_IN_LOOP
#undef FL_LOAD_A
#undef FL_LOAD_W
    raw_barrier();  // every consumer has left the loop: nothing is being read
    raw_barrier();  // the partial tiles are written
    raw_barrier();  // ... and read: the A ring is free again
    // the next GEMM tile's first W chunks into the W ring (free since the loop ended), behind the tile's last barrier so that
    // no consumer ever waits for this wave's issue slots; the stream runs on through the epilogue, the phase barrier and
    // any row phase in between
    int issued = 0;
    if (prefetch) {
      fl_pjob nj;  // (only the W side of the next tile's descriptor is unpacked)
      load_job(njw, nj);
      FL_DMA_SRC(n_, nj, io);
      const bool ntw = (nj.flags & FL_PG_TW) != 0, npack = (nj.flags & FL_PG_WPACK) != 0;
      const int n = n_nC < NW ? n_nC : NW;
      for (int c = 0; c < n; ++c) {
        FL_DMA_CHUNK(n_, c);
        dma_w(Wb_, ldw_, kk_, pf_ct, ntw, npack, ncs_, ringW + c * DIMG, w, lane);
      }
      issued = 4 * n;
    }
    return issued;
  }
  // -------------------------------------------------------------------------------------------------- consumer waves
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
  FL_STAMP(8);
  // fragments of chunk c + 1 are read (behind barrier c + 1) BEFORE the MFMAs of chunk c are issued: the LDS latency hides
  // under 16 MFMAs. Passing barrier c + 1 also tells the loaders that every consumer holds chunk c in registers.
  f32x4v fa[2][4], fw[2][4];
#define FL_READ_FRAGS(c, set)                                                                           \
  do {                                                                                                  \
    const float* Ia_ = smem + ca * DIMG;                                                                \
    const float* Iw_ = ringW + cw * DIMG;                                                               \
    ca = ca == NA - 1 ? 0 : ca + 1;                                                                     \
    cw = cw == NW - 1 ? 0 : cw + 1;                                                                     \
    if (FL_PG_ABLATE == 4 && (c) > 0) break;                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
      const int j_ = 8 * w + 4 * lh + q_; /* 16-byte k block of this lane's fragment: k = 4 j + e */    \
      fa[set][q_] = *(const lf4_t*)(Ia_ + lr * DKC + 4 * (j_ ^ (lr & 15)));                             \
      if constexpr (TW) {                                                                               \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) fw[set][q_][e_] = *(const lf1_t*)(Iw_ + (4 * j_ + e_) * TS + lr); \
      } else {                                                                                          \
        fw[set][q_] = *(const lf4_t*)(Iw_ + lr * DKC + 4 * (j_ ^ (lr & 15)));                           \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
#define FL_MFMAS(set)                                                                                   \
  do {                                                                                                  \
    if (FL_PG_ABLATE == 1) break;                                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                  \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][q_][e_], fw[set][q_][e_], acc, 0, 0, 0);     \
    }                                                                                                   \
  } while (0)
  int ca = 0, cw = 0;  // ring slots of the next chunk to read
  raw_barrier();  // barrier 0: chunk 0 is in LDS
  FL_STAMP(9);
  FL_READ_FRAGS(0, 0);
  for (int c = 0; c < nC; c += 2) {
    if (c + 1 < nC) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      raw_barrier();  // barrier c + 1
      FL_READ_FRAGS(c + 1, 1);
    }
    FL_MFMAS(0);
    if (c + 1 < nC) {
      if (c + 2 < nC) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();  // barrier c + 2
        FL_READ_FRAGS(c + 2, 0);
      }
      FL_MFMAS(1);
    }
  }
#undef FL_READ_FRAGS
#undef FL_MFMAS
  FL_STAMP(1);
  raw_barrier();  // every consumer has left the loop
  // meet in LDS: sm[w][row][col]; C/D layout of 32x32x2: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* mine = part + w * (TS * SLD);
#pragma unroll
  for (int r = 0; r < 16; ++r) *(lf1_t*)(mine + ((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + lr) = acc[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();  // the partial tiles are written
  FL_STAMP(10);
  const float* s0 = part + er * SLD + ec;
  f32x4v vv = *(const lf4_t*)(s0);
#pragma unroll
  for (int k = 1; k < 4; ++k) vv += *(const lf4_t*)(s0 + k * (TS * SLD));
  float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();  // ... and read
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
  return 0;
}

// Sum over the 64 lanes of a wave, the same value in every lane: four DPP steps (quad swaps, half-row and row mirrors: a
// few cycles each) leave each 16-lane row's sum in all of its lanes, v_readlane picks the four row sums up. (The ds_bpermute
// butterfly of __shfl_xor pays an LDS round trip per step: six dependent ~120-cycle hops per sum, four sums per LayerNorm pair.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wsum(float v) {  // needs all 64 lanes active
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

// nvq (1, 2 or 4) consecutive floats per lane
struct f4a { float v[4]; };
__device__ __forceinline__ f4a ldn(const float* p, int nvq) {
  f4a r;
  r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f;
  if (nvq == 4) {
    const float4 t = ld4g(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if (nvq == 2) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v t = *(__attribute__((address_space(1))) const f32x2v*)(p);
    r.v[0] = t[0]; r.v[1] = t[1];
  } else {
    r.v[0] = ld1g(p);
  }
  return r;
}
__device__ __forceinline__ void stn(bool wt, float* p, const f4a& x, int nvq) {
  if (nvq == 4) {
    st4g(wt, p, make_float4(x.v[0], x.v[1], x.v[2], x.v[3]));
  } else {
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (e < nvq) st1g(wt, p + e, x.v[e]);
  }
}

// Row LayerNorm (+mask, act, residual, gated extra) and an optional second LayerNorm of the result: ONE ROW PER TILE, a
// quarter row per wave (waves 0-3; the four quarter sums meet in LDS), so that the 32 rows of an XCD's row tile run on 32
// CUs. One wave per row (four rows per tile, eight tiles per row tile) kept 24 of an XCD's 32 CUs idle while eight of them
// pulled 128 KB each through a path that delivers ~45 GB/s per CU: ~9 us for the 1024-wide rows. Every operand of the row
// is requested BEFORE the first reduction: one memory round trip per tile.
// All 8 waves run the barriers (raw: no memory fence, the loaders' prefetch DMAs stay in flight).
__device__ __forceinline__ float block4_sum(float v, float* slot, int w) {  // waves 0-3; `slot`: 4 floats of LDS, used once per tile
  v = wsum(v);
  if ((threadIdx.x & 63) == 0) *(lf1_t*)(slot + w) = v;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
  return (*(const lf1_t*)(slot) + *(const lf1_t*)(slot + 1)) + (*(const lf1_t*)(slot + 2) + *(const lf1_t*)(slot + 3));
}
__device__ __forceinline__ void rownorm_tile(const fl_pjob& jb, int row_, const ProgIO& io, float* smem, const bool wt) {
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const bool two = jb.W[1] != 0;
  raw_barrier();  // the reduction slots of the block's previous tile have been read
  if (w8 >= 4) {  // the loader waves only keep the barriers company
    raw_barrier();
    raw_barrier();
    if (two) {
      raw_barrier();
      raw_barrier();
    }
    return;
  }
  const int w = w8;
  const bool valid = row_ < jb.M;          // a ragged last row tile: compute on the last row, store nothing
  const int row = valid ? row_ : jb.M - 1;
  const int D = jb.N, nvq = D >> 8;        // floats per lane (D in {256, 512, 1024}: 1, 2, 4)
  const int o = w * (D >> 2) + lane * nvq;
  float* red = smem;                       // 16 floats at the start of ring stage 0's A image (no prefetch lands there)
  const float* x = rp(jb.A[0], io) + (long long)row * jb.lda[0] + o;
  const f4a v0 = ldn(x, nvq);
  const f4a ww = ldn(rp(jb.W[0], io) + o, nvq), bb = ldn(rp(jb.bias[0], io) + o, nvq);
  f4a km, rr, gg, ww2, bb2;
  if (jb.mask) km = ldn(rp(jb.mask, io) + (long long)row * D + o, nvq);
  if (jb.R) rr = ldn(rp(jb.R, io) + (long long)row * jb.ldr + o, nvq);
  if (jb.gate) gg = ldn(rp(jb.aux[1], io) + (long long)row * jb.lda[1] + o, nvq);
  if (two) {
    ww2 = ldn(rp(jb.W[1], io) + o, nvq);
    bb2 = ldn(rp(jb.bias[1], io) + o, nvq);
  }
  const float sg = jb.gate ? fl::sigmoid_acc(ld1g(rp(jb.gate, io))) : 0.f;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < nvq) s += v0.v[e];
  const float m = block4_sum(s, red, w) / D;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < nvq) q += (v0.v[e] - m) * (v0.v[e] - m);
  const float r = 1.0f / sqrtf(block4_sum(q, red + 4, w) / D + jb.eps);
  f4a y;
  float s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y.v[e] = 0.f;
    if (e < nvq) {
      float t = (v0.v[e] - m) * r * ww.v[e] + bb.v[e];
      if (jb.mask) t *= km.v[e];
      t = fl::apply_act(t, jb.act, 0.f);
      if (jb.R) t += rr.v[e];
      if (jb.gate) t += sg * gg.v[e];
      y.v[e] = t;
      s2 += t;
    }
  }
  if (jb.C && valid) stn(wt, wp(jb.C, io) + (long long)row * jb.ldc + o, y, nvq);
  if (!two) return;
  const float m2 = block4_sum(s2, red + 8, w) / D;
  float q2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < nvq) q2 += (y.v[e] - m2) * (y.v[e] - m2);
  const float r2 = 1.0f / sqrtf(block4_sum(q2, red + 12, w) / D + jb.eps2);
  f4a z;
#pragma unroll
  for (int e = 0; e < 4; ++e) z.v[e] = e < nvq ? (y.v[e] - m2) * r2 * ww2.v[e] + bb2.v[e] : 0.f;
  if (valid) stn(wt, wp(jb.aux[0], io) + (long long)row * jb.ldc + o, z, nvq);
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
  const float m = wsum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nv) {
      const float a = v[i].x - m, b = v[i].y - m, c = v[i].z - m, d = v[i].w - m;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float r = 1.0f / sqrtf(wsum(q) / D + eps);
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
  const float t1 = wsum(s1) / D, t2 = wsum(s2) / D;
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
  const int tidx = threadIdx.x & 255;  // waves 4-7 of the 512-thread block shadow waves 0-3 (same loads, same LDS words, same stores)
  const int c4 = (tidx & 7) * 4, rg = tidx >> 3;
  const float* gl = rp(jb.A[0], io) + k * 32 + c4;
  const float* x = rp(jb.A[1], io) + k * 32 + c4;
  const float* st = rp(jb.aux[1], io);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  for (int base = 0; base < jb.M; base += 1024) {
    const int rows = jb.M - base < 1024 ? jb.M - base : 1024;
    __syncthreads();
    for (int i = tidx; i < 2 * rows; i += 256) *(lf1_t*)(smem + i) = ld1c(st + 2 * base + i);
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
  if (tidx < 64) {
    const int which = tidx >> 5, c = tidx & 31;
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
  const float nrm = sqrtf(wsum(s) + 1e-8f);
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
  dot = wsum(dot);
  if (lane == 0) {
    st1g(wt, wp(jb.aux[0], io) + row, nrm);
    if (jb.aux[1]) st1g(wt, wp(jb.aux[1], io) + row, dot);
  }
}

__device__ __forceinline__ float block_sum_lds(float v, float* smem) {
  const int tidx = threadIdx.x & 255;  // waves 4-7 of the 512-thread block shadow waves 0-3 (same loads, same LDS words, same stores)
  v = wsum(v);
  __syncthreads();
  if ((tidx & 63) == 0) *(lf1_t*)(smem + (tidx >> 6)) = v;
  __syncthreads();
  const float t = (*(const lf1_t*)(smem) + *(const lf1_t*)(smem + 1)) + (*(const lf1_t*)(smem + 2) + *(const lf1_t*)(smem + 3));
  __syncthreads();
  return t;
}

// loss = mean of the row norms; gradient of the residual gate = s(1-s) * sum of the row dots (gate_grad_kernel)
__device__ __forceinline__ void fin_tile(const fl_pjob& jb, const ProgIO& io, float* smem, const bool wt) {
  const int tidx = threadIdx.x & 255;  // waves 4-7 of the 512-thread block shadow waves 0-3 (same loads, same LDS words, same stores)
  float a = 0.f, b = 0.f;
  const float* n = rp(jb.A[0], io);
  const float* d = jb.A[1] ? rp(jb.A[1], io) : nullptr;
  for (int i = tidx; i < jb.M; i += 256) {
    a += ld1c(n + i);
    if (d) b += ld1c(d + i);
  }
  a = block_sum_lds(a, smem);
  b = block_sum_lds(b, smem);
  if (tidx == 0) {
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
  const int tidx = threadIdx.x & 255;  // waves 4-7 of the 512-thread block shadow waves 0-3 (same loads, same LDS words, same stores)
  const float* g = rp(jb.A[0], io);
  int lo, hi;
  flat_slice(jb.K[0], t, lo, hi);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = lo + tidx * 4;
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
  if (tidx == 0) st1g(wt, wp(jb.C, io) + t, tot);
}

// clip_grad_norm_ + Adam / AdamW: adam_kernel of loss_optim.hip with the clip coefficient derived in every tile from
// the SQNORM partials (the same 256 numbers summed in the same order: identical in every tile)
__device__ __forceinline__ void adam_tile(const fl_pjob& jb, int t, const ProgIO& io, float* smem) {
  const int tidx = threadIdx.x & 255;  // waves 4-7 of the 512-thread block shadow waves 0-3 (same loads, same LDS words, same stores)
  const float part = ld1c(rp(jb.W[0], io) + tidx);  // FLAT_TILES == 256 threads
  const float sq = block_sum_lds(part, smem);
  const float gs = jb.alpha;
  const float nrm = gs * sqrtf(sq);
  const float maxn = jb.bias_scale[2];
  float coef = 1.f;
  if (maxn > 0.f) {
    const float cf = maxn / (nrm + 1e-6f);
    coef = cf < 1.f ? cf : 1.f;
  }
  if (t == 0 && tidx == 0) {
    if (jb.aux[2]) *(__attribute__((address_space(1))) float*)wp(jb.aux[2], io) = coef;
    if (jb.aux[3]) *(__attribute__((address_space(1))) float*)wp(jb.aux[3], io) = nrm;
  }
  const float lr = ld1g(rp(jb.W[1], io));
  const float step = (float)(*(const __attribute__((address_space(1))) int*)rp(jb.W[2], io) + 1);
  float b1 = jb.bias_scale[0], b2 = jb.bias_scale[1];
  if (jb.bias[0]) {  // (beta1, beta2) in device memory: a scheduler changes them between replays of one captured graph
    b1 = ld1g(rp(jb.bias[0], io));
    b2 = ld1g(rp(jb.bias[0], io) + 1);
  }
  const float eps = jb.eps, wd = jb.eps2;
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
  if (threadIdx.x >= 256) return;  // the update is a read-modify-write: the shadow waves stop here (no barrier follows)
  for (int i = lo + tidx * 4; i < hi; i += 2048) {  // two float4 of each of the four arrays in flight
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
  __builtin_amdgcn_s_barrier();  // (not __syncthreads(): its fence would wait for the prefetch DMAs too)
}

// Device-wide barrier for a fully resident grid. Returns false if the run is dead (timeout here or in another block):
// every wave then leaves. release: the block's XCD holds dirty lines of plain-stored outputs (local phases) that blocks
// of other XCDs are about to read: write the L2 back first.
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned target, bool acquire, bool release, int pending, int* s_ok) {
  drain_stores(pending);
  if (threadIdx.x == 0) {
    if (release) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop its own wait after buffer_wbl2)
    }
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
  // (raw barrier: __syncthreads() would make the loader waves wait for their prefetch DMAs here)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
  return *(volatile int*)s_ok != 0;
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
  // (raw barrier: __syncthreads() would make the loader waves wait for their prefetch DMAs here)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
  return *(volatile int*)s_ok != 0;
}

// ---------------------------------------------------------------------------------------------- tile dealing
// Per job, next to the descriptors in LDS: jinfo[j] = { nrt << 16 | per, tile_begin | swizzle << 31 | prefetchable << 30,
// magic(per), magic(nrt) } with magic(d) = floor(2^32 / d) + 1 (0 for d = 1): n / d == umulhi(n, magic) for n, d < 2^16 -
// an integer division by a run-time divisor is ~40 instructions, several of them per tile.
// A phase's window = those words of its (<= 64) jobs, one per lane, read ONCE per phase; every dealing decision is then
// v_readlane + scalar arithmetic (a dependent LDS read per field cost ~100 cycles each: 2 us per tile before).
constexpr int JW = sizeof(fl_pjob) / 4;
constexpr unsigned JI_SWZ = 1u << 31, JI_PF = 1u << 30, JI_TB = (1u << 30) - 1;
struct Win {
  int jb0, jb1, ntiles, flags;
  int dimv, tbv, mperv, mnrtv;  // per lane: lane l holds the words of job jb0 + l
};
__device__ __forceinline__ int udiv_magic(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }
__device__ __forceinline__ void load_win(const int* lph, const int* ljinfo, int p, Win& w) {
  const ldsint_p ph = (ldsint_p)(lph + 4 * p);
  w.jb0 = __builtin_amdgcn_readfirstlane(ph[0]);
  w.jb1 = __builtin_amdgcn_readfirstlane(ph[1]);
  w.ntiles = __builtin_amdgcn_readfirstlane(ph[2]);
  w.flags = __builtin_amdgcn_readfirstlane(ph[3]);
  int j = w.jb0 + (int)(threadIdx.x & 63);
  j = j < w.jb1 ? j : w.jb1 - 1;
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  const i32x4v v = *((const __attribute__((address_space(3))) i32x4v*)ljinfo + j);
  w.dimv = v[0];
  w.tbv = v[1];
  w.mperv = v[2];
  w.mnrtv = v[3];
}
__device__ __forceinline__ unsigned win_mper(const Win& w, int j) { return (unsigned)__builtin_amdgcn_readlane(w.mperv, j - w.jb0); }
__device__ __forceinline__ unsigned win_mnrt(const Win& w, int j) { return (unsigned)__builtin_amdgcn_readlane(w.mnrtv, j - w.jb0); }
__device__ __forceinline__ int win_dim(const Win& w, int j) { return __builtin_amdgcn_readlane(w.dimv, j - w.jb0); }
__device__ __forceinline__ unsigned win_tb(const Win& w, int j) { return (unsigned)__builtin_amdgcn_readlane(w.tbv, j - w.jb0); }

// a tile of a phase as this block sees it
struct Cur {
  int j;     // job index (-1: none)
  int t;     // local phases: tile index among the job's tiles of this XCD; device-wide phases: tile index in the phase
  int base;  // local phases: tiles of this XCD in the phase's earlier jobs
  int cnt;   // local phases: tiles of this XCD in job j
};
struct Deal {
  int xcc, slot, nx, G;
};

// local phase: the first tile at or after job j (given `base` tiles of this XCD in the earlier jobs) that belongs to this block:
// XCD-tile t of a job runs on the block with slot == (base + t) % nx
__device__ __forceinline__ Cur local_seek(const Deal& d, const Win& w, int j, int base) {
  Cur c;
  for (; j < w.jb1; ++j) {
    const int dim = win_dim(w, j);
    const int nrt = dim >> 16, per = dim & 0xFFFF;
    const int mine = nrt > d.xcc ? (nrt - d.xcc + NCTR - 1) / NCTR : 0;
    const int cnt = mine * per;
    const int t = (d.slot - base) & (d.nx - 1);  // nx is a power of two (fl_program_run)
    if (t < cnt) {
      c.j = j; c.t = t; c.base = base; c.cnt = cnt;
      return c;
    }
    base += cnt;
  }
  c.j = -1; c.t = 0; c.base = base; c.cnt = 0;
  return c;
}
__device__ __forceinline__ Cur global_at(const Win& w, int t) {
  Cur c;
  c.base = 0; c.cnt = 0; c.t = t; c.j = -1;
  if (t >= w.ntiles) return c;
  // the job whose [tile_begin, ...) holds t: count the jobs of the phase that begin at or before t
  const int lane = threadIdx.x & 63;
  const bool le = lane < w.jb1 - w.jb0 && (int)((unsigned)w.tbv & JI_TB) <= t;
  c.j = w.jb0 + __builtin_popcountll(__builtin_amdgcn_ballot_w64(le)) - 1;
  return c;
}
__device__ __forceinline__ Cur first_tile(const Deal& d, const Win& w, bool local) {
  if (local) return local_seek(d, w, w.jb0, 0);
  return global_at(w, (int)blockIdx.x);
}
__device__ __forceinline__ Cur next_tile(const Deal& d, const Win& w, bool local, const Cur& c) {
  if (!local) return global_at(w, c.t + d.G);
  if (c.t + d.nx < c.cnt) {
    Cur n = c;
    n.t = c.t + d.nx;
    return n;
  }
  return local_seek(d, w, c.j + 1, c.base + c.cnt);
}
// (row tile, tile inside the row tile) of a dealt tile
__device__ __forceinline__ void tile_coords(const Deal& d, const Win& w, bool local, const Cur& c, int& rt, int& k) {
  const int dim = win_dim(w, c.j);
  const int nrt = dim >> 16, per = dim & 0xFFFF;
  if (local) {
    const int rl = udiv_magic(c.t, win_mper(w, c.j));
    k = c.t - rl * per;
    rt = d.xcc + NCTR * rl;
    return;
  }
  const unsigned tb = win_tb(w, c.j);
  const int lt = c.t - (int)(tb & JI_TB);
  if (tb & JI_SWZ) {
    // GEMM with a multiple of 8 column tiles: tile lt runs on XCD lt % 8 (round-robin block placement): keep a weight
    // column block on one XCD
    const int x = lt & 7, jj = lt >> 3, jc = udiv_magic(jj, win_mnrt(w, c.j));
    k = x + 8 * jc;
    rt = jj - jc * nrt;
  } else {
    rt = udiv_magic(lt, win_mper(w, c.j));
    k = lt - rt * per;
  }
}
__device__ __forceinline__ bool prefetchable(const Win& w, const Cur& c) { return c.j >= 0 && (win_tb(w, c.j) & JI_PF) != 0; }

// returns the prefetch DMA instructions the wave left in flight for the block's next tile
// (tjw, tnext, trt, tk): the tile this block runs next IN THIS PHASE, if any (the weight-gradient tiles stream it in behind themselves)
__device__ __forceinline__ int run_tile(const jobwords_t jw, int rt, int k, const ProgIO& io, float* smem, const bool wt, const int pre,
                                        const jobwords_t& njw, const bool prefetch, const int pf_ct, const jobwords_t& tjw,
                                        const bool tnext, const int trt, const int tk) {
  const int kind = ldsi(jw, FL_JW(kind)), flags = ldsi(jw, FL_JW(flags));
  if (kind == FL_PJOB_GEMM && (flags & FL_PG_DMA)) {
    if (flags & FL_PG_TW) return gemm_dma_tile<true>(jw, rt, k, io, smem, wt, pre, njw, prefetch, pf_ct);
    return gemm_dma_tile<false>(jw, rt, k, io, smem, wt, pre, njw, prefetch, pf_ct);
  }
  fl_pjob jb;  // (each case reads the fields its tile uses: the rest of the unpacking is dead code there)
  load_job(jw, jb);
  const int r0 = rt * TS, r1 = r0 + TS < jb.M ? r0 + TS : jb.M;
  const bool shadow = threadIdx.x >= 256;  // waves 4-7 only matter to the tiles that meet at workgroup barriers
  switch (kind) {
    case FL_PJOB_GEMM:
      switch (flags & (FL_PG_TA | FL_PG_TW)) {
        case 0: gemm_tile<false, false>(jb, rt, k, io, smem, wt); break;
        case FL_PG_TW: gemm_tile<false, true>(jb, rt, k, io, smem, wt); break;
        case FL_PG_TA | FL_PG_TW:
          if (flags & FL_PG_T64) return wg64_tile(jb, rt, k, io, smem, wt, pre, tjw, tnext, trt, tk);
          gemm_tile<true, true>(jb, rt, k, io, smem, wt);
          break;
        default: break;  // TA alone: rejected by fl_program_job_tiles
      }
      break;
    case FL_PJOB_ROWNORM: rownorm_tile(jb, r0 + k, io, smem, wt); break;
    case FL_PJOB_SINUSOID: if (!shadow) sinusoid_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_GATHER: if (!shadow) gather_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_PSAMPLE: if (!shadow) psample_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_LNBWD: if (!shadow) lnbwd_tile(jb, r0 + 4 * k, io, wt); break;
    case FL_PJOB_LNCOL: lncol_tile(jb, k, io, smem, wt); break;
    case FL_PJOB_LOSS: if (!shadow) loss_tile(jb, r0 + 4 * k, io, wt); break;
    case FL_PJOB_FIN: fin_tile(jb, io, smem, wt); break;
    case FL_PJOB_EWADD: if (!shadow) ewadd_tile(jb, r0, r1, k, io, wt); break;
    case FL_PJOB_EMBBWD: if (!shadow) embbwd_tile(jb, k, io, smem, wt); break;
    case FL_PJOB_SQNORM: sqnorm_tile(jb, k, io, smem, wt); break;
    case FL_PJOB_ADAM: adam_tile(jb, k, io, smem); break;
    default: break;
  }
  return pre;  // (a weight stream started for a later GEMM tile stays in flight across ring-safe tiles)
}

// Device-wide phases: any block may run any tile, outputs are stored write-through, the phase ends in the device-wide
// barrier. Local phases (FL_PPHASE_LOCAL, honoured when the launch asked for FL_PROG_XCD_LOCAL and the grid is spread
// evenly over the XCDs): the 32-row tiles of the batch are dealt to the XCDs (row tile rt belongs to XCD rt % 8, the XCD
// being what the block READS from HW_REG_XCC_ID, not an assumption about placement), a block only ever runs tiles of its
// own XCD's rows, data is handed from phase to phase through that XCD's L2 (plain stores, loads that hit L2 instead of
// memory) and two local phases are separated by xcd_barrier among the 32 blocks of the XCD. Every XCD then streams all the
// weights (8 x the weight traffic, out of the 256 MB Infinity Cache): the first weight chunks of a block's next tile are
// requested BEFORE the barrier (gemm_dma_tile).
// 512 threads: waves 4-7 are the LDS-DMA loaders of the GEMM tiles; the other tile kinds run on waves 0-3 (the loaders
// shadow them where a tile meets at workgroup barriers, and skip the rest).
__global__ __launch_bounds__(512, 1) void program_kernel(const fl_pphase* __restrict__ phases, const fl_pjob* __restrict__ jobs,
                                                         int nphases, int njobs, const ProgIO io, unsigned* sync, int allow_local,
                                                         const unsigned* guard) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_ok, s_slot;
  __shared__ unsigned s_epoch;
  // the program lives in LDS for the whole run: phases, jobs, jinfo
  int* lprog = reinterpret_cast<int*>(smem + 2 * STAGE);
  const int nph_w = nphases * (int)(sizeof(fl_pphase) / 4);
  int* ljobs = lprog + nph_w;
  int* ljinfo = ljobs + njobs * JW;
  {
    const int* gp = reinterpret_cast<const int*>(phases);
    for (int i = threadIdx.x; i < nph_w; i += 512) ((__attribute__((address_space(3))) int*)lprog)[i] = gp[i];
    // jobs: 64-bit words; the leading pointer fields that name an io slot (1..FL_PROG_MAX_IO) are bound here, once,
    // so that no tile ever waits on a kernel-argument fetch
    constexpr int JQ = sizeof(fl_pjob) / 8, PTRQ = offsetof(fl_pjob, kind) / 8;
    const unsigned long long* gj = reinterpret_cast<const unsigned long long*>(jobs);
    __attribute__((address_space(3))) unsigned long long* lj = (__attribute__((address_space(3))) unsigned long long*)ljobs;
    for (int i = threadIdx.x; i < njobs * JQ; i += 512) {
      unsigned long long v = gj[i];
      if (i % JQ < PTRQ && v >= 1 && v <= FL_PROG_MAX_IO) v = reinterpret_cast<unsigned long long>(io.p[v - 1]);
      lj[i] = v;
    }
    for (int j = threadIdx.x; j < njobs; j += 512) {
      const fl_pjob& g = jobs[j];
      const unsigned pf = (g.kind == FL_PJOB_GEMM && (g.flags & (FL_PG_DMA | FL_PG_WCONST)) == (FL_PG_DMA | FL_PG_WCONST)) ? JI_PF : 0u;
      const unsigned swz = (g.kind == FL_PJOB_GEMM && (g.per & 7) == 0) ? JI_SWZ : 0u;
      ((__attribute__((address_space(3))) int*)ljinfo)[4 * j] = (g.nrt << 16) | (g.per & 0xFFFF);
      ((__attribute__((address_space(3))) int*)ljinfo)[4 * j + 1] = (int)((unsigned)g.tile_begin | swz | pf);
      ((__attribute__((address_space(3))) int*)ljinfo)[4 * j + 2] = g.per > 1 ? (int)(0xFFFFFFFFu / (unsigned)g.per + 1u) : 0;
      ((__attribute__((address_space(3))) int*)ljinfo)[4 * j + 3] = g.nrt > 1 ? (int)(0xFFFFFFFFu / (unsigned)g.nrt + 1u) : 0;
    }
  }
  const unsigned G = gridDim.x;
  Deal d;
  d.G = (int)G;
  d.xcc = 0;
  d.slot = 0;
  d.nx = 1;
  // a run that timed out left the abort word set and the arrival counters un-armed: every later launch must do NOTHING
  // (its barriers would fall through on the stale counters and the phases would compute garbage) until the host has
  // seen the word, re-armed the counters and raised (Program.check / Program.run)
  if (__hip_atomic_load(&sync[ABORT_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  // ... and a program whose INPUT comes from a run that was voided (the optimizer behind the training program) must not apply it
  if (guard && __hip_atomic_load(const_cast<unsigned*>(&guard[ABORT_WORD]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
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
  int pre = 0;                   // (loader waves) prefetch DMAs in flight for the next tile to run
  // the tile this block runs next, fully decoded: job descriptor in scalar registers, coordinates
  Win cw, nw;
  load_win(lprog, ljinfo, 0, cw);
  nw = cw;
  bool local = allow_local && (cw.flags & FL_PPHASE_LOCAL) != 0;
  Cur cur = first_tile(d, cw, local);
  bool have = cur.j >= 0;
  jobwords_t jw = 0;  // the descriptor of `cur`, one word per lane
  int rt = 0, k = 0;
  if (have) {
    jw = fetch_job(ljobs + cur.j * JW);
    tile_coords(d, cw, local, cur, rt, k);
  }
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
    if (threadIdx.x == 0 && g_trace) g_trace[((size_t)blockIdx.x * g_trace_phases + g_trace_p) * 16 + 11] = clock64();
#endif
    const bool wt = !local;
    const bool has_next = p + 1 < nphases;
    if (has_next) load_win(lprog, ljinfo, p + 1, nw);
    FL_STAMP(12);
    const bool nlocal = has_next && allow_local && (nw.flags & FL_PPHASE_LOCAL) != 0;
    bool carried = false;  // (cur, jw, rt, k) already describe this block's first tile of phase p + 1
    while (have) {
      const Cur nxt = next_tile(d, cw, local, cur);
      // what this block runs after `cur`: the phase's next tile, or its first tile of the next phase
      Cur pf = nxt;
      const bool pf_next = nxt.j < 0 && has_next;
      if (pf_next) pf = first_tile(d, nw, nlocal);
      const bool pf_ok = pf.j >= 0;
      const bool pf_w = pf_ok && prefetchable(pf_next ? nw : cw, pf);
      jobwords_t jwn = 0;
      int nrt = 0, nk = 0;
      if (pf_ok) {
        tile_coords(d, pf_next ? nw : cw, pf_next ? nlocal : local, pf, nrt, nk);
        jwn = fetch_job(ljobs + pf.j * JW);  // (an LDS read in flight: nothing waits for it before the tile's end)
      }
      FL_STAMP(13);
      // whose weights to stream once this tile's loop is over: the next tile's - or, from the block's last tile of a phase
      // that is followed by a ring-safe phase (the row phase between two GEMM phases), its first tile of the phase after
      bool pw = pf_w;
      jobwords_t jwp = jwn;
      int pk = nk;
      if (!pw && nxt.j < 0 && p + 2 < nphases && (nw.flags & FL_PPHASE_RINGSAFE) != 0) {
        Win w2;
        load_win(lprog, ljinfo, p + 2, w2);
        const bool local2 = allow_local && (w2.flags & FL_PPHASE_LOCAL) != 0;
        const Cur t2 = first_tile(d, w2, local2);
        if (prefetchable(w2, t2)) {
          int prt;
          tile_coords(d, w2, local2, t2, prt, pk);
          jwp = fetch_job(ljobs + t2.j * JW);
          pw = true;
        }
      }
      FL_STAMP(3);  // the next tile(s) are dealt and their descriptors requested
      pre = run_tile(jw, rt, k, io, smem, wt, pre, jwp, pw, pk, jwn, nxt.j >= 0, nrt, nk);
      have = nxt.j >= 0;
      if (pf_ok) {
        jw = jwn;
        rt = nrt;
        k = nk;
        cur = pf;
        carried = pf_next;
      }
    }
    if (has_next && !carried) {  // (no tile here, or none left over: look the next phase's first tile up before the barrier)
      cur = first_tile(d, nw, nlocal);
      carried = cur.j >= 0;
      if (carried) {
        jw = fetch_job(ljobs + cur.j * JW);
        tile_coords(d, nw, nlocal, cur, rt, k);
      }
    }
    dirty = dirty || local;
    FL_STAMP(2);
#if FL_PROG_TRACE
    if (threadIdx.x == 0 && g_trace) {
      unsigned long long* e = &g_trace[((size_t)blockIdx.x * g_trace_phases + g_trace_p) * 16 + 11];
      *e = clock64() - *e;  // core-clock cycles spent on this phase's tiles
    }
#endif
    if (has_next) {
      const bool acq = (nw.flags & FL_PPHASE_ACQUIRE) != 0;
      if (local && nlocal) {
        ++lbar;
        alive = xcd_barrier(sync, d.xcc, d.slot, d.nx, (epoch << 12) + lbar, acq, pre, &s_ok);
      } else {
        ++gbar;
        alive = grid_barrier(sync, gbar * G, acq, dirty, pre, &s_ok);
        dirty = false;
      }
      cw = nw;
      local = nlocal;
      have = carried;
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
      for (int kq = 0; kq < 2; ++kq)
        if (io.post.ptr64[kq]) *(__attribute__((address_space(1))) long long*)(io.post.ptr64[kq]) += io.post.delta64[kq];
      if (io.post.ptr32) *(__attribute__((address_space(1))) int*)(io.post.ptr32) += io.post.delta32;
    }
  }
}

// FL_PG_WPACK image of a row-major weight (fl_program_pack_weight): one thread per 16-byte block of the output
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K, int ldw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // output block index
  const long long total = (long long)N * K / 4;
  if (i >= total) return;
  const int pb = (int)(i & 31);
  const int r = (int)((i >> 5) & 31);
  const long long tc = i >> 10;  // ct * nC + c
  const int nC = K / DKC;
  const int ct = (int)(tc / nC), c = (int)(tc - (long long)ct * nC);
  const float* src = w + (size_t)(ct * TS + r) * ldw + c * DKC + 4 * (pb ^ (r & 15));
  *reinterpret_cast<float4*>(out + i * 4) = *reinterpret_cast<const float4*>(src);
}

// tiles a job contributes: nrt row tiles x per tiles each; the kinds that are not laid out by row tiles count as ONE row tile
bool flat_kind(int kind) { return kind == FL_PJOB_LNCOL || (kind >= FL_PJOB_FIN && kind != FL_PJOB_EWADD); }
bool t64(const fl_pjob& jb) { return jb.kind == FL_PJOB_GEMM && (jb.flags & FL_PG_T64) != 0; }
int row_tiles(const fl_pjob& jb) { return flat_kind(jb.kind) ? 1 : t64(jb) ? jb.M / 64 : (jb.M + TS - 1) / TS; }
int tiles_per_rowtile(const fl_pjob& jb) {
  switch (jb.kind) {
    case FL_PJOB_GEMM: return (jb.flags & FL_PG_T64) ? jb.N / 64 : jb.N / TS;
    case FL_PJOB_ROWNORM: return TS;  // one row per tile
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
      return ((j.M + TS - 1) / TS) * TS;
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
  FL_REQUIRE(j.nrt * j.per == n && n < 65536 && j.nrt < 32768, "program job: tile layout %d x %d != %d (or >= 2^16 tiles)", j.nrt, j.per, n);
  j.flags &= ~FL_PG_DMA;
  if (j.kind == FL_PJOB_GEMM && !(j.flags & FL_PG_TA)) {
    bool ok = true;
    for (int s = 0; s < j.nseg; ++s) {
      // (not TW: K_s is the contraction length of both operands; TW: W has K_s rows of N floats)
      ok = ok && j.K[s] % DKC == 0 && (j.A[s] <= FL_PROG_MAX_IO || j.A[s] % 16 == 0) && (j.W[s] <= FL_PROG_MAX_IO || j.W[s] % 16 == 0);
    }
    if (ok) j.flags |= FL_PG_DMA;
  }
  FL_REQUIRE(!(j.flags & FL_PG_WPACK) || ((j.flags & FL_PG_DMA) && !(j.flags & FL_PG_TW)),
             "program GEMM: FL_PG_WPACK needs an LDS-DMA job with W[n][k] operands (every K_s %% 128 == 0)");
  return n;
}

int fl_program_pack_weight(const float* w, float* out, int N, int K, int ldw, fl_stream_t stream) {
  FL_REQUIRE(w && out && N >= TS && N % TS == 0 && K >= DKC && K % DKC == 0 && ldw >= K && ldw % 4 == 0 &&
                 (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
             "fl_program_pack_weight: N=%d K=%d ldw=%d (N %% 32, K %% 128, 16-byte aligned operands)", N, K, ldw);
  const long long blocks = ((long long)N * K / 4 + 255) / 256;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, out, N, K, ldw);
  return fl::check_launch("pack_weight_kernel");
}

int fl_program_sync_bytes(void) { return SYNC_UINTS * 4; }

int fl_program_run(const void* phases, const void* jobs, int n_phases, int n_jobs, const void* const* io, int n_io,
                   void* sync, int max_blocks, int flags, const void* post, const void* guard_sync, fl_stream_t stream) {
  FL_REQUIRE(phases && jobs && sync && n_phases >= 1 && n_jobs >= 1 && n_io >= 0 && n_io <= FL_PROG_MAX_IO && (n_io == 0 || io),
             "fl_program_run: bad args (n_phases=%d n_jobs=%d n_io=%d)", n_phases, n_jobs, n_io);
  FL_REQUIRE((size_t)n_phases * sizeof(fl_pphase) + (size_t)n_jobs * (sizeof(fl_pjob) + 16) <= (size_t)PROG_LDS_BYTES,
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
  const int nx = G / NCTR;
  const int allow_local = (flags & FL_PROG_XCD_LOCAL) != 0 && G == cus && G % NCTR == 0 && nx <= 32 && (nx & (nx - 1)) == 0;
  hipLaunchKernelGGL(program_kernel, dim3(G), dim3(512), LDS_BYTES, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const fl_pphase*>(phases), reinterpret_cast<const fl_pjob*>(jobs), n_phases, n_jobs, pio,
                     reinterpret_cast<unsigned*>(sync), allow_local, reinterpret_cast<const unsigned*>(guard_sync));
  return fl::check_launch("program_kernel");
}

}  // extern "C"
