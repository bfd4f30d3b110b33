// Dense-contraction core for gfx950: one templated LDS-tiled kernel on the
// fp32-in/fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32, exact fp32, 64 FLOP/clk/SIMD).
//
// Every GEMM-shaped op of the v2 module surface goes through this file:
//   nn.Linear fwd / dgrad / wgrad, nn.Conv2d fwd / dgrad / wgrad as implicit GEMM over
//   NHWC activations (im2col is never materialised: the A (or B) tile loader gathers
//   taps straight from HBM into LDS), nn.ConvTranspose2d(4,2,1) as 4 sub-pixel classes.
//
// Tiling: block = 256 threads = 4 waves (64 lanes each); block tile BM x BN x 32;
// each wave owns a (BM/WGM) x (BN/WGN) sub-tile made of 32x32 MFMA tiles whose
// accumulators live in VGPR/AGPRs. Global -> registers -> LDS staging, double-buffered
// in LDS, one barrier per K-chunk; the next chunk's global loads are issued before the
// current chunk's MFMAs so HBM/L2 latency hides under the (long) fp32 MFMA phase.
//
// LDS images (conflict-free by construction, see MI355X_MICROARCH §LDS):
//   KC operand (reduction index contiguous in memory):  tile[rows][36]  (32 + 4 pad)
//       written with ds_write_b128, read with ds_read_b128: slot = 9*row + k/4 (mod 16)
//       is a permutation over any 16 consecutive rows.
//   KS operand (reduction index strided in memory):     tile[32][rows]
//       written with ds_write_b128 along rows, read with ds_read_b32 (32 consecutive
//       rows per half-wave -> conflict-free).
// The k-slot assignment is the same for both images: lane half h, step q, element j
// uses k = 4*(2q+h)+j of the 32-chunk, so A and B fragments always pair the same k.
#include "fl_common.h"
#include <cstring>
#include <cstdlib>
#ifndef FL_ABLATE
#define FL_ABLATE 0
#endif
#ifndef FL_USE_DMA
#define FL_USE_DMA 0  // direct-to-LDS staging: correct, measured 3% slower than register staging on gfx950 (r01)
#endif
// FL_SPLIT = 0: v_mfma_f32_32x32x2_f32 (fp32 operands on the fp32 matrix path, 256 flop/clk/CU).
// FL_SPLIT = 6 | 8: every fp32 operand is split EXACTLY into three 8-bit-mantissa pieces x = h + m + l (bf16 by
// truncation; x - trunc(x) is exact), and the product is assembled from 6 (or 8) of the 9 piece products with
// v_mfma_f32_32x32x16_bf16 (4096 flop/clk/CU): piece products are exact in fp32, accumulation is fp32 as before.
// Dropped terms: m*l, l*m, l*l with 6 (<= 2^-23 relative per product, the size of one fp32 rounding); only l*l
// (2^-32) with 8. 16 k-values cost 6 x 32 = 192 (8 x 32 = 256) matrix-pipe cycles instead of 8 x 64 = 512.
// Both variants are compiled (template parameter SPLIT = 0 | FL_SPLIT_TERMS); fl_set_gemm_math / FL_GEMM_MATH select
// at run time.
#ifndef FL_SPLIT_TERMS
#define FL_SPLIT_TERMS 6
#endif
#ifndef FL_MFMA_ROT
#define FL_MFMA_ROT 1  // issue consecutive MFMAs on different accumulators
#endif

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct Split3 {
  u32x4_t h, m, l;  // 8 bf16 each: element j of this lane's k-slots (two quarters of a chunk: 4 + 4 floats)
};
__device__ __forceinline__ Split3 split3(const float4& q0, const float4& q1) {
  const float x[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#if FL_ABLATE & 16
  Split3 t;  // timing-only: no and/sub chain, the three pieces are the same truncation
#pragma unroll
  for (int p = 0; p < 4; ++p)
    t.h[p] = t.m[p] = t.l[p] = __builtin_amdgcn_perm(__float_as_uint(x[2 * p + 1]), __float_as_uint(x[2 * p]), 0x07060302u);
  return t;
#endif
  // round-to-nearest pieces (v_cvt_pk_bf16_f32): h = rn(x), m = rn(x - h), l = rn(x - h - m). x - h and x - h - m are exact
  // in fp32; the pieces' signs alternate, so what the 6-product scheme drops (m*l, l*m, l*l <= 2^-26 |xy|) is unbiased -
  // the truncating split of round 1 dropped up to 2^-23 |xy| with the sign of xy (a bias that showed up as 2-8x larger
  // gradient errors than the native fp32 MFMA on heavily cancelling sums, tools/vaegan_grad_diag.py)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  Split3 s;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = x[2 * p], x1 = x[2 * p + 1];
    const bf16x2_t h = {(__bf16)x0, (__bf16)x1};
    const float a0 = x0 - (float)h[0], a1 = x1 - (float)h[1];
    const bf16x2_t m = {(__bf16)a0, (__bf16)a1};
    const float b0 = a0 - (float)m[0], b1 = a1 - (float)m[1];
    const bf16x2_t l = {(__bf16)b0, (__bf16)b1};
    s.h[p] = __builtin_bit_cast(unsigned, h);
    s.m[p] = __builtin_bit_cast(unsigned, m);
    s.l[p] = __builtin_bit_cast(unsigned, l);
  }
  return s;
}

constexpr int BK = 32;
constexpr int LDK = 36;

enum { A_KC_PLAIN = 0, A_KC_CONV = 1, A_KC_TCONV = 2, A_KS_PLAIN = 3 };
enum { B_KC_PLAIN = 0, B_KS_PLAIN = 1, B_KS_TAPFLIP = 2, B_KS_TAPCLS = 3, B_KS_IM2COL = 4, B_FRAGS = 5 };
// B_FRAGS (split math only): the B operand (weights, constant over the batch) was split into its three bf16 pieces
// ONCE, on the host side of the step (fl_split_pack_weight), and laid out in MFMA-fragment order:
//   [16-wide k group][32-column tile][piece h|m|l][lane 0..63][8 bf16]   (k slots of a lane as in frag_load)
// so a wave fetches a B fragment with one coalesced 1 KiB global_load_dwordx4 straight into the MFMA operand
// registers: no LDS image, no staging registers and - the point - no split work for B in the VALU-bound inner loop.
enum { OUT_ROWMAJOR = 0, OUT_NCHW = 1, OUT_TCONV = 2 };

struct GArgs {
  int M, N, K;
  // A operand
  const float* A;
  long long lda;
  int a_vec;  // 16-B vector loads legal
  // gather geometry shared by A_KC_CONV / A_KC_TCONV / B_KS_IM2COL
  int H, W, Cin;   // spatial dims + channels of the tensor being gathered from
  int Ho, Wo;      // dims the GEMM row (or reduction) index is decoded over
  int kw, stride, pad;
  // B operand
  const float* B;
  long long ldb;
  int b_vec;
  int b_taps, b_Cred, b_Cn;  // tapped weights (dgrad / transposed conv)
  // epilogue
  float* C;
  long long ldc;
  int out_mode;
  const float* bias;
  float bias_scale;
  const float* R;
  long long ldr;
  float r_scale;
  int r_pre;
  int act;
  float act_param;
  // tiling
  int tiles_n;
  int splitk, chunks_per_split;
  float* ws;
  int prio_mode;
  float* stats;  // optional [M/WM][2][N] per-wave column sums (sum y, sum y^2) of the epilogue output (fused LayerNorm2d stats)
  int hw_shift, wo_shift;  // >= 0 when Ho*Wo / Wo are powers of two (pixel decode by shifts), else -1
};

// 16 zero bytes: source of out-of-bounds / padding lanes of direct-to-LDS loads
__device__ const float4 g_zero16 = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// guarded load of up to 4 consecutive floats (nvalid in 0..4), vector when legal
__device__ __forceinline__ float4 ld4g(const float* p, int nvalid, int vec) {
  if (nvalid >= 4 && vec) return ld4(p);
  float4 v = zero4();
  if (nvalid > 0) v.x = p[0];
  if (nvalid > 1) v.y = p[1];
  if (nvalid > 2) v.z = p[2];
  if (nvalid > 3) v.w = p[3];
  return v;
}

// FAST: branch-free predicated vector load (clamped address + select) — legal when every valid float4 is whole
// and 16-byte aligned; otherwise the guarded (slow, branchy) form handles tails / unaligned operands.
// one direct-to-LDS 16-byte load per lane: lds_piece is the wave-uniform 1-KiB destination
__device__ __forceinline__ void dma16(const float* p, bool ok, float* lds_piece) {
  const float* src = ok ? p : reinterpret_cast<const float*>(&g_zero16);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_piece, 16, 0, 0);
}

template <bool FAST>
__device__ __forceinline__ float4 ldsel(const float* p, bool ok, int nvalid, int vec, const float* safe) {
  if constexpr (FAST) {
    return ld4(ok ? p : safe);  // zeroing of !ok lanes is deferred to the LDS store (keeps the load in flight)
  } else {
    return ok ? ld4g(p, nvalid, vec) : zero4();
  }
}

__device__ __forceinline__ float epilogue_value(const GArgs& g, float v, int col, const float* rptr) {
  if (g.bias) v += g.bias_scale * g.bias[col];
  if (rptr && g.r_pre) v += g.r_scale * (*rptr);
  v = fl::apply_act(v, g.act, g.act_param);
  if (rptr && !g.r_pre) v += g.r_scale * (*rptr);
  return v;
}

// PATH 0: guarded loads (tails / unaligned operands); 1: branch-free vector loads, per-thread tap decode;
// 2: as 1 + wave-uniform tap decode (Cin % 32 == 0: a 32-chunk never straddles a tap) kept in scalar registers.
template <int BM, int BN, int WGM, int WGN, int AM, int BMODE, int PATH, int SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void gemm_kernel(const GArgs g) {
  constexpr bool FAST = PATH >= 1;
  constexpr bool CHK = PATH == 2;
  static_assert(WGM * WGN == 4, "4 waves per block");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int ANR = BM / 32, BNR = BN / 32;  // float4 per thread per chunk
  constexpr bool A_KS = (AM == A_KS_PLAIN);
  constexpr bool B_FR = (BMODE == B_FRAGS);
  static_assert(!B_FR || (SPLIT == 6 && !A_KS), "B_FRAGS needs the 6-term split math and a K-contiguous A");
  constexpr bool B_KS = (BMODE != B_KC_PLAIN) && !B_FR;
  // PATH >= 1 stages global -> LDS directly (LDS-DMA, global_load_lds_dwordx4): no VGPR staging, no ds_write, no
  // zero-selects. A wave-instruction writes 1 KiB contiguously (wave-uniform base + lane*16), so the K-contiguous
  // image is UNPADDED [rows][32] with an XOR swizzle on the 16-byte chunk index applied on the SOURCE side
  // (lane l of a piece loads logical chunk (l&7) ^ ((row>>1)&7)) and again on the fragment read; over the 16 rows
  // of a ds_read_b128 lane group (row&1, (row>>1)&7) takes 16 distinct values -> conflict-free. K-strided images
  // ([32][rows]) are lane-linear already. Padding / out-of-range lanes read 16 zero bytes (g_zero16).
  constexpr bool DMA = (PATH >= 1) && (FL_USE_DMA != 0);
  constexpr int KCS = DMA ? BK : LDK;                   // row stride (floats) of a K-contiguous image
  constexpr int A_TILE = BM * KCS, B_TILE = BN * KCS;   // floats per buffer (the K-strided image is 32*rows <= this)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_TILE;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WGN, wc = wid % WGN;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: hardware deals consecutive block ids round-robin over the 8 XCDs (private L2s), so remap
  // id -> tile such that each XCD owns a contiguous range of tiles (the N-tiles of an M-tile and neighbouring
  // M-tiles, which share conv halo rows, then hit the same L2). Bijective for any grid size. Speed only.
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m0 = (tile / g.tiles_n) * BM;
  const int tile_n0 = (tile % g.tiles_n) * BN;
  const int cls = blockIdx.y;  // sub-pixel class (transposed conv only)
  const int py = cls >> 1, px = cls & 1;
  const int z = blockIdx.z;

  int kc_begin = 0, kc_end = (g.K + BK - 1) / BK;
  if (g.splitk > 1) {
    kc_begin = z * g.chunks_per_split;
    int e = kc_begin + g.chunks_per_split;
    kc_end = e < kc_end ? e : kc_end;
  }

  // ------------------------------------------------------------------ A loader state
  // KC mapping: thread -> (row = tid>>3 (+32r), 4 consecutive k at (tid&7)*4)
  // DMA: the thread's logical 16-byte chunk is XOR-swizzled by its row ((row>>1)&7 == ((lane>>4) + 4*wid) & 7)
  const int ld_row = tid >> 3;
  const int kq = DMA ? (((tid & 7) ^ (((lane >> 4) + 4 * wid) & 7)) * 4) : (tid & 7) * 4;
  // KS mapping for an operand of R rows: TPR = R/4 threads per k-row
  constexpr int A_TPR = BM / 4, A_RPP = 256 / A_TPR;
  constexpr int B_TPR = BN / 4, B_RPP = 256 / B_TPR;
  const int a_ks_row = tid / A_TPR, a_ks_c = (tid % A_TPR) * 4;
  const int b_ks_row = tid / B_TPR, b_ks_c = (tid % B_TPR) * 4;

  long long a_base[ANR];
  int a_iy0[ANR], a_ix0[ANR];
  bool a_ok[ANR];
  if constexpr (!A_KS) {
#pragma unroll
    for (int r = 0; r < ANR; ++r) {
      const int m = tile_m0 + ld_row + 32 * r;
      a_ok[r] = m < g.M;
      a_iy0[r] = 0;
      a_ix0[r] = 0;
      if constexpr (AM == A_KC_PLAIN) {
        a_base[r] = (long long)m * g.lda;
      } else {
        const int hw = g.Ho * g.Wo;
        const int mm = a_ok[r] ? m : 0;
        const int img = mm / hw, p = mm - img * hw;
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        if constexpr (AM == A_KC_CONV) {
          a_iy0[r] = oy * g.stride - g.pad;
          a_ix0[r] = ox * g.stride - g.pad;
        } else {  // A_KC_TCONV: rows are (img, yh, xh) of the low-res tensor
          a_iy0[r] = oy + py;
          a_ix0[r] = ox + px;
        }
        a_base[r] = (((long long)img * g.H + a_iy0[r]) * g.W + a_ix0[r]) * g.Cin;
      }
    }
  }

  // ------------------------------------------------------------------ B loader state
  long long b_base[BNR];
  bool b_ok[BNR];
  int b_ky = 0, b_kx = 0, b_ci = 0;  // im2col column decode (fixed per thread)
  if constexpr (BMODE == B_KC_PLAIN) {
#pragma unroll
    for (int r = 0; r < BNR; ++r) {
      const int n = tile_n0 + ld_row + 32 * r;
      b_ok[r] = n < g.N;
      b_base[r] = (long long)n * g.ldb;
    }
  } else if constexpr (BMODE == B_KS_IM2COL) {
    const int n = tile_n0 + b_ks_c;
    const int tap = n / g.Cin;
    b_ci = n - tap * g.Cin;
    b_ky = tap / g.kw;
    b_kx = tap - b_ky * g.kw;
  }

  auto load_A = [&](int kc, float4 (&regs)[ANR], unsigned& mask, float* dst) {
    mask = 0u;
    if constexpr (!A_KS) {
      const int k = kc * BK + kq;
      const int kleft = g.K - k;
      long long off = k;
      int dy = 0, dx = 0;
      if constexpr (AM == A_KC_CONV || AM == A_KC_TCONV) {
        int tap, c;
        if constexpr (CHK) {  // wave-uniform: scalar division, result lives in SGPRs
          const int k0 = kc * BK;
          tap = k0 / g.Cin;
          c = k0 - tap * g.Cin + kq;
        } else {
          tap = k / g.Cin;
          c = k - tap * g.Cin;
        }
        if constexpr (AM == A_KC_CONV) {
          dy = tap / g.kw;
          dx = tap - dy * g.kw;
        } else {
          dy = -(tap >> 1);
          dx = -(tap & 1);
        }
        off = ((long long)dy * g.W + dx) * g.Cin + c;
      }
#pragma unroll
      for (int r = 0; r < ANR; ++r) {
        bool ok = a_ok[r] & (kleft > 0);
        if constexpr (AM != A_KC_PLAIN) {
          ok = ok & ((unsigned)(a_iy0[r] + dy) < (unsigned)g.H) & ((unsigned)(a_ix0[r] + dx) < (unsigned)g.W);
        }
        if constexpr (DMA) {
          dma16(g.A + a_base[r] + off, ok, dst + (wid * 8 + 32 * r) * BK);
        } else {
          regs[r] = ldsel<FAST>(g.A + a_base[r] + off, ok, kleft, g.a_vec, g.A);
          mask |= (ok ? 1u : 0u) << r;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < ANR; ++p) {
        const int kk = kc * BK + a_ks_row + p * A_RPP;
        const int m = tile_m0 + a_ks_c;
        const bool ok = (kk < g.K) & (m < g.M);
        if constexpr (DMA) {
          dma16(g.A + (long long)kk * g.lda + m, ok, dst + (wid * 64 + p * 256) * 4);
        } else {
          regs[p] = ldsel<FAST>(g.A + (long long)kk * g.lda + m, ok, g.M - m, g.a_vec, g.A);
          mask |= (ok ? 1u : 0u) << p;
        }
      }
    }
  };

  auto load_B = [&](int kc, float4 (&regs)[BNR], unsigned& mask, float* dst) {
    mask = 0u;
    if constexpr (B_FR) {
      return;  // B never passes through the staging registers / LDS in this mode
    } else if constexpr (BMODE == B_KC_PLAIN) {
      const int k = kc * BK + kq;
      const int kleft = g.K - k;
#pragma unroll
      for (int r = 0; r < BNR; ++r) {
        const bool ok = b_ok[r] & (kleft > 0);
        if constexpr (DMA) {
          dma16(g.B + b_base[r] + k, ok, dst + (wid * 8 + 32 * r) * BK);
        } else {
          regs[r] = ldsel<FAST>(g.B + b_base[r] + k, ok, kleft, g.b_vec, g.B);
          mask |= (ok ? 1u : 0u) << r;
        }
      }
    } else {
      const int n = tile_n0 + b_ks_c;
#pragma unroll
      for (int p = 0; p < BNR; ++p) {
        const int kk = kc * BK + b_ks_row + p * B_RPP;
        bool ok = (kk < g.K) & (n < g.N);
        const float* src = g.B;
        if constexpr (BMODE == B_KS_PLAIN) {
          src += (long long)kk * g.ldb + n;
        } else if constexpr (BMODE == B_KS_TAPFLIP || BMODE == B_KS_TAPCLS) {
          int tapi, cr;
          if constexpr (CHK) {
            const int k0 = kc * BK;
            tapi = k0 / g.b_Cred;
            cr = kk - tapi * g.b_Cred;
          } else {
            const int kks = ok ? kk : 0;
            tapi = kks / g.b_Cred;
            cr = kks - tapi * g.b_Cred;
          }
          int tap;
          if constexpr (BMODE == B_KS_TAPFLIP) {
            tap = g.b_taps - 1 - tapi;
          } else {
            const int ky = (1 - py) + 2 * (tapi >> 1), kx = (1 - px) + 2 * (tapi & 1);
            tap = ky * 4 + kx;
          }
          src += ((long long)cr * g.b_taps + tap) * g.b_Cn + n;
        } else {  // B_KS_IM2COL: reduction index = output pixel, column = (tap, ci)
          const int kks = ok ? kk : 0;
          const int hw = g.Ho * g.Wo;
          int img, pp, oy, ox;
          if (g.hw_shift >= 0) {  // wave-uniform branch: shifts instead of two integer divisions per load
            img = kks >> g.hw_shift;
            pp = kks & (hw - 1);
            oy = pp >> g.wo_shift;
            ox = pp & (g.Wo - 1);
          } else {
            img = kks / hw;
            pp = kks - img * hw;
            oy = pp / g.Wo;
            ox = pp - oy * g.Wo;
          }
          const int iy = oy * g.stride - g.pad + b_ky, ix = ox * g.stride - g.pad + b_kx;
          ok = ok & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
          src += (((long long)img * g.H + iy) * g.W + ix) * g.Cin + b_ci;
        }
        if constexpr (DMA) {
          dma16(src, ok, dst + (wid * 64 + p * 256) * 4);
        } else {
          regs[p] = ldsel<FAST>(src, ok, g.N - n, g.b_vec, g.B);
          mask |= (ok ? 1u : 0u) << p;
        }
      }
    }
  };

  auto store_A = [&](int buf, const float4 (&regs)[ANR], unsigned mask) {
    float* S = As + buf * A_TILE;
#pragma unroll
    for (int r = 0; r < ANR; ++r) {
      const float4 v = (!FAST || ((mask >> r) & 1u)) ? regs[r] : zero4();
      if constexpr (!A_KS)
        *reinterpret_cast<float4*>(S + (ld_row + 32 * r) * LDK + kq) = v;
      else
        *reinterpret_cast<float4*>(S + (a_ks_row + r * A_RPP) * BM + a_ks_c) = v;
    }
  };
  auto store_B = [&](int buf, const float4 (&regs)[BNR], unsigned mask) {
    if constexpr (B_FR) return;
    float* S = Bs + buf * B_TILE;
#pragma unroll
    for (int r = 0; r < BNR; ++r) {
      const float4 v = (!FAST || ((mask >> r) & 1u)) ? regs[r] : zero4();
      if constexpr (!B_KS)
        *reinterpret_cast<float4*>(S + (ld_row + 32 * r) * LDK + kq) = v;
      else
        *reinterpret_cast<float4*>(S + (b_ks_row + r * B_RPP) * BN + b_ks_c) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Software pipeline: two register stages + two LDS buffers, prefetch distance 2.
  //   iteration k: issue loads(k+2) -> stage[k&1]; MFMAs on buf[k&1]; store stage[(k+1)&1] (= chunk k+1, issued one
  //   iteration ago) -> buf[(k+1)&1]; lgkmcnt(0) + raw s_barrier (VMEM stays in flight across the barrier).
  // A global load therefore has ~2 MFMA phases (~3.5 us) to land. The prefetch index is clamped instead of
  // branching so every iteration is one basic block; loads are pinned into the first quarter of the MFMAs.
  float4 ra0[ANR], rb0[BNR], ra1[ANR], rb1[BNR];
  unsigned ma0 = 0u, mb0 = 0u, ma1 = 0u, mb1 = 0u;
  const int kc_last = kc_end - 1;

  // fragment (operand) registers are double-buffered: quarter q+1's LDS reads are issued before quarter q's MFMAs
  auto frag_load = [&](int q, const float* Sa, const float* Sb, float4 (&af)[TM], float4 (&bf)[TN]) {
    const int k4 = 4 * (2 * q + lh);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = wr * WM + tm * 32 + li;
      if constexpr (!A_KS) {
        if constexpr (DMA) af[tm] = *reinterpret_cast<const float4*>(Sa + row * BK + (((2 * q + lh) ^ ((row >> 1) & 7)) << 2));
        else af[tm] = *reinterpret_cast<const float4*>(Sa + row * LDK + k4);
      } else {
        af[tm].x = Sa[(k4 + 0) * BM + row];
        af[tm].y = Sa[(k4 + 1) * BM + row];
        af[tm].z = Sa[(k4 + 2) * BM + row];
        af[tm].w = Sa[(k4 + 3) * BM + row];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if constexpr (B_FR) continue;
      const int row = wc * WN + tn * 32 + li;
      if constexpr (!B_KS) {
        if constexpr (DMA) bf[tn] = *reinterpret_cast<const float4*>(Sb + row * BK + (((2 * q + lh) ^ ((row >> 1) & 7)) << 2));
        else bf[tn] = *reinterpret_cast<const float4*>(Sb + row * LDK + k4);
      } else {
        bf[tn].x = Sb[(k4 + 0) * BN + row];
        bf[tn].y = Sb[(k4 + 1) * BN + row];
        bf[tn].z = Sb[(k4 + 2) * BN + row];
        bf[tn].w = Sb[(k4 + 3) * BN + row];
      }
    }
  };
  auto frag_mfma = [&](const float4 (&af)[TM], const float4 (&bf)[TN]) {
#if FL_MFMA_ROT
#define FL_MFMA_STEP(c)                                                                                          \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)          \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm].c, bf[tn].c, acc[tm][tn], 0, 0, 0);
    FL_MFMA_STEP(x) FL_MFMA_STEP(y) FL_MFMA_STEP(z) FL_MFMA_STEP(w)
#undef FL_MFMA_STEP
#else
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm].x, bf[tn].x, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm].y, bf[tn].y, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm].z, bf[tn].z, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm].w, bf[tn].w, acc[tm][tn], 0, 0, 0);
      }
#endif
  };
  float4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  static_assert(!(DMA && SPLIT), "the split path is implemented for the register-staged pipeline only");
  float4 fa2[TM], fb2[TN], fa3[TM], fb3[TN];
  // half a chunk (two quarters = this lane's 8 k-slots of a 16-wide bf16 MFMA step): split, then 6 or 8 piece products
  // per tile, smallest terms first
  auto frag_mfma2 = [&](const float4 (&a0)[TM], const float4 (&a1)[TM], const float4 (&b0)[TN], const float4 (&b1)[TN]) {
    Split3 sa[TM], sb[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) sa[tm] = split3(a0[tm], a1[tm]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) sb[tn] = split3(b0[tn], b1[tn]);
#define FL_PIECE(pa, pb)                                                                                        \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)          \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, sa[tm].pa),         \
                                                              __builtin_bit_cast(bf16x8_t, sb[tn].pb), acc[tm][tn], 0, 0, 0);
    if constexpr (SPLIT >= 8) {
      FL_PIECE(m, l) FL_PIECE(l, m)
    }
    FL_PIECE(l, h) FL_PIECE(h, l) FL_PIECE(m, m) FL_PIECE(m, h) FL_PIECE(h, m) FL_PIECE(h, h)
#undef FL_PIECE
  };
  // On entry (fa0, fb0), (fa1, fb1) hold quarters 0, 1 of chunk kc; on exit those of chunk kc+1.
  auto iteration_split = [&](int kc, int buf, float4 (&lda)[ANR], float4 (&ldb)[BNR], unsigned& lma, unsigned& lmb,
                       const float4 (&sta)[ANR], const float4 (&stb)[BNR], unsigned sma, unsigned smb) {
    const int k2 = kc + 2 < kc_end ? kc + 2 : kc_last;
    load_A(k2, lda, lma, nullptr);
    load_B(k2, ldb, lmb, nullptr);
    const float* Sa = As + buf * A_TILE;
    const float* Sb = Bs + buf * B_TILE;
    frag_load(2, Sa, Sb, fa2, fb2);
    frag_load(3, Sa, Sb, fa3, fb3);
    frag_mfma2(fa0, fa1, fb0, fb1);
    __builtin_amdgcn_sched_barrier(0);
    store_A(buf ^ 1, sta, sma);
    store_B(buf ^ 1, stb, smb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frag_load(0, As + (buf ^ 1) * A_TILE, Bs + (buf ^ 1) * B_TILE, fa0, fb0);
    frag_load(1, As + (buf ^ 1) * A_TILE, Bs + (buf ^ 1) * B_TILE, fa1, fb1);
    frag_mfma2(fa2, fa3, fb2, fb3);  // covers the fragment reads of the next chunk
  };

  // one pipeline iteration; LD* = stage receiving chunk kc+2, ST* = stage holding chunk kc+1.
  // On entry (fa0, fb0) hold quarter 0 of chunk kc; on exit they hold quarter 0 of chunk kc+1.
  auto iteration_fp32 = [&](int kc, int buf, float4 (&lda)[ANR], float4 (&ldb)[BNR], unsigned& lma, unsigned& lmb,
                       const float4 (&sta)[ANR], const float4 (&stb)[BNR], unsigned sma, unsigned smb) {
    const int k2 = kc + 2 < kc_end ? kc + 2 : kc_last;  // clamped: tail iterations re-load the last chunk (unused)
#if !(FL_ABLATE & 1)
    load_A(k2, lda, lma, nullptr);
    load_B(k2, ldb, lmb, nullptr);
#endif
    const float* Sa = As + buf * A_TILE;
    const float* Sb = Bs + buf * B_TILE;
    frag_load(1, Sa, Sb, fa1, fb1);
    frag_mfma(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);  // the prefetch is issued before the remaining 3/4 of the MFMAs
    frag_load(2, Sa, Sb, fa0, fb0);
    frag_mfma(fa1, fb1);
    frag_load(3, Sa, Sb, fa1, fb1);
#if !(FL_ABLATE & 2)
    store_A(buf ^ 1, sta, sma);
    store_B(buf ^ 1, stb, smb);
#endif
    frag_mfma(fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS traffic is done; VMEM is NOT drained
#if !(FL_ABLATE & 4)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    // quarter 3's MFMAs run after the barrier and cover the latency of the next chunk's first fragment reads
    frag_load(0, As + (buf ^ 1) * A_TILE, Bs + (buf ^ 1) * B_TILE, fa0, fb0);
    frag_mfma(fa1, fb1);
  };
  // ---- B_FRAGS: pre-split B fragments straight from global memory, two plane sets alternating by half chunk ----
  u32x4_t bpA[TN][3], bpB[TN][3];
  const int nt32 = g.N >> 5;
  const int g16_last = 2 * kc_end - 1;
  auto load_bp = [&](int g16, u32x4_t (&bp)[TN][3]) {
#if FL_ABLATE & 8
    if (g16 > 2 * kc_begin + 1) return;  // timing-only: B pieces are fetched once
#endif
    g16 = g16 < g16_last ? g16 : g16_last;  // clamped: the tail re-reads the last group (never consumed)
    const u32x4_t* base = reinterpret_cast<const u32x4_t*>(g.B) +
                          ((size_t)g16 * nt32 + ((tile_n0 + wc * WN) >> 5)) * (3 * 64) + lane;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[tn][p] = base[(tn * 3 + p) * 64];
  };
  // Software-pipelined inside the wave: the 24 MFMAs of one half chunk (on pieces split earlier) and the VALU split of
  // the NEXT half's A fragments sit in one scheduling region and are interleaved 1 MFMA : 8 VALU, so the matrix pipe
  // works while the vector pipe prepares its next operands (a wave issues in order, but an MFMA executes for 32
  // cycles after its 4-cycle issue). sa01 / sa23 are the split A pieces of quarters (0,1) / (2,3).
  Split3 sa01[TM], sa23[TM];
  auto mfma_and_split = [&](const Split3 (&sa)[TM], const u32x4_t (&bp)[TN][3], Split3 (&sn)[TM], const float4 (&n0)[TM],
                            const float4 (&n1)[TM]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) sn[tm] = split3(n0[tm], n1[tm]);
#define FL_PIECE(pa, pi)                                                                                        \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)          \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, sa[tm].pa),         \
                                                              __builtin_bit_cast(bf16x8_t, bp[tn][pi]), acc[tm][tn], 0, 0, 0);
#if FL_ABLATE & 32
    FL_PIECE(h, 0)
#else
    FL_PIECE(l, 0) FL_PIECE(h, 2) FL_PIECE(m, 1) FL_PIECE(m, 0) FL_PIECE(h, 1) FL_PIECE(h, 0)
#endif
#undef FL_PIECE
    // pattern for the scheduler: 4 MFMAs up front (cover the LDS latency of the fragments being split), then
    // 1 MFMA : 8 VALU until the MFMAs run out
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
    for (int i = 0; i < 6 * TM * TN - 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  // On entry sa01 holds the split A pieces of quarters 0, 1 of chunk kc and bpA the B pieces of its first half.
  auto iteration_frags = [&](int kc, int buf, float4 (&lda)[ANR], unsigned& lma, const float4 (&sta)[ANR], unsigned sma) {
    const int k2 = kc + 2 < kc_end ? kc + 2 : kc_last;
    load_A(k2, lda, lma, nullptr);
    load_bp(2 * kc + 1, bpB);
    const float* Sa = As + buf * A_TILE;
    frag_load(2, Sa, Sa, fa2, fb2);
    frag_load(3, Sa, Sa, fa3, fb3);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(sa01, bpA, sa23, fa2, fa3);
    __builtin_amdgcn_sched_barrier(0);
#if !(FL_ABLATE & 2)
    store_A(buf ^ 1, sta, sma);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(FL_ABLATE & 4)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    load_bp(2 * kc + 2, bpA);
    frag_load(0, As + (buf ^ 1) * A_TILE, As, fa0, fb0);
    frag_load(1, As + (buf ^ 1) * A_TILE, As, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(sa23, bpB, sa01, fa0, fa1);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto iteration = [&](int kc, int buf, float4 (&lda)[ANR], float4 (&ldb)[BNR], unsigned& lma, unsigned& lmb,
                       const float4 (&sta)[ANR], const float4 (&stb)[BNR], unsigned sma, unsigned smb) {
    if constexpr (B_FR) iteration_frags(kc, buf, lda, lma, sta, sma);
    else if constexpr (SPLIT) iteration_split(kc, buf, lda, ldb, lma, lmb, sta, stb, sma, smb);
    else iteration_fp32(kc, buf, lda, ldb, lma, lmb, sta, stb, sma, smb);
  };

  // Two waves share each SIMD (2 blocks/CU) and run the same program: left alone they fall into lockstep and do
  // their non-MFMA phases together, leaving the matrix pipe idle. A static priority for one of the two
  // (hardware wave slot parity, wave-uniform scalar read) makes one stream the "owner" of the pipe and the other
  // the gap filler (MI355X_MICROARCH §Two waves per SIMD, item 4). Speed only, never correctness.
  if (g.prio_mode) {
    const unsigned hw_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 4);  // HW_REG_HW_ID[3:0] = wave slot
    if (__builtin_amdgcn_readfirstlane(hw_id) & 1u) __builtin_amdgcn_s_setprio(1);
  }
  if constexpr (!DMA) {
    if (kc_begin < kc_end) {
      load_A(kc_begin, ra0, ma0, nullptr);
      load_B(kc_begin, rb0, mb0, nullptr);
      store_A(0, ra0, ma0);
      store_B(0, rb0, mb0);
      const int k1 = kc_begin + 1 < kc_end ? kc_begin + 1 : kc_last;
      load_A(k1, ra1, ma1, nullptr);
      load_B(k1, rb1, mb1, nullptr);
    }
    __syncthreads();
    frag_load(0, As, Bs, fa0, fb0);
    if constexpr (SPLIT) frag_load(1, As, Bs, fa1, fb1);
    if constexpr (B_FR) {
      load_bp(2 * kc_begin, bpA);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) sa01[tm] = split3(fa0[tm], fa1[tm]);
    }

    int kc = kc_begin;
    for (; kc + 1 < kc_end; kc += 2) {
      iteration(kc, 0, ra0, rb0, ma0, mb0, ra1, rb1, ma1, mb1);
      iteration(kc + 1, 1, ra1, rb1, ma1, mb1, ra0, rb0, ma0, mb0);
    }
    if (kc < kc_end) iteration(kc, 0, ra0, rb0, ma0, mb0, ra1, rb1, ma1, mb1);
  } else {
    // ---- direct-to-LDS pipeline: 2 LDS buffers, DMA(k+1) in flight under the MFMAs of chunk k ----
    // hipcc waits vmcnt(0) before any ds_read that follows an LDS-DMA in program order (it cannot prove that the
    // two buffers do not alias), so within an iteration every LDS read precedes the DMA issue: the whole chunk's
    // fragments (4 quarters, 64 VGPRs) are read right after the barrier, overlapped with the previous chunk's
    // last 16 MFMAs, then the DMA is issued, then the MFMAs run from registers.
    auto issue = [&](int kc_, int buf_) {
      unsigned dm;
      load_A(kc_, ra0, dm, As + buf_ * A_TILE);
      load_B(kc_, rb0, dm, Bs + buf_ * B_TILE);
    };
    float4 qa0[TM], qb0[TN], qa1[TM], qb1[TN], qa2[TM], qb2[TN], qa3[TM], qb3[TN];
    if (kc_begin < kc_end) issue(kc_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frag_load(0, As, Bs, qa0, qb0);
    frag_load(1, As, Bs, qa1, qb1);
    frag_load(2, As, Bs, qa2, qb2);
    frag_load(3, As, Bs, qa3, qb3);
    int buf = 0;
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      const int k1 = kc + 1 < kc_end ? kc + 1 : kc_last;  // clamped: the last iteration re-stages the last chunk (unused)
      issue(k1, buf ^ 1);  // buf^1: every read of it was retired before the previous barrier
      frag_mfma(qa0, qb0);  // same scheduling region: address arithmetic + DMA issue interleave with 16 MFMAs
      __builtin_amdgcn_sched_barrier(0);  // ... but the DMA is issued before the remaining MFMAs
      frag_mfma(qa1, qb1);
      frag_mfma(qa2, qb2);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa1[i] = qa3[i];
#pragma unroll
      for (int i = 0; i < TN; ++i) fb1[i] = qb3[i];
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
      __builtin_amdgcn_s_barrier();                      // ... and everyone else's
      asm volatile("" ::: "memory");
      const float* Sa = As + (buf ^ 1) * A_TILE;
      const float* Sb = Bs + (buf ^ 1) * B_TILE;
      frag_load(0, Sa, Sb, qa0, qb0);
      frag_load(1, Sa, Sb, qa1, qb1);
      frag_load(2, Sa, Sb, qa2, qb2);
      frag_load(3, Sa, Sb, qa3, qb3);
      frag_mfma(fa1, fb1);  // quarter 3 of chunk k covers the fragment reads of chunk k+1
      __builtin_amdgcn_sched_barrier(0);
      buf ^= 1;
    }
  }

  // ------------------------------------------------------------------ epilogue
  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float st1[TN], st2[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) st1[tn] = st2[tn] = 0.f;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = tile_m0 + wr * WM + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (m >= g.M) continue;
      if (g.splitk > 1) {
        float* wrow = g.ws + ((long long)z * g.M + m) * g.N;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = tile_n0 + wc * WN + tn * 32 + li;
          if (col < g.N) wrow[col] = acc[tm][tn][e];
        }
        continue;
      }
      long long row_off, col_stride = 1, rrow_off;
      if (g.out_mode == OUT_ROWMAJOR) {
        row_off = (long long)m * g.ldc;
        rrow_off = (long long)m * g.ldr;
      } else if (g.out_mode == OUT_NCHW) {
        const int hw = g.Ho * g.Wo;
        const int img = m / hw, p = m - img * hw;
        row_off = (long long)img * g.N * hw + p;
        col_stride = hw;
        rrow_off = row_off;
      } else {  // OUT_TCONV: row (img,yh,xh) of class (py,px) -> pixel (2yh+py, 2xh+px)
        const int hw = g.Ho * g.Wo;
        const int img = m / hw, p = m - img * hw;
        const int yh = p / g.Wo, xh = p - yh * g.Wo;
        row_off = (((long long)img * (2 * g.Ho) + (2 * yh + py)) * (2 * g.Wo) + (2 * xh + px)) * g.ldc;
        rrow_off = row_off;
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = tile_n0 + wc * WN + tn * 32 + li;
        if (col >= g.N) continue;
        const float* rptr = g.R ? g.R + rrow_off + (long long)col * col_stride : nullptr;
        const float v = epilogue_value(g, acc[tm][tn][e], col, rptr);
        g.C[row_off + (long long)col * col_stride] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  // fused LayerNorm2d statistics (v2:152-153): this wave's WM rows belong to one image (host guarantees HW % WM == 0),
  // so its column sums are one slab of that image's (sum y, sum y^2); the two lane halves hold disjoint rows.
  if (g.stats && g.splitk <= 1) {
    const int grp = (tile_m0 + wr * WM) / WM;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const float a = st1[tn] + __shfl_xor(st1[tn], 32, 64);
      const float b = st2[tn] + __shfl_xor(st2[tn], 32, 64);
      const int col = tile_n0 + wc * WN + tn * 32 + li;
      if (lh == 0 && col < g.N && tile_m0 + wr * WM < g.M) {
        g.stats[((long long)grp * 2) * g.N + col] = a;
        g.stats[((long long)grp * 2 + 1) * g.N + col] = b;
      }
    }
  }
}

// fp32 [N][K] -> B_FRAGS layout (see the enum): one thread per (16-wide k group, 32-column tile, lane)
__global__ void split_pack_kernel(const float* __restrict__ w, u32x4_t* __restrict__ out, int N, int K) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nt = N >> 5;
  if (i >= (long long)(K >> 4) * nt * 64) return;
  const int lane = (int)(i & 63), lr = lane & 31, lh = lane >> 5;
  const long long gt = i >> 6;
  const int t = (int)(gt % nt), g16 = (int)(gt / nt);
  const float* row = w + (long long)(32 * t + lr) * K + 16 * g16 + 4 * lh;  // k slots: 4*lh + j and 8 + 4*lh + j
  const Split3 s = split3(*reinterpret_cast<const float4*>(row), *reinterpret_cast<const float4*>(row + 8));
  u32x4_t* o = out + (gt * 3) * 64 + lane;
  o[0] = s.h;
  o[64] = s.m;
  o[128] = s.l;
}

// split-K combine + epilogue (row-major outputs only)
__global__ void splitk_reduce_kernel(const GArgs g) {
  const long long total = (long long)g.M * g.N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / g.N), col = (int)(i - (long long)m * g.N);
    float v = 0.f;
    for (int zz = 0; zz < g.splitk; ++zz) v += g.ws[(long long)zz * total + i];
    const float* rptr = g.R ? g.R + (long long)m * g.ldr + col : nullptr;
    g.C[(long long)m * g.ldc + col] = epilogue_value(g, v, col, rptr);
  }
}

// ------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int AM, int BMODE, int FAST, int SPLIT>
int launch_cfg3(GArgs g, int classes, hipStream_t stream) {
  constexpr size_t lds = 2u * (BM + (BMODE == B_FRAGS ? 0 : BN)) * ((FAST >= 1 && FL_USE_DMA) ? BK : LDK) * sizeof(float);
  static bool attr_set = false;
  auto kern = gemm_kernel<BM, BN, WGM, WGN, AM, BMODE, FAST, SPLIT>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles_m = fl::cdiv(g.M, BM);
  g.tiles_n = fl::cdiv(g.N, BN);
  {
    const int hw = g.Ho * g.Wo;
    g.hw_shift = (hw > 0 && (hw & (hw - 1)) == 0 && g.Wo > 0 && (g.Wo & (g.Wo - 1)) == 0) ? __builtin_ctz(hw) : -1;
    g.wo_shift = g.hw_shift >= 0 ? __builtin_ctz(g.Wo) : -1;
  }
  {
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("FL_GEMM_PRIO"); prio = e ? atoi(e) : 1; }
    g.prio_mode = prio;
  }
  dim3 grid((unsigned)(tiles_m * g.tiles_n), (unsigned)classes, (unsigned)(g.splitk > 1 ? g.splitk : 1));
  static int lds_pad = -1;  // experiment knob: extra dynamic LDS to lower the number of resident blocks per CU
  if (lds_pad < 0) { const char* e = getenv("FL_GEMM_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
  if (lds_pad > 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + lds_pad);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds + lds_pad, stream, g);
    return fl::check_launch("gemm_kernel");
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g);
  return fl::check_launch("gemm_kernel");
}

int g_gemm_math = -1;  // -1: read FL_GEMM_MATH on first use; 0: fp32 MFMA; 1: exact bf16 split
int gemm_math() {
  if (g_gemm_math < 0) {
    const char* e = getenv("FL_GEMM_MATH");
    g_gemm_math = (e && (!strcmp(e, "fp32") || !strcmp(e, "0"))) ? 0 : 1;
  }
  return g_gemm_math;
}

// the guarded slow path (FAST == 0: unaligned operands) always uses the fp32 MFMA; the vectorised paths pick per mode
template <int BM, int BN, int WGM, int WGN, int AM, int BMODE, int FAST>
int launch_cfg2(GArgs g, int classes, hipStream_t stream) {
  if constexpr (FAST >= 1 && !FL_USE_DMA) {
    if (gemm_math()) return launch_cfg3<BM, BN, WGM, WGN, AM, BMODE, FAST, FL_SPLIT_TERMS>(g, classes, stream);
  }
  return launch_cfg3<BM, BN, WGM, WGN, AM, BMODE, FAST, 0>(g, classes, stream);
}

template <int BM, int BN, int WGM, int WGN, int AM, int BMODE>
int launch_cfg(GArgs g, int classes, hipStream_t stream) {
  bool fast = g.a_vec && g.b_vec;
  if (AM != A_KS_PLAIN || BMODE == B_KC_PLAIN) fast = fast && (g.K % 4 == 0);  // a K-contiguous operand exists
  if (AM == A_KS_PLAIN) fast = fast && (g.M % 4 == 0);
  if (BMODE != B_KC_PLAIN) fast = fast && (g.N % 4 == 0);
  if constexpr (AM == A_KC_CONV || AM == A_KC_TCONV) {
    const bool chunked = (g.Cin % BK == 0) && (BMODE == B_KC_PLAIN || g.b_Cred % BK == 0);
    if (fast && chunked) return launch_cfg2<BM, BN, WGM, WGN, AM, BMODE, 2>(g, classes, stream);
  }
  if (fast) return launch_cfg2<BM, BN, WGM, WGN, AM, BMODE, 1>(g, classes, stream);
  return launch_cfg2<BM, BN, WGM, WGN, AM, BMODE, 0>(g, classes, stream);
}

// tile-shape + split-K policy. `classes` = 4 for transposed conv, else 1.
// split-K is used when a GEMM's grid has fewer tiles than this (default 160: fill the 256 CUs of an otherwise idle
// chip). A caller whose GEMMs share the chip with another stream (the denoiser next to the frozen encoder) lowers it:
// there the extra reduce launch costs more than the idle CUs it would fill (fl_set_gemm_splitk_tiles).
int g_splitk_tiles = 160;
// 3x3 halo conv variant: 0 = 4 waves x (64 x 64) tiles, 2 blocks per CU (fastest alone); 1 = 8 waves x (32 x 64) tiles,
// 1 block per CU with VGPRs / LDS left for another stream's waves (fl_set_conv_halo_mode)
int g_conv_halo_mode = 0;

template <int AM, int BMODE>
int launch_gemm(GArgs g, int classes, void* ws, size_t ws_bytes, hipStream_t stream, int* wm_out = nullptr) {
  if (g.M <= 0 || g.N <= 0) return FL_OK;
  const int nk = fl::cdiv(g.K, BK);
  // pick tile
  int bm = 128, bn = 128;
  if (g.N <= 32) bn = 32;
  else if (g.N <= 64 || (g.N % 128 != 0 && g.N % 128 <= 64 && g.N < 512)) bn = 64;
  long long tiles = (long long)fl::cdiv(g.M, bm) * fl::cdiv(g.N, bn) * classes;
  if (bn == 128 && tiles < 192) {
    // few tiles: with a long reduction keep the MFMA-dense 128x128 tile and fill the chip by split-K (wgrad,
    // the 32768-wide fc layers); with a short one use 64x64 tiles so that more CUs get a tile at all.
    const int want = (int)((320 + tiles - 1) / tiles);
    const bool long_k = g.out_mode == OUT_ROWMAJOR && classes == 1 && ws != nullptr && nk / want >= 16 &&
                        (size_t)want * g.M * g.N * sizeof(float) <= ws_bytes;
    if (!long_k) {
      bm = 64;
      bn = 64;
      tiles = (long long)fl::cdiv(g.M, bm) * fl::cdiv(g.N, bn) * classes;
    }
  }
  // split-K when the grid still cannot fill 256 CUs (row-major epilogue only)
  g.splitk = 1;
  g.chunks_per_split = nk;
  g.ws = reinterpret_cast<float*>(ws);
  if (g.out_mode == OUT_ROWMAJOR && classes == 1 && tiles < g_splitk_tiles && nk >= 8 && ws != nullptr) {
    int want = (int)((320 + tiles - 1) / tiles);
    int maxs = nk / 4;
    if (want > maxs) want = maxs;
    const size_t per = (size_t)g.M * g.N * sizeof(float);
    if (per > 0 && (size_t)want * per > ws_bytes) want = (int)(ws_bytes / per);
    if (want >= 2) {
      g.chunks_per_split = fl::cdiv(nk, want);
      g.splitk = fl::cdiv(nk, g.chunks_per_split);
    }
  }
  if (wm_out) *wm_out = (bm == 128 && bn == 32) ? 32 : bm / 2;  // rows per wave tile (WM) of the chosen config
  if (g.stats) g.splitk = 1;                                     // fused statistics need the final values in the epilogue
  if (g.splitk <= 1) g.chunks_per_split = nk;
  int st;
  if (bm == 128 && bn == 128) st = launch_cfg<128, 128, 2, 2, AM, BMODE>(g, classes, stream);
  else if (bm == 128 && bn == 64) st = launch_cfg<128, 64, 2, 2, AM, BMODE>(g, classes, stream);
  else if (bm == 128 && bn == 32) st = launch_cfg<128, 32, 4, 1, AM, BMODE>(g, classes, stream);
  else st = launch_cfg<64, 64, 2, 2, AM, BMODE>(g, classes, stream);
  if (st != FL_OK) return st;
  if (g.splitk > 1) {
    const long long total = (long long)g.M * g.N;
    int blocks = fl::cdiv(total, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
    return fl::check_launch("splitk_reduce_kernel");
  }
  return FL_OK;
}

inline int is_vec(const void* p, long long ld) { return (((uintptr_t)p) % 16 == 0) && (ld % 4 == 0); }

GArgs base_args() {
  GArgs g;
  memset(&g, 0, sizeof(g));
  g.bias_scale = 1.f;
  g.r_scale = 1.f;
  g.kw = 1;
  g.stride = 1;
  g.Cin = 1;
  g.Ho = g.Wo = 1;
  g.b_Cred = 1;
  return g;
}

// (O, I, kh, kw) <-> [O][kh][kw][Ipad]
__global__ void pack_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int O, int I, int taps, int Ipad) {
  const long long total = (long long)O * taps * Ipad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ipad);
    const long long r = i / Ipad;
    const int tap = (int)(r % taps);
    const int o = (int)(r / taps);
    wp[i] = ci < I ? w[((long long)o * I + ci) * taps + tap] : 0.f;
  }
}
__global__ void unpack_w_kernel(const float* __restrict__ wp, float* __restrict__ w, int O, int I, int taps, int Ipad) {
  const long long total = (long long)O * I * taps;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long long r = i / taps;
    const int ci = (int)(r % I);
    const int o = (int)(r / I);
    w[i] = wp[((long long)o * taps + tap) * Ipad + ci];
  }
}

// per-(image, channel) LayerNorm2d scale/shift from the conv epilogue's per-wave slabs (same math as norm.hip)
__global__ void conv_inorm_finalize(const float* partial, const float* w, const float* bias, float eps, float* scale,
                                    float* shift, float* mean, float* rstd, int B, int HW, int C, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < S; ++s) {
    const float* p = partial + ((long long)(b * S + s) * 2) * C + c;
    s1 += (double)p[0];
    s2 += (double)p[C];
  }
  const double m = s1 / HW;
  double var = s2 / HW - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = r * (w ? w[c] : 1.f);
  scale[i] = sc;
  shift[i] = (bias ? bias[c] : 0.f) - (float)m * sc;
  if (mean) mean[i] = (float)m;
  if (rstd) rstd[i] = r;
}

}  // namespace

namespace fl {
// per-(image, channel) LayerNorm2d scale/shift from a conv epilogue's statistics slabs (shared with conv_h2.hip)
int launch_conv_inorm_finalize(const float* partial, const float* ln_w, const float* ln_b, float eps, float* scale, float* shift,
                               float* mean, float* rstd, int B, int HW, int C, int slabs, hipStream_t s) {
  hipLaunchKernelGGL(conv_inorm_finalize, dim3(fl::cdiv((long long)B * C, 256)), dim3(256), 0, s, partial, ln_w, ln_b, eps, scale,
                     shift, mean, rstd, B, HW, C, slabs);
  return fl::check_launch("conv_inorm_finalize");
}
}  // namespace fl

// =====================================================================================
extern "C" {

size_t fl_split_pack_weight_bytes(int N, int K) { return (size_t)N * K * 6; }

// w: [N][K] fp32 (K-contiguous: a packed conv weight or an nn.Linear weight), N % 32 == 0, K % 16 == 0
int fl_split_pack_weight(const float* w, void* out, int N, int K, fl_stream_t stream) {
  FL_REQUIRE(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "fl_split_pack_weight: N=%d (mult. of 32), K=%d (mult. of 16)", N, K);
  const long long total = (long long)(K / 16) * (N / 32) * 64;
  hipLaunchKernelGGL(split_pack_kernel, dim3(fl::cdiv(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     reinterpret_cast<u32x4_t*>(out), N, K);
  return fl::check_launch("split_pack_kernel");
}

// conv forward with a pre-split weight (B_FRAGS): 128x128 tiles, Cin % 32 == 0, Cout % 128 == 0; optional fused
// LayerNorm2d statistics like fl_conv2d_fwd_instnorm (ln_w/ln_b/scale/shift may all be NULL: plain conv + bias + act)
int fl_conv2d_fwd_presplit(const float* x, const void* wsplit, const float* bias, float* y, int B, int H, int W, int Cin,
                           int Cout, int kh, int kw, int stride, int pad, int act, float act_param, const float* ln_w,
                           const float* ln_b, float eps, float* scale, float* shift, float* mean, float* rstd, void* ws,
                           size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(x && wsplit && y, "fl_conv2d_fwd_presplit: null operand");
  FL_REQUIRE(Cin % 32 == 0 && Cout % 128 == 0 && is_vec(x, Cin), "fl_conv2d_fwd_presplit: needs Cin %% 32 == 0, Cout %% 128 == 0 (Cin=%d Cout=%d)", Cin, Cout);
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  const int HW = Ho * Wo;
  FL_REQUIRE(Ho > 0 && Wo > 0, "fl_conv2d_fwd_presplit: empty output");
  const bool stats = scale != nullptr;
  if (stats) {
    FL_REQUIRE(shift && act == FL_ACT_NONE && HW % 64 == 0, "fl_conv2d_fwd_presplit: fused statistics need shift, act NONE, H*W %% 64 == 0");
    FL_REQUIRE(ws && ws_bytes >= fl_conv2d_fwd_instnorm_workspace_bytes(B, Ho, Wo, Cout), "fl_conv2d_fwd_presplit: workspace too small");
  }
  GArgs g = base_args();
  g.M = B * HW; g.N = Cout; g.K = kh * kw * Cin;
  g.A = x; g.a_vec = 1;
  g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.kw = kw; g.stride = stride; g.pad = pad;
  g.B = reinterpret_cast<const float*>(wsplit); g.ldb = g.K; g.b_vec = 1;
  g.C = y; g.ldc = Cout; g.out_mode = OUT_ROWMAJOR;
  g.bias = bias; g.ldr = Cout; g.act = act; g.act_param = act_param;
  g.stats = stats ? reinterpret_cast<float*>(ws) : nullptr;
  g.splitk = 1;
  g.chunks_per_split = fl::cdiv(g.K, BK);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // 3x3 s1 p1 / 4x4 s2 p1 on the encoder's square maps: the halo-tile kernels stage each channel chunk once for all taps
  static const bool halo_on = !(getenv("FL_CONV_HALO") && getenv("FL_CONV_HALO")[0] == '0');
  const bool halo = halo_on && fl::conv3x3_halo_applicable(x, B, H, W, Cin, Cout, kh, kw, stride, pad);
  const int slab = halo ? fl::conv_halo_slab_rows(kh, g_conv_halo_mode) : 64;  // rows per statistics slab
  const int rc = halo ? fl::launch_conv_halo(x, wsplit, bias, y, g.stats, B, H, W, Cin, Cout, kh, act, act_param, slab, s)
                      : launch_cfg3<128, 128, 2, 2, A_KC_CONV, B_FRAGS, 2, 6>(g, 1, s);
  if (rc || !stats) return rc;
  hipLaunchKernelGGL(conv_inorm_finalize, dim3(fl::cdiv((long long)B * Cout, 256)), dim3(256), 0, s, g.stats, ln_w, ln_b, eps,
                     scale, shift, mean, rstd, B, HW, Cout, HW / slab);
  return fl::check_launch("conv_inorm_finalize");
}

int fl_set_gemm_math(int mode) {
  FL_REQUIRE(mode == 0 || mode == 1, "fl_set_gemm_math: mode %d (0 = fp32 MFMA, 1 = exact bf16 split)", mode);
  g_gemm_math = mode;
  return FL_OK;
}
int fl_get_gemm_math(void) { return gemm_math(); }

int fl_set_gemm_splitk_tiles(int tiles) {
  FL_REQUIRE(tiles >= 0 && tiles <= 4096, "fl_set_gemm_splitk_tiles: %d out of range", tiles);
  g_splitk_tiles = tiles;
  return FL_OK;
}
int fl_get_gemm_splitk_tiles(void) { return g_splitk_tiles; }

int fl_set_conv_halo_mode(int mode) {
  FL_REQUIRE(mode >= 0 && mode <= 2, "fl_set_conv_halo_mode: mode %d (0 = 64x64 wave tiles, 1 = 32x64 for the 3x3 convs, 2 = 32x64 for all)", mode);
  g_conv_halo_mode = mode;
  return FL_OK;
}
int fl_get_conv_halo_mode(void) { return g_conv_halo_mode; }

size_t fl_conv2d_fwd_instnorm_workspace_bytes(int B, int Ho, int Wo, int Cout) {
  return ((size_t)B * Ho * Wo / 32 + 1) * 2 * (size_t)Cout * sizeof(float);
}

// conv (+bias) with the LayerNorm2d statistics of its OUTPUT accumulated in the epilogue, then finalized to
// per-(image,channel) scale/shift/mean/rstd: replaces conv -> LayerNorm2d statistics pass (v2:162-163, 186-187, 192-193).
int fl_conv2d_fwd_instnorm(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                           int Cout, int kh, int kw, int stride, int pad, const float* ln_w, const float* ln_b,
                           float eps, float* scale, float* shift, float* mean, float* rstd, void* ws, size_t ws_bytes,
                           fl_stream_t stream) {
  FL_REQUIRE(x && wp && y && scale && shift, "fl_conv2d_fwd_instnorm: null operand");
  FL_REQUIRE(Cin % 4 == 0, "fl_conv2d_fwd_instnorm: Cin must be a multiple of 4");
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  const int HW = Ho * Wo;
  FL_REQUIRE(HW % 64 == 0, "fl_conv2d_fwd_instnorm: H*W of the output must be a multiple of 64");
  FL_REQUIRE(ws && ws_bytes >= fl_conv2d_fwd_instnorm_workspace_bytes(B, Ho, Wo, Cout), "fl_conv2d_fwd_instnorm: workspace too small");
  if (fl::conv3x3_c4_o64_applicable(B, H, W, Cin, Cout, kh, kw, stride, pad, FL_ACT_NONE)) {  // RGB stem: dedicated kernel (conv_first.hip)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* partial = reinterpret_cast<float*>(ws);
    const int rc0 = fl::launch_conv3x3_c4_o64(x, wp, bias, y, partial, B, H, W, FL_ACT_NONE, 0.f, st, gemm_math());
    if (rc0) return rc0;
    hipLaunchKernelGGL(conv_inorm_finalize, dim3(fl::cdiv((long long)B * Cout, 256)), dim3(256), 0, st, partial, ln_w, ln_b, eps,
                       scale, shift, mean, rstd, B, HW, Cout, fl::conv3x3_c4_o64_slabs(H, W));
    return fl::check_launch("conv_inorm_finalize");
  }
  GArgs g = base_args();
  g.M = B * HW; g.N = Cout; g.K = kh * kw * Cin;
  g.A = x; g.a_vec = is_vec(x, Cin);
  g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.kw = kw; g.stride = stride; g.pad = pad;
  g.B = wp; g.ldb = g.K; g.b_vec = is_vec(wp, g.K);
  g.C = y; g.ldc = Cout; g.out_mode = OUT_ROWMAJOR;
  g.bias = bias; g.ldr = Cout;
  g.stats = reinterpret_cast<float*>(ws);
  int wm = 64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = launch_gemm<A_KC_CONV, B_KC_PLAIN>(g, 1, nullptr, 0, s, &wm);
  if (rc) return rc;
  const int S = HW / wm;
  hipLaunchKernelGGL(conv_inorm_finalize, dim3(fl::cdiv((long long)B * Cout, 256)), dim3(256), 0, s, g.stats, ln_w, ln_b, eps,
                     scale, shift, mean, rstd, B, HW, Cout, S);
  return fl::check_launch("conv_inorm_finalize");
}


size_t fl_gemm_workspace_bytes(int M, int N, int K) {
  const int nk = fl::cdiv(K, BK);
  int maxs = nk / 4;
  if (maxs < 1) maxs = 1;
  if (maxs > 80) maxs = 80;
  return (size_t)maxs * (size_t)M * (size_t)N * sizeof(float);
}

int fl_gemm_f32(const float* A, int64_t lda, int a_layout, const float* B, int64_t ldb, int b_layout,
                float* C, int64_t ldc, int M, int N, int K, const float* bias, float bias_scale,
                const float* R, int64_t ldr, float r_scale, int r_pre, int act, float act_param,
                void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(A && B && C, "fl_gemm_f32: null operand");
  FL_REQUIRE(M >= 0 && N >= 0 && K > 0, "fl_gemm_f32: bad dims M=%d N=%d K=%d", M, N, K);
  GArgs g = base_args();
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.a_vec = is_vec(A, lda);
  g.B = B; g.ldb = ldb; g.b_vec = is_vec(B, ldb);
  g.C = C; g.ldc = ldc; g.out_mode = OUT_ROWMAJOR;
  g.bias = bias; g.bias_scale = bias_scale;
  g.R = R; g.ldr = ldr; g.r_scale = r_scale; g.r_pre = r_pre;
  g.act = act; g.act_param = act_param;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a_layout == FL_LAYOUT_KC && b_layout == FL_LAYOUT_KC) return launch_gemm<A_KC_PLAIN, B_KC_PLAIN>(g, 1, ws, ws_bytes, s);
  if (a_layout == FL_LAYOUT_KC && b_layout == FL_LAYOUT_KS) return launch_gemm<A_KC_PLAIN, B_KS_PLAIN>(g, 1, ws, ws_bytes, s);
  if (a_layout == FL_LAYOUT_KS && b_layout == FL_LAYOUT_KS) return launch_gemm<A_KS_PLAIN, B_KS_PLAIN>(g, 1, ws, ws_bytes, s);
  fl::set_error("fl_gemm_f32: unsupported layout combination a=%d b=%d", a_layout, b_layout);
  return FL_ERR_UNSUPPORTED;
}

int fl_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                  int Cout, int kh, int kw, int stride, int pad, int act, float act_param, const float* R,
                  int out_nchw, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(x && wp && y, "fl_conv2d_fwd: null operand");
  FL_REQUIRE(Cin % 4 == 0, "fl_conv2d_fwd: Cin=%d must be a multiple of 4 (pad RGB to 4)", Cin);
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  FL_REQUIRE(Ho > 0 && Wo > 0, "fl_conv2d_fwd: empty output");
  if (!R && !out_nchw && fl::conv3x3_c4_o64_applicable(B, H, W, Cin, Cout, kh, kw, stride, pad, act))  // RGB stem (conv_first.hip)
    return fl::launch_conv3x3_c4_o64(x, wp, bias, y, nullptr, B, H, W, act, act_param, reinterpret_cast<hipStream_t>(stream), gemm_math());
  GArgs g = base_args();
  g.M = B * Ho * Wo; g.N = Cout; g.K = kh * kw * Cin;
  g.A = x; g.a_vec = is_vec(x, Cin);
  g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.kw = kw; g.stride = stride; g.pad = pad;
  g.B = wp; g.ldb = g.K; g.b_vec = is_vec(wp, g.K);
  g.C = y; g.ldc = Cout; g.out_mode = out_nchw ? OUT_NCHW : OUT_ROWMAJOR;
  g.bias = bias; g.R = R; g.ldr = Cout; g.act = act; g.act_param = act_param;
  FL_REQUIRE(!(out_nchw && R), "fl_conv2d_fwd: residual with NCHW output unsupported");
  return launch_gemm<A_KC_CONV, B_KC_PLAIN>(g, 1, ws, ws_bytes, reinterpret_cast<hipStream_t>(stream));
}

int fl_conv2d_dgrad(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cin, int Cout, int kh,
                    int kw, int stride, int pad, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(dy && wp && dx, "fl_conv2d_dgrad: null operand");
  FL_REQUIRE(Cout % 4 == 0 && Cin % 4 == 0, "fl_conv2d_dgrad: channels must be multiples of 4");
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  GArgs g = base_args();
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stride == 1) {
    // dx[n,iy,ix,ci] = sum_{tap',co} dy[n, iy-(kh-1-pad)+ky', ix-(kw-1-pad)+kx', co] * w[co][flip(tap')][ci]
    FL_REQUIRE(kh - 1 - pad >= 0 && kw - 1 - pad >= 0, "fl_conv2d_dgrad: pad > k-1 unsupported");
    FL_REQUIRE(kh - 1 - pad == kw - 1 - pad, "fl_conv2d_dgrad: square padding only");
    g.M = B * H * W; g.N = Cin; g.K = kh * kw * Cout;
    g.A = dy; g.a_vec = is_vec(dy, Cout);
    g.H = Ho; g.W = Wo; g.Cin = Cout; g.Ho = H; g.Wo = W; g.kw = kw; g.stride = 1; g.pad = kh - 1 - pad;
    g.B = wp; g.b_vec = is_vec(wp, Cin); g.b_taps = kh * kw; g.b_Cred = Cout; g.b_Cn = Cin;
    g.C = dx; g.ldc = Cin; g.out_mode = OUT_ROWMAJOR;
    return launch_gemm<A_KC_CONV, B_KS_TAPFLIP>(g, 1, ws, ws_bytes, s);
  }
  FL_REQUIRE(stride == 2 && kh == 4 && kw == 4 && pad == 1 && H == 2 * Ho && W == 2 * Wo,
             "fl_conv2d_dgrad: stride-2 path needs 4x4 p1 and even input dims");
  // 4 sub-pixel classes; rows per class = (img, yh, xh) over the low-res (Ho, Wo) grid
  g.M = B * Ho * Wo; g.N = Cin; g.K = 4 * Cout;
  g.A = dy; g.a_vec = is_vec(dy, Cout);
  g.H = Ho; g.W = Wo; g.Cin = Cout; g.Ho = Ho; g.Wo = Wo;
  g.B = wp; g.b_vec = is_vec(wp, Cin); g.b_taps = 16; g.b_Cred = Cout; g.b_Cn = Cin;
  g.C = dx; g.ldc = Cin; g.out_mode = OUT_TCONV;
  return launch_gemm<A_KC_TCONV, B_KS_TAPCLS>(g, 4, ws, ws_bytes, s);
}

int fl_convT2d_fwd(const float* x, const float* wtp, const float* bias, float* y, int B, int H, int W, int Cin,
                   int Cout, int act, float act_param, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(x && wtp && y, "fl_convT2d_fwd: null operand");
  FL_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "fl_convT2d_fwd: channels must be multiples of 4");
  GArgs g = base_args();
  g.M = B * H * W; g.N = Cout; g.K = 4 * Cin;
  g.A = x; g.a_vec = is_vec(x, Cin);
  g.H = H; g.W = W; g.Cin = Cin; g.Ho = H; g.Wo = W;
  g.B = wtp; g.b_vec = is_vec(wtp, Cout); g.b_taps = 16; g.b_Cred = Cin; g.b_Cn = Cout;
  g.C = y; g.ldc = Cout; g.out_mode = OUT_TCONV;
  g.bias = bias; g.act = act; g.act_param = act_param;
  return launch_gemm<A_KC_TCONV, B_KS_TAPCLS>(g, 4, ws, ws_bytes, reinterpret_cast<hipStream_t>(stream));
}

size_t fl_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  return fl_gemm_workspace_bytes(Cout, kh * kw * Cin, B * Ho * Wo);
}

int fl_conv2d_wgrad(const float* dy, const float* x, float* dwp, int B, int H, int W, int Cin, int Cout, int kh,
                    int kw, int stride, int pad, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(dy && x && dwp, "fl_conv2d_wgrad: null operand");
  FL_REQUIRE(Cin % 4 == 0, "fl_conv2d_wgrad: Cin must be a multiple of 4");
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  GArgs g = base_args();
  // dwp[co][(tap,ci)] = sum_pix dy[pix][co] * x_col[pix][(tap,ci)]
  g.M = Cout; g.N = kh * kw * Cin; g.K = B * Ho * Wo;
  g.A = dy; g.lda = Cout; g.a_vec = is_vec(dy, Cout);
  g.B = x; g.b_vec = is_vec(x, Cin);
  g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.kw = kw; g.stride = stride; g.pad = pad;
  g.C = dwp; g.ldc = g.N; g.out_mode = OUT_ROWMAJOR;
  return launch_gemm<A_KS_PLAIN, B_KS_IM2COL>(g, 1, ws, ws_bytes, reinterpret_cast<hipStream_t>(stream));
}

int fl_pack_conv_weight(const float* w, float* wp, int O, int I, int kh, int kw, int Ipad, fl_stream_t stream) {
  FL_REQUIRE(w && wp && Ipad >= I, "fl_pack_conv_weight: bad args");
  const long long total = (long long)O * kh * kw * Ipad;
  int blocks = fl::cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_w_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, wp, O, I, kh * kw, Ipad);
  return fl::check_launch("pack_w_kernel");
}
int fl_unpack_conv_weight(const float* wp, float* w, int O, int I, int kh, int kw, int Ipad, fl_stream_t stream) {
  FL_REQUIRE(w && wp && Ipad >= I, "fl_unpack_conv_weight: bad args");
  const long long total = (long long)O * I * kh * kw;
  int blocks = fl::cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpack_w_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wp, w, O, I, kh * kw, Ipad);
  return fl::check_launch("unpack_w_kernel");
}

}  // extern "C"
