// 3x3 / stride 1 / pad 1 convolution in "f16x2" split math: the ResidualBlock convs of the frozen encoder (v2:159-178),
// the dominant kernel of the latent-DDPM step.
//
// Math. Every fp32 operand is scaled by a power of two (exact) and split into TWO fp16 pieces by round-to-nearest,
//   X = x * 2^e,  h = f16(X),  l = f16(X - h)        (X - h is exact in fp32; |X - h - l| <= 2^-22 |X|, unbiased)
// and a product is assembled from 3 of the 4 piece products (l*h, h*l, h*h - smallest first) on
// v_mfma_f32_32x32x16_f16: an f16 x f16 product is exact in fp32, accumulation is the fp32 accumulation of the MFMA. The
// dropped l*l term is <= 2^-22 |xy|. Measured against fp64: 0.5-1.7e-6 of the output's max for K = 1024..4608, i.e. 2-4x the
// native fp32 MFMA kernel's own accumulation error and the same class as the 6-product bf16x3 scheme of gemm.hip
// (tests/test_kernels_gpu.py::test_f16x2_conv_is_fp32_equivalent, profiles/r02_conv_h2_nine_shapes.txt), at 3 matrix
// instructions per 16 k-values instead of 6. The power-of-two scales keep every piece inside fp16's exponent range:
// weights are scaled per tensor at pack time (amax -> [2^14, 2^15), fl_pack_weight_f16x2), activations by 2^a_exp or
// from a device-resident amax word kept by the producing kernel; values > 2^-17 amax keep both pieces in the normal
// range, smaller ones lose at most 2^-40 amax. The epilogue undoes both scales exactly. |x| * 2^e >= 65520 gives inf
// and then NaN in the output - loud, never silently wrong.
//
// Structure (MI355X: one 512-thread block per CU, 2 waves per SIMD):
//   * block tile 256 output pixels x 128 output channels, wave tile 64 x 64 (2 x 2 MFMA tiles of 32 x 32);
//   * A: the tile's input halo ((rows + 2) x (W + 2) pixels) of one 16-channel chunk is loaded ONCE per chunk as fp32,
//     split ONCE by the staging thread into an h plane and an l plane of fp16 ([pixel][16 + 8 pad] halfs, double
//     buffered) and all nine taps read their MFMA A fragments from it with one ds_read_b128 per (32 rows, plane) at a
//     shifted address. (r01 kernel: the fp32 halo was re-split by every wave for every tap, 176 of ~210 VALU
//     instructions per tap; here the inner loop has no VALU work besides addresses.)
//   * B: the weights were split and laid out in MFMA-fragment order once (fl_pack_weight_f16x2); the 8 KB a block needs
//     per (tap, chunk) go global -> LDS by LDS-DMA (global_load_lds_dwordx4, one per wave and tap) into a 3-slot ring,
//     three taps ahead, and are shared by the four waves of a column (r01: every wave fetched its own copy from L2).
//   * one s_barrier per tap (12 MFMAs per wave); fragments of tap t+1 are read under the MFMAs of tap t; LDS-DMA stays in
//     flight across barriers (counted vmcnt, raw s_barrier).
#include "fl_common.h"
#include <mutex>
#include <type_traits>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int H2_HDR = 256;  // bytes in front of the weight fragments: [0] amax bits, [1] 2^-e_w (float)

struct H2Args {
  const float* x;
  const char* wpack;
  const float* bias;
  float* y;
  float* stats;
  const unsigned* a_amax;  // optional device word: bits of a bound on max|x|; NULL -> a_exp
  int a_exp;
  // optional pre-op fused into the halo staging (template PRE): the conv consumes swish(x * pre_scale[b,c] + pre_shift[b,c]),
  // i.e. the producer's LayerNorm2d-apply + Swish pass (fl_affine_act) never touches HBM
  const float* pre_scale;
  const float* pre_shift;
  int Cin, Cout, M, tiles_n, ntiles, act;
  float act_param;
  // in-kernel finish of the fused LayerNorm2d statistics (what conv_inorm_finalize does in a launch of its own), possible where
  // a block's tile holds whole images: fin = 1: H*W == 64, a wave's 64-row slab IS an image; fin = 2: H*W == 256 == the tile, the
  // block's four slabs meet in LDS. Same arithmetic in the same order (doubles): bit-identical scale / shift.
  int fin;
  const float* ln_w;
  const float* ln_b;
  float eps;
  float* scale;
  float* shift;
  float* mean;
  float* rstd;
};

using fl::swish_fast;

// per-(image, channel) LayerNorm2d scale / shift from (sum y, sum y^2) over HW pixels: conv_inorm_finalize's arithmetic
__device__ __forceinline__ void h2_finish_stats(const H2Args& g, int img, int col, double s1, double s2, int HW) {
  const double m = s1 / HW;
  double var = s2 / HW - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)g.eps));
  const float sc = r * (g.ln_w ? g.ln_w[col] : 1.f);
  const size_t i = (size_t)img * g.Cout + col;
  g.scale[i] = sc;
  g.shift[i] = (g.ln_b ? g.ln_b[col] : 0.f) - (float)m * sc;
  if (g.mean) g.mean[i] = (float)m;
  if (g.rstd) g.rstd[i] = r;
}
// epilogue tail shared by both conv kernels: st1 / st2 = this lane's column sums over the wave's 64 rows (two 32-column tiles)
__device__ __forceinline__ void h2_stats_tail(const H2Args& g, const float (&st1)[2], const float (&st2)[2], int tile_m0, int tile_n0,
                                              int wr, int wc, int li, int lh, char* smem) {
  if (!g.stats && !g.fin) return;
  float a[2], b[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    a[tn] = st1[tn] + __shfl_xor(st1[tn], 32, 64);
    b[tn] = st2[tn] + __shfl_xor(st2[tn], 32, 64);
  }
  if (g.fin == 1) {
    const int img = (tile_m0 + wr * 64) / 64;
    if (lh == 0) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) h2_finish_stats(g, img, tile_n0 + wc * 64 + tn * 32 + li, (double)a[tn], (double)b[tn], 64);
    }
  } else if (g.fin == 2) {
    float* ex = reinterpret_cast<float*>(smem);  // [slab 4][sum | sum of squares][128 columns]
    __syncthreads();                             // every wave has left the main loop: the tile buffers are free
    if (lh == 0) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        ex[(wr * 2) * 128 + wc * 64 + tn * 32 + li] = a[tn];
        ex[(wr * 2 + 1) * 128 + wc * 64 + tn * 32 + li] = b[tn];
      }
    }
    __syncthreads();
    if (wr == 0 && lh == 0) {
      const int img = tile_m0 / 256;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int c = wc * 64 + tn * 32 + li;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          s1 += (double)ex[(sl * 2) * 128 + c];
          s2 += (double)ex[(sl * 2 + 1) * 128 + c];
        }
        h2_finish_stats(g, img, tile_n0 + c, s1, s2, 256);
      }
    }
  } else {
    const int grp = (tile_m0 + wr * 64) / 64;
    if (lh == 0) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int col = tile_n0 + wc * 64 + tn * 32 + li;
        g.stats[((size_t)grp * 2) * g.Cout + col] = a[tn];
        g.stats[((size_t)grp * 2 + 1) * g.Cout + col] = b[tn];
      }
    }
  }
}

// the same tail for the 16 x 16 MFMA layout of conv3x3_h2p_kernel: st1 / st2 = this lane's sums over its 16 of the wave's 64 rows
// for four 16-column tiles; lanes kg == 0 end up with the wave's column sums
__device__ __forceinline__ void h2p_stats_tail(const H2Args& g, const float (&st1)[4], const float (&st2)[4], int tile_m0, int tile_n0,
                                               int wr, int wc, int l16, int kg, char* smem) {
  if (!g.stats && !g.fin) return;
  float a[4], b[4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    a[tn] = st1[tn] + __shfl_xor(st1[tn], 16, 64);
    b[tn] = st2[tn] + __shfl_xor(st2[tn], 16, 64);
    a[tn] += __shfl_xor(a[tn], 32, 64);
    b[tn] += __shfl_xor(b[tn], 32, 64);
  }
  if (g.fin == 1) {
    const int img = (tile_m0 + wr * 64) / 64;
    if (kg == 0) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) h2_finish_stats(g, img, tile_n0 + wc * 64 + tn * 16 + l16, (double)a[tn], (double)b[tn], 64);
    }
  } else if (g.fin == 2) {
    float* ex = reinterpret_cast<float*>(smem);  // [slab 4][sum | sum of squares][128 columns]
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        ex[(wr * 2) * 128 + wc * 64 + tn * 16 + l16] = a[tn];
        ex[(wr * 2 + 1) * 128 + wc * 64 + tn * 16 + l16] = b[tn];
      }
    }
    __syncthreads();
    if (wr == 0 && kg == 0) {
      const int img = tile_m0 / 256;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int c = wc * 64 + tn * 16 + l16;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          s1 += (double)ex[(sl * 2) * 128 + c];
          s2 += (double)ex[(sl * 2 + 1) * 128 + c];
        }
        h2_finish_stats(g, img, tile_n0 + c, s1, s2, 256);
      }
    }
  } else {
    const int grp = (tile_m0 + wr * 64) / 64;
    if (kg == 0) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int col = tile_n0 + wc * 64 + tn * 16 + l16;
        g.stats[((size_t)grp * 2) * g.Cout + col] = a[tn];
        g.stats[((size_t)grp * 2 + 1) * g.Cout + col] = b[tn];
      }
    }
  }
}

__device__ __forceinline__ int scale_exp_from_amax(unsigned bits) {
  const int E = (int)((bits >> 23) & 0xffu);
  int e = E ? 141 - E : 0;  // amax * 2^e in [2^14, 2^15)
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return e;
}

// staging schedule of the 3x3 kernel's main loop: step q of nstep goes to MFMA gap 1 + 39 q / nstep of the 40 gaps of taps 3..7
constexpr int h2_step_pos(int q, int nstep) { return 1 + (q * 39) / nstep; }
constexpr int h2_step_at(int G, int nstep) {
  for (int q = 0; q < nstep; ++q)
    if (h2_step_pos(q, nstep) == G) return q;
  return -1;
}
// item j's pre-op coefficients are read in the last gap of the k group BEFORE its first element (-1: the group before the window),
// so that the group's closing lgkmcnt(0) covers them and no wait lands among the MFMAs
constexpr int h2_ssload_pos(int j, int nst) { return 8 * (h2_step_pos(4 * j, 4 * nst) / 8) - 1; }

template <int W, int BM_ = 256>
struct H2Geom {
  static constexpr int BM = BM_, BN = 128;
  static constexpr int R = BM / W, RI = R < W ? R : W, NI = R / RI, HR = RI + 2, HC = W + 2;
  static constexpr int PXB = 48;  // bytes per staged pixel and plane: 16 halfs + 8 pad (odd multiple of 16 B: conflict-free b128)
  static constexpr int PAD_OFF = 32;  // the idle staging lanes write the pad of pixel 0
  // row pitch: the 32 lanes of a fragment read are 32 consecutive output pixels; for W < 32 they span several rows and the
  // jump to the next row must look like "the next W pixels" to the 64 banks: RPB == W * PXB (mod 256 B)
  static constexpr int RPB = W == 32 ? HC * PXB : (W == 16 ? 1024 : 640);
  static constexpr int PLANE_B = NI * HR * RPB, ABUF_B = 2 * PLANE_B;
  static constexpr int NS = 3, SLOT_B = 8192, BOFF = 2 * ABUF_B;
  static constexpr int LDS_B = BOFF + NS * SLOT_B;
  static constexpr int NT = 2 * BM;      // 8 waves for a 256-pixel tile (one block per CU), 4 for 128 (two independent blocks per CU)
  static constexpr int PPW = 512 / NT;   // 1-KB weight-DMA pieces per wave and tap (8 per tap)
  static constexpr int P = NI * HR * HC, ITEMS = P * 4, NST = (ITEMS + NT - 1) / NT;
  static_assert(RPB >= HC * PXB && (W == 32 || (RPB - W * PXB) % 256 == 0), "row pitch");
  static_assert(LDS_B <= 160 * 1024, "LDS");
};

template <int W, int PRE, int BM_ = 256>
__global__ __launch_bounds__(2 * BM_, 2) void conv3x3_h2_kernel(const H2Args g) {
  using G = H2Geom<W, BM_>;
  constexpr int PPW = G::PPW;
  constexpr int BM = G::BM, BN = G::BN, RI = G::RI, NI = G::NI, HR = G::HR, HC = G::HC, PXB = G::PXB, RPB = G::RPB;
  constexpr int PLANE_B = G::PLANE_B, ABUF_B = G::ABUF_B, BOFF = G::BOFF, SLOT_B = G::SLOT_B;
  constexpr int ITEMS = G::ITEMS, NT = G::NT, NST = G::NST;
  static_assert(NST >= 3 && NST <= 5, "staging passes");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int li = lane & 31, lh = lane >> 5;

  int tile;
  {  // XCD-aware order: blocks with equal id mod 8 share an XCD (speed only) -> give each XCD a contiguous tile range
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m0 = (tile / g.tiles_n) * BM;
  const int tile_n0 = (tile % g.tiles_n) * BN;
  const int img0 = tile_m0 / (W * W);
  const int oy0 = (tile_m0 - img0 * (W * W)) / W;
  const int Cin = g.Cin, nC = Cin >> 4;  // 16-channel chunks

  // ---- scales (powers of two, exact)
  const int ea = g.a_amax ? scale_exp_from_amax(*g.a_amax) : g.a_exp;
  const float sa = __uint_as_float((unsigned)(ea + 127) << 23);
  const float out_scale = reinterpret_cast<const float*>(g.wpack)[1] * __uint_as_float((unsigned)(127 - ea) << 23);

  // ---- staging descriptors: item i = tid + 512 j -> (halo pixel i >> 2, float4 i & 3); invariant over chunks
  int goff[NST], sdst[NST], ssoff[NST];
  unsigned vmask = 0u;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i = tid + NT * j;
    const int pix = i >> 2, c4 = i & 3;
    const int nl = pix / (HR * HC), rem = pix - nl * (HR * HC);
    const int hy = rem / HC, hx = rem - hy * HC;
    const int iy = oy0 - 1 + hy, ix = hx - 1;
    const bool ok = (i < ITEMS) && ((unsigned)iy < (unsigned)W) && ((unsigned)ix < (unsigned)W);
    goff[j] = ok ? (((img0 + nl) * W + iy) * W + ix) * Cin + c4 * 4 : 0;
    // threads past the last item write into the 16 pad bytes of pixel 0 (branch-free staging: the conversion is interleaved
    // with MFMAs, which needs one basic block)
    sdst[j] = (i < ITEMS) ? (nl * HR + hy) * RPB + hx * PXB + c4 * 8 : G::PAD_OFF;
    ssoff[j] = (nl < NI ? nl : 0) * Cin + c4 * 4;
    vmask |= (ok ? 1u : 0u) << j;
  }
  // pre-op coefficients of the tile's images: [scale | shift][NI][Cin] floats behind the weight ring
  float* ss = reinterpret_cast<float*>(smem + G::LDS_B);
  if constexpr (PRE) {
    for (int i = tid; i < NI * Cin; i += NT) {
      const int nl = i / Cin, ch = i - nl * Cin;
      ss[i] = g.pre_scale[(size_t)(img0 + nl) * Cin + ch];
      ss[NI * Cin + i] = g.pre_shift[(size_t)(img0 + nl) * Cin + ch];
    }
    __syncthreads();
  }

  // halo loads are inline asm: hipcc must not count them (beside LDS-DMA it would drain vmcnt to 0 at their first use);
  // stage_wait() is the one place their registers become valid
  f32x4v stg[NST];
  auto stage_load = [&](int c) {
    const float* xc = g.x + c * 16;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const float* p = xc + goff[j];
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[j]) : "v"(p) : "memory");
    }
  };
  // s_waitcnt vmcnt(N) that also pins the staging registers (N literal)
#define FL_STAGE_WAIT(N)                                                                                                      \
  do {                                                                                                                        \
    if constexpr (NST == 3)                                                                                                   \
      asm volatile("s_waitcnt vmcnt(%3)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]) : "i"(N) : "memory");                      \
    else if constexpr (NST == 4)                                                                                              \
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[NST > 3 ? 3 : 0]) : "i"(N) : "memory"); \
    else                                                                                                                      \
      asm volatile("s_waitcnt vmcnt(%5)"                                                                                      \
                   : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[NST > 3 ? 3 : 0]), "+v"(stg[NST > 4 ? 4 : 0])           \
                   : "i"(N) : "memory");                                                                                      \
  } while (0)
  // one staging item: pre-op + split + two ds_write_b64 (~65 VALU instructions). Items are spread over the taps after the
  // halo loads have landed and issued INSIDE those taps' MFMA clusters (sched_group_barrier interleave): done in one piece
  // behind a tap's MFMAs they cost ~15 % of the kernel, with every wave converting while the matrix pipe idles
  auto stage_item = [&](auto jc, int buf, int c) {  // c: the chunk in the staging registers
    constexpr int j = decltype(jc)::value;
    if constexpr (j < NST) {
      char* dst = smem + buf * ABUF_B;
      const bool ok = (vmask >> j) & 1u;
      f32x4v v = stg[j];
      if constexpr (PRE) {
        const f32x4v sc = *reinterpret_cast<const f32x4v*>(ss + ssoff[j] + c * 16);
        const f32x4v sh = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[j] + c * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = swish_fast(v[e] * sc[e] + sh[e]);
      }
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float X = ok ? v[e] * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        h[e] = hh;
        l[e] = (_Float16)(X - (float)hh);
      }
      *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = h;
      *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = l;
    }
  };
  auto stage_store = [&](int buf, int c) {
    stage_item(std::integral_constant<int, 0>{}, buf, c);
    stage_item(std::integral_constant<int, 1>{}, buf, c);
    stage_item(std::integral_constant<int, 2>{}, buf, c);
    stage_item(std::integral_constant<int, 3>{}, buf, c);
    stage_item(std::integral_constant<int, 4>{}, buf, c);
  };

  // The same item, cut into slots for the main loop: element e of item j is staging step q = 4 j + e; the chunk's
  // 4 NST steps are dealt over the 40 MFMA gaps of taps 3..7 (h2_step_at), an item's pre-op coefficients are read at the end
  // of the tap before its first element and its two ds_write_b64 follow its last. Pieces wait in registers (sth / stl) in between.
  f32x4v ssc[NST], ssh[NST];
  f16x4_t sth[NST], stl[NST];
#define FL_SS_CH 16
  auto stage_slot = [&](auto tc, auto pc, int buf, int c) {
    constexpr int T = decltype(tc)::value, p = decltype(pc)::value;
    if constexpr (T >= 2 && T <= 7) {
      constexpr int G = (T - 3) * 8 + p;
      constexpr int q = G >= 0 ? h2_step_at(G, 4 * NST) : -1;
      if constexpr (PRE) {
#define FL_SSLOAD(jl)                                                                                 \
        if constexpr (jl < NST && h2_ssload_pos(jl < NST ? jl : 0, NST) == G) {                       \
          ssc[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + ssoff[jl < NST ? jl : 0] + c * FL_SS_CH);               \
          ssh[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[jl < NST ? jl : 0] + c * FL_SS_CH);    \
        }
        FL_SSLOAD(0) FL_SSLOAD(1) FL_SSLOAD(2) FL_SSLOAD(3) FL_SSLOAD(4) FL_SSLOAD(5)
#undef FL_SSLOAD
      }
      if constexpr (q >= 0) {
        constexpr int j = q >> 2, e = q & 3;
        float v = stg[j][e];
        if constexpr (PRE) v = swish_fast(v * ssc[j][e] + ssh[j][e]);
        const float X = ((vmask >> j) & 1u) ? v * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        sth[j][e] = hh;
        stl[j][e] = (_Float16)(X - (float)hh);
        if constexpr (e == 3) {
          char* dst = smem + buf * ABUF_B;
          *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = sth[j];
          *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = stl[j];
        }
      }
    }
  };

  // ---- A fragment addresses (bytes inside a plane): tile row r -> halo pixel of tap (0,0); tap (dy,dx) adds dy*RPB + dx*PXB
  int abase[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int r = wr * 64 + tm * 32 + li;
    const int nl = r / (RI * W), rr = r - nl * (RI * W);
    const int oyl = rr / W, ox = rr - oyl * W;
    abase[tm] = (nl * HR + oyl) * RPB + ox * PXB + 16 * lh;
  }
  const int bfrag = BOFF + wc * 4096 + lane * 16;  // + slot * 8192 + tn * 2048 + piece * 1024

  // ---- B ring: tap (c, t) -> global fragments [g16 = t * nC + c][32-col tile][piece][lane] x 16 B; this wave moves the
  // wid-th KB of the tap's 8 KB
  const int nt32 = g.Cout >> 5;
  const char* wsrc = g.wpack + H2_HDR + ((size_t)(tile_n0 >> 5) * 128 + wid * PPW * 64 + lane) * 16;
  const size_t g16_stride = (size_t)nt32 * 128 * 16;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma_tap = [&](int c, int t, int slot) {  // this wave's PPW of the tap's eight 1-KB pieces
    const int cc = c < nC ? c : nC - 1;  // past the end: a harmless reload into a slot nobody reads
    const char* src = wsrc + (size_t)(t * nC + cc) * g16_stride;
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 1024), (lptr_t)(smem + BOFF + slot * SLOT_B + (wid * PPW + q) * 1024), 16, 0, 0);
  };

  struct Frags {
    f16x8_t a[2][2];  // [tm][plane]
    f16x8_t b[2][2];  // [tn][piece]
  };
  auto frag_load = [&](Frags& f, int abuf, int tapoff, int slot) {  // (prologue only: the main loop reads one fragment per MFMA gap)
    const char* A = smem + abuf * ABUF_B + tapoff;
    const char* Bp = smem + bfrag + slot * SLOT_B;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int p = 0; p < 2; ++p) f.b[tn][p] = *reinterpret_cast<const f16x8_t*>(Bp + tn * 2048 + p * 1024);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int p = 0; p < 2; ++p) f.a[tm][p] = *reinterpret_cast<const f16x8_t*>(A + abase[tm] + p * PLANE_B);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- prologue: chunk 0 staged, taps 0..2 of chunk 0 in the ring, fragments of tap 0 in registers
  stage_load(0);
  dma_tap(0, 0, 0);
  dma_tap(0, 1, 1);
  dma_tap(0, 2, 2);
  FL_STAGE_WAIT(3 * PPW);
  stage_store(0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frags fA, fB;
  frag_load(fA, 0, 0, 0);
  fB = fA;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // one tap: DMA three taps ahead into the slot this tap's fragments came from, fragments of the next tap under this
  // tap's MFMAs, then [counted vmcnt: the DMA of tap t+2 has landed] + lgkmcnt(0) + barrier
  // The tap OPENS with MFMAs. The r02 order [weight DMA, 8 fragment reads, 12 MFMAs] left the matrix pipe
  // empty behind every barrier while both waves of a SIMD issued their DMA piece (60-180 cycles) and their reads; now the
  // first two MFMAs go out at once, one fragment read of the next tap follows each of the next eight, and the DMA piece (and
  // the halo loads of tap 0) sit behind MFMA 9. Every step is pinned by a sched_barrier: the order below is the issue order.
  auto tap = [&](auto tap_c, Frags& cur, Frags& nxt, int c) {
    constexpr int T = decltype(tap_c)::value;
    constexpr int T3 = (T + 3) % 9, C3 = (T + 3) / 9;        // tap t+3
    constexpr int T1 = (T + 1) % 9, C1 = (T + 1) / 9;        // tap t+1
    const char* An = smem + ((c + C1) & 1) * ABUF_B + (T1 / 3) * RPB + (T1 % 3) * PXB;
    const char* Bn = smem + bfrag + (T1 % 3) * SLOT_B;
    auto MF = [&](auto ic) {  // MFMA i of the tap's twelve: pieces l*h, h*l, h*h (smallest first) x (tm, tn)
      constexpr int i = decltype(ic)::value, piece = i >> 2, tm = (i >> 1) & 1, tn = i & 1;
      constexpr int pa = piece == 0 ? 1 : 0, pb = piece == 1 ? 1 : 0;
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[tm][pa], cur.b[tn][pb], acc[tm][tn], 0, 0, 0);
    };
    auto RD = [&](auto jc) {  // fragment read j of the next tap's eight: B first (its DMA landed a tap ago)
      constexpr int j = decltype(jc)::value;
      if constexpr (j < 4) nxt.b[j >> 1][j & 1] = *reinterpret_cast<const f16x8_t*>(Bn + (j >> 1) * 2048 + (j & 1) * 1024);
      else nxt.a[(j - 4) >> 1][j & 1] = *reinterpret_cast<const f16x8_t*>(An + abase[(j - 4) >> 1] + (j & 1) * PLANE_B);
    };
#define FL_I(n) std::integral_constant<int, n>{}
#define FL_SB __builtin_amdgcn_sched_barrier(0)
    MF(FL_I(0)); FL_SB; MF(FL_I(1)); FL_SB;
    // per MFMA of the middle eight: one fragment read of the next tap and - in taps 3..7 - this slot's share of the NEXT chunk's
    // staging (stage_slot below: the items' elements dealt over 40 slots, so no tap carries more VALU work than its MFMAs cover)
    const int cn = c + 1 < nC ? c + 1 : c, bn = (c + 1) & 1;
#define FL_STEP(p) RD(FL_I(p)); FL_SB; stage_slot(tap_c, FL_I(p), bn, cn); FL_SB; MF(FL_I(p + 2)); FL_SB;
    FL_STEP(0) FL_STEP(1) FL_STEP(2) FL_STEP(3) FL_STEP(4) FL_STEP(5) FL_STEP(6) FL_STEP(7)
#undef FL_STEP
    dma_tap(c + C3, T3, T % 3);
    if constexpr (T == 0) stage_load(cn);
    FL_SB;
    MF(FL_I(10)); FL_SB; MF(FL_I(11)); FL_SB;
#undef FL_I
#undef FL_SB
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (T == 0 || T == 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST + PPW) : "memory");
    } else if constexpr (T == 2) {
      FL_STAGE_WAIT(PPW);  // the halo loads of tap 0 are older than DMA(t+2): now complete
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) as a builtin: hipcc then knows the fragment reads have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](auto par, int c) {
    constexpr int PAR = decltype(par)::value;  // which register set holds tap 0's fragments
    Frags& f0 = PAR ? fB : fA;
    Frags& f1 = PAR ? fA : fB;
    tap(std::integral_constant<int, 0>{}, f0, f1, c);
    tap(std::integral_constant<int, 1>{}, f1, f0, c);
    tap(std::integral_constant<int, 2>{}, f0, f1, c);
    tap(std::integral_constant<int, 3>{}, f1, f0, c);
    tap(std::integral_constant<int, 4>{}, f0, f1, c);
    tap(std::integral_constant<int, 5>{}, f1, f0, c);
    tap(std::integral_constant<int, 6>{}, f0, f1, c);
    tap(std::integral_constant<int, 7>{}, f1, f0, c);
    tap(std::integral_constant<int, 8>{}, f0, f1, c);
  };
#pragma unroll 1
  for (int c = 0; c < nC; c += 2) {  // nC is even (Cin % 32 == 0)
    chunk(std::integral_constant<int, 0>{}, c);
    chunk(std::integral_constant<int, 1>{}, c + 1);
  }
  // Drain. The last period requested one more (clamped) set of halo loads that nothing consumes: the wait NAMES the staging
  // registers, so that the compiler keeps them allocated until the loads have landed (with a bare s_waitcnt it moved epilogue
  // index arithmetic into one of them above the wait, and a load that landed late overwrote it: DESIGN 7c)
  FL_STAGE_WAIT(0);  // (and no LDS-DMA may outlive the block's LDS allocation)
#undef FL_STAGE_WAIT

  // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
  float st1[2], st2[2], bv[2];
  const bool has_act = g.act != FL_ACT_NONE;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    st1[tn] = st2[tn] = 0.f;
    bv[tn] = g.bias ? g.bias[tile_n0 + wc * 64 + tn * 32 + li] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = tile_m0 + wr * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* yrow = g.y + (size_t)m * g.Cout + tile_n0 + wc * 64 + li;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        float v = acc[tm][tn][e] * out_scale + bv[tn];
        if (has_act) v = fl::apply_act(v, g.act, g.act_param);  // wave-uniform; the encoder's convs have none
        yrow[tn * 32] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  // fused LayerNorm2d statistics: this wave's 64 rows lie in one image (H*W % 64 == 0): one slab of (sum y, sum y^2) - or,
  // where the tile holds whole images, the finished scale / shift
  h2_stats_tail(g, st1, st2, tile_m0, tile_n0, wr, wc, li, lh, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same 3x3 convolution on v_mfma_f32_16x16x32_f16 ("pair" kernel; Cin % 64 == 0, weights packed in layout 2).
// tools/mfma_shape_probe.hip: with RANDOM operands coming out of LDS the chip sustains 1.43 PFLOP/s of issued f16 MFMA in the
// 32x32x16 shape and 1.60 in the 16x16x32 shape (2.15 / 1.99 on zero operands: the difference is the clock the chip holds under
// load, profiles/r03_mfma_shape_probe.txt) - the 32x32x16 kernel above runs at 1.22, i.e. 85 % of ITS ceiling.
// A 32-wide k step pairs two (tap, 16-channel chunk) groups: lanes 0-31 (k slots 0-15) take the even member, lanes 32-63 the odd
// one - a per-lane byte offset into the same staged planes, whose layout does not change. Over two chunks the 18 taps make 9
// pairs (pair 4 = tap 8 of the even chunk + tap 0 of the odd one: both buffers live), so the loop body is one 2-chunk period.
// Wave tile 64 x 64 = 4 x 4 MFMA tiles, 48 MFMAs (16 cycles each) per pair and ONE barrier per pair (half the barriers).
// Registers: a full double buffer of the fragments (2 x 64 VGPRs) does not fit, so the sets retire piecewise. Piece order
// (a_l b_h), (a_h b_l), (a_h b_h); a_l', b_h' (second buffer), b_l' of the NEXT pair are read as their registers fall free,
// a_h is read INSIDE its pair (under piece 1) - which moves a chunk buffer's last read one pair later: the odd chunk is
// converted in pairs 1-2 (halo loads issued in pair 7, waited for behind pair 0), the even one in pairs 6-7 (loads in pair 3,
// waited for behind pair 5). Weight ring: 3 slots of 16 KB (one pair of the block's 128 columns); a pair's fragments are all read
// during the pair before it, so pair p refills its own slot with pair p + 3.
template <int W, int BM_>
struct H2PGeom {
  static constexpr int BM = BM_, BN = 128;
  static constexpr int R = BM / W, RI = R < W ? R : W, NI = R / RI, HR = RI + 2, HC = W + 2;
#ifndef FL_H2P_PXB
#define FL_H2P_PXB 32  // 48: the 32x32x16 kernel's pitches (A/B build, tools/build_variant_lib.sh)
#endif
  // Pixel pitch 32 B = the 16 halfs, no pad. The A fragment of the 16x16x32 MFMA is read by lane (l16, kg) at pixel l16 + 16 B * (kg & 1),
  // and a ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, ... : with two 16-byte slots per pixel the eight kg-even
  // lanes of a group land on the even slots and the eight kg-odd lanes on the odd ones - conflict-free. (The 32x32x16 kernel's 48-byte
  // pitch, conflict-free for ITS lane map, cost this one a 2-way conflict on every A read, 3-way on the 8-wide maps, and 2-way on the
  // staging writes: SQ_LDS_BANK_CONFLICT was 33 % / 47 % of SQ_LDS_IDX_ACTIVE, profiles/r03_conv_sq_pmc.md.) A group spans two image rows
  // only on the 8-wide maps: there the row pitch is a multiple of 256 B. Every row keeps >= 16 B of pad (the idle staging lanes' target).
  static constexpr int PXB = FL_H2P_PXB;
  static constexpr int RPB = PXB == 32 ? (W == 32 ? HC * PXB + 32 : (W == 16 ? HC * PXB + 64 : 512))
                                       : (W == 32 ? HC * PXB : (W == 16 ? 1024 : 480));
  static constexpr int PAD_OFF = PXB == 32 ? HC * PXB : 32;  // 8 bytes nobody reads, in both planes
  static constexpr int PLANE_B = NI * HR * RPB, ABUF_B = 2 * PLANE_B;
  static constexpr int NS = 3, SLOT_B = 16384, BOFF = 2 * ABUF_B;
  static constexpr int LDS_B = BOFF + NS * SLOT_B;
  static constexpr int NT = 2 * BM, NWV = NT / 64, PPW = 16 / NWV;  // 1-KB weight-DMA pieces per wave and pair
  static constexpr int P = NI * HR * HC, ITEMS = P * 4, NST = (ITEMS + NT - 1) / NT;
  static_assert(RPB >= HC * PXB, "row pitch");
  static_assert(LDS_B + 2 * NI * 512 * 4 <= 160 * 1024, "LDS");
};
// staging schedule of the pair kernel: nstep element steps over ngap MFMA gaps starting at gap 1
constexpr int h2p_step_pos(int q, int nstep, int ngap) { return 1 + (q * (ngap - 1)) / nstep; }
constexpr int h2p_step_at(int G, int nstep, int ngap) {
  for (int q = 0; q < nstep; ++q)
    if (h2p_step_pos(q, nstep, ngap) == G) return q;
  return -1;
}
constexpr int h2p_ssload_pos(int j, int nst, int ngap) { return 48 * (h2p_step_pos(4 * j, 4 * nst, ngap) / 48) - 1; }

template <int W, int PRE, int BM_ = 256>
__global__ __launch_bounds__(2 * BM_, 2) void conv3x3_h2p_kernel(const H2Args g) {
  using G = H2PGeom<W, BM_>;
  constexpr int PPW = G::PPW;
  constexpr int BM = G::BM, BN = G::BN, RI = G::RI, NI = G::NI, HR = G::HR, HC = G::HC, PXB = G::PXB, RPB = G::RPB;
  constexpr int PLANE_B = G::PLANE_B, ABUF_B = G::ABUF_B, BOFF = G::BOFF, SLOT_B = G::SLOT_B;
  constexpr int ITEMS = G::ITEMS, NT = G::NT, NST = G::NST;
  static_assert(NST >= 3 && NST <= 5, "staging passes");
  constexpr int NGAP = 88;  // gaps of a conversion window: a whole pair + the next one up to the point where new halo loads go out
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int l16 = lane & 15, kg = lane >> 4;

  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m0 = (tile / g.tiles_n) * BM;
  const int tile_n0 = (tile % g.tiles_n) * BN;
  const int img0 = tile_m0 / (W * W);
  const int oy0 = (tile_m0 - img0 * (W * W)) / W;
  const int Cin = g.Cin, nC = Cin >> 4;  // 16-channel chunks; nC % 4 == 0

  const int ea = g.a_amax ? scale_exp_from_amax(*g.a_amax) : g.a_exp;
  const float sa = __uint_as_float((unsigned)(ea + 127) << 23);
  const float out_scale = reinterpret_cast<const float*>(g.wpack)[1] * __uint_as_float((unsigned)(127 - ea) << 23);

  // ---- staging descriptors (as in conv3x3_h2_kernel)
  int goff[NST], sdst[NST], ssoff[NST];
  unsigned vmask = 0u;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i = tid + NT * j;
    const int pix = i >> 2, c4 = i & 3;
    const int nl = pix / (HR * HC), rem = pix - nl * (HR * HC);
    const int hy = rem / HC, hx = rem - hy * HC;
    const int iy = oy0 - 1 + hy, ix = hx - 1;
    const bool ok = (i < ITEMS) && ((unsigned)iy < (unsigned)W) && ((unsigned)ix < (unsigned)W);
    goff[j] = ok ? (((img0 + nl) * W + iy) * W + ix) * Cin + c4 * 4 : 0;
    sdst[j] = (i < ITEMS) ? (nl * HR + hy) * RPB + hx * PXB + c4 * 8 : G::PAD_OFF;
    ssoff[j] = (nl < NI ? nl : 0) * Cin + c4 * 4;
    vmask |= (ok ? 1u : 0u) << j;
  }
  float* ss = reinterpret_cast<float*>(smem + G::LDS_B);

  f32x4v stg[NST];
  auto stage_load = [&](int c) {
    const float* xc = g.x + c * 16;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const float* p = xc + goff[j];
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[j]) : "v"(p) : "memory");
    }
  };
#define FL_STAGE_WAIT(N)                                                                                                      \
  do {                                                                                                                        \
    if constexpr (NST == 3)                                                                                                   \
      asm volatile("s_waitcnt vmcnt(%3)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]) : "i"(N) : "memory");                      \
    else if constexpr (NST == 4)                                                                                              \
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[NST > 3 ? 3 : 0]) : "i"(N) : "memory"); \
    else                                                                                                                      \
      asm volatile("s_waitcnt vmcnt(%5)"                                                                                      \
                   : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[NST > 3 ? 3 : 0]), "+v"(stg[NST > 4 ? 4 : 0])           \
                   : "i"(N) : "memory");                                                                                      \
  } while (0)
  auto stage_item = [&](auto jc, int buf, int c) {  // prologue: one whole item
    constexpr int j = decltype(jc)::value;
    if constexpr (j < NST) {
      char* dst = smem + buf * ABUF_B;
      const bool ok = (vmask >> j) & 1u;
      f32x4v v = stg[j];
      if constexpr (PRE) {
        const f32x4v sc = *reinterpret_cast<const f32x4v*>(ss + ssoff[j] + c * 16);
        const f32x4v sh = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[j] + c * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = swish_fast(v[e] * sc[e] + sh[e]);
      }
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float X = ok ? v[e] * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        h[e] = hh;
        l[e] = (_Float16)(X - (float)hh);
      }
      *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = h;
      *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = l;
    }
  };
  // main-loop staging: the chunk's 4 NST element steps dealt over the NGAP MFMA gaps of a conversion window (gap G = 48 * (pair
  // inside the window) + gap inside the pair); an item's coefficients are read in the last gap of the pair before its first element
  f32x4v ssc[NST], ssh[NST];
  f16x4_t sth[NST], stl[NST];
  auto stage_slot = [&](auto gc, int buf, int c) {
    constexpr int G_ = decltype(gc)::value;
    if constexpr (G_ >= -1 && G_ < NGAP) {
      constexpr int q = G_ >= 0 ? h2p_step_at(G_, 4 * NST, NGAP) : -1;
      if constexpr (PRE) {
#define FL_SSLOAD(jl)                                                                                                         \
        if constexpr (jl < NST && h2p_ssload_pos(jl < NST ? jl : 0, NST, NGAP) == G_) {                                        \
          ssc[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + ssoff[jl < NST ? jl : 0] + c * 16);                   \
          ssh[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[jl < NST ? jl : 0] + c * 16);        \
        }
        FL_SSLOAD(0) FL_SSLOAD(1) FL_SSLOAD(2) FL_SSLOAD(3) FL_SSLOAD(4)
#undef FL_SSLOAD
      }
      if constexpr (q >= 0) {
        constexpr int j = q >> 2, e = q & 3;
        float v = stg[j][e];
        if constexpr (PRE) v = swish_fast(v * ssc[j][e] + ssh[j][e]);
        const float X = ((vmask >> j) & 1u) ? v * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        sth[j][e] = hh;
        stl[j][e] = (_Float16)(X - (float)hh);
        if constexpr (e == 3) {
          char* dst = smem + buf * ABUF_B;
          *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = sth[j];
          *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = stl[j];
        }
      }
    }
  };

  // ---- A fragment addresses: tile row r -> halo pixel of tap (0,0) (+ 16 B for the upper 8 channels of the lane's 16-channel
  // group); the lane's tap of a pair: sel = 0 even member, 1 odd member
  int abase[4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int r = wr * 64 + tm * 16 + l16;
    const int nl = r / (RI * W), rr = r - nl * (RI * W);
    const int oyl = rr / W, ox = rr - oyl * W;
    abase[tm] = (nl * HR + oyl) * RPB + ox * PXB + 16 * (kg & 1);
  }
  const int sel = kg >> 1;
  const int bfrag = BOFF + wc * 8192 + lane * 16;  // + slot * 16384 + tn * 2048 + piece * 1024

  // ---- weight ring: pair P -> global fragments [P][16-col tile][piece][lane] x 16 B; this wave moves PPW KB of the pair's 16
  const int nt16 = g.Cout >> 4;
  const char* wsrc = g.wpack + H2_HDR + ((size_t)(tile_n0 >> 4) * 2048 + wid * PPW * 1024 + lane * 16);
  const size_t pair_stride = (size_t)nt16 * 2048;
  const int npairs = (nC >> 1) * 9;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma_pair = [&](int P, int slot) {
    const int pp = P < npairs ? P : npairs - 1;  // past the end: a harmless reload into a slot nobody reads
    const char* src = wsrc + (size_t)pp * pair_stride;
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 1024), (lptr_t)(smem + BOFF + slot * SLOT_B + (wid * PPW + q) * 1024), 16, 0, 0);
  };

  f32x4v acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
  f16x8_t al[4], ah[4], bl[4], bh[4], bn[4];  // bn: b_h of the next pair

  // tap s of the period (0..17): chunk s / 9 (buffer s / 9: the period starts at an even chunk), tap s % 9
  auto tapoff = [](int s2) constexpr { return (s2 / 9) * ABUF_B + ((s2 % 9) / 3) * RPB + ((s2 % 9) % 3) * PXB; };

  // ---- prologue: chunk 0 staged, pairs 0 and 1 in the ring, chunk 1 requested, a_l / b_h / b_l of pair 0 in registers
  stage_load(0);
  dma_pair(0, 0);
  dma_pair(1, 1);
  dma_pair(2, 2);
  // the pre-op coefficients of the tile's images come in BEHIND the first halo loads and weight pieces (in front of them they cost
  // every tile one global round trip before its first load was even requested)
  if constexpr (PRE) {
    for (int i = tid; i < NI * Cin; i += NT) {
      const int nl = i / Cin, ch = i - nl * Cin;
      ss[i] = g.pre_scale[(size_t)(img0 + nl) * Cin + ch];
      ss[NI * Cin + i] = g.pre_shift[(size_t)(img0 + nl) * Cin + ch];
    }
    __syncthreads();
  }
  FL_STAGE_WAIT(3 * PPW);
  stage_item(std::integral_constant<int, 0>{}, 0, 0);
  stage_item(std::integral_constant<int, 1>{}, 0, 0);
  stage_item(std::integral_constant<int, 2>{}, 0, 0);
  stage_item(std::integral_constant<int, 3>{}, 0, 0);
  stage_item(std::integral_constant<int, 4>{}, 0, 0);
  stage_load(1);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"i"(PPW + NST) : "memory");  // the weights of pairs 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    const int po = sel ? tapoff(1) : tapoff(0);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      al[t4] = *reinterpret_cast<const f16x8_t*>(smem + PLANE_B + abase[t4] + po);
      bh[t4] = *reinterpret_cast<const f16x8_t*>(smem + bfrag + t4 * 2048);
      bl[t4] = *reinterpret_cast<const f16x8_t*>(smem + bfrag + t4 * 2048 + 1024);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#define FL_I(n) std::integral_constant<int, n>{}
#define FL_SB __builtin_amdgcn_sched_barrier(0)
  // one pair. c = the period's even chunk. Issue order pinned by sched_barriers: MFMA first, one LDS read (or staging slot) per gap
  auto pair = [&](auto ic, int c) {
    constexpr int I = decltype(ic)::value;      // pair of the period, 0..8
    constexpr int IN = (I + 1) % 9;             // next pair (of the next period when I == 8: buffers 0 | 1 again, chunk + 2)
    const int po = sel ? tapoff(2 * I + 1) : tapoff(2 * I);
    const int pon = sel ? tapoff(2 * IN + 1) : tapoff(2 * IN);
    const char* Bn = smem + bfrag + (IN % 3) * SLOT_B;
    // conversion windows: the odd chunk (c + 1 -> buffer 1) in pairs 1-2, the even chunk of the next period (c + 2 -> buffer 0) in 6-7
    constexpr int WIN = (I == 1 || I == 2) ? 1 : ((I == 6 || I == 7) ? 2 : 0);
    constexpr int GB = (I == 1 || I == 6) ? 0 : 48;  // first gap of this pair inside its window
    const int cs = WIN == 1 ? (c + 1 < nC ? c + 1 : c) : (c + 2 < nC ? c + 2 : c);
    auto SLOT = [&](auto gp) {
      constexpr int gi = decltype(gp)::value;
      if constexpr (WIN != 0) stage_slot(FL_I(GB + gi), WIN == 1 ? 1 : 0, cs);
      else if constexpr (I == 0 || I == 5) {  // the pair before a window: only the first item's coefficient read (gap -1)
        if constexpr (gi == 47) stage_slot(FL_I(-1), I == 0 ? 1 : 0, I == 0 ? (c + 1 < nC ? c + 1 : c) : (c + 2 < nC ? c + 2 : c));
      }
    };
#define FL_MF(AA, BB, t)                                                                                             \
    acc[(t) >> 2][(t) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AA[(t) >> 2], BB[(t) & 3], acc[(t) >> 2][(t) & 3], 0, 0, 0); FL_SB;
    // piece 1: a_l b_h; under it: a_h of THIS pair
    FL_MF(al, bh, 0) FL_MF(al, bh, 1)
    ah[0] = *reinterpret_cast<const f16x8_t*>(smem + abase[0] + po); FL_SB; SLOT(FL_I(0)); FL_SB;
    FL_MF(al, bh, 2)
    ah[1] = *reinterpret_cast<const f16x8_t*>(smem + abase[1] + po); FL_SB; SLOT(FL_I(1)); FL_SB;
    FL_MF(al, bh, 3)
    ah[2] = *reinterpret_cast<const f16x8_t*>(smem + abase[2] + po); FL_SB; SLOT(FL_I(2)); FL_SB;
    FL_MF(al, bh, 4)
    ah[3] = *reinterpret_cast<const f16x8_t*>(smem + abase[3] + po); FL_SB; SLOT(FL_I(3)); FL_SB;
#define FL_MS(AA, BB, t, gi) FL_MF(AA, BB, t) SLOT(FL_I(gi)); FL_SB;
    FL_MS(al, bh, 5, 4) FL_MS(al, bh, 6, 5) FL_MS(al, bh, 7, 6) FL_MS(al, bh, 8, 7) FL_MS(al, bh, 9, 8) FL_MS(al, bh, 10, 9)
    FL_MS(al, bh, 11, 10) FL_MS(al, bh, 12, 11) FL_MS(al, bh, 13, 12) FL_MS(al, bh, 14, 13) FL_MS(al, bh, 15, 14)
    // piece 2: a_h b_l; under it: a_l and b_h of the NEXT pair (a_l is free; b_h goes to its second buffer)
#define FL_MR(AA, BB, t, gi, dst, src)                                                                               \
    FL_MF(AA, BB, t) dst = *reinterpret_cast<const f16x8_t*>(src); FL_SB; SLOT(FL_I(gi)); FL_SB;
    FL_MS(ah, bl, 0, 15) FL_MS(ah, bl, 1, 16)
    FL_MR(ah, bl, 2, 17, al[0], smem + PLANE_B + abase[0] + pon) FL_MR(ah, bl, 3, 18, al[1], smem + PLANE_B + abase[1] + pon)
    FL_MR(ah, bl, 4, 19, al[2], smem + PLANE_B + abase[2] + pon) FL_MR(ah, bl, 5, 20, al[3], smem + PLANE_B + abase[3] + pon)
    FL_MR(ah, bl, 6, 21, bn[0], Bn) FL_MR(ah, bl, 7, 22, bn[1], Bn + 2048)
    FL_MR(ah, bl, 8, 23, bn[2], Bn + 4096) FL_MR(ah, bl, 9, 24, bn[3], Bn + 6144)
    FL_MS(ah, bl, 10, 25) FL_MS(ah, bl, 11, 26) FL_MS(ah, bl, 12, 27) FL_MS(ah, bl, 13, 28) FL_MS(ah, bl, 14, 29) FL_MS(ah, bl, 15, 30)
    // piece 3: a_h b_h; under it: b_l of the next pair, then the weight DMA three pairs ahead (and the halo loads of pairs 3 / 7)
    FL_MS(ah, bh, 0, 31) FL_MS(ah, bh, 1, 32)
    FL_MR(ah, bh, 2, 33, bl[0], Bn + 1024) FL_MR(ah, bh, 3, 34, bl[1], Bn + 2048 + 1024)
    FL_MR(ah, bh, 4, 35, bl[2], Bn + 4096 + 1024) FL_MR(ah, bh, 5, 36, bl[3], Bn + 6144 + 1024)
    FL_MS(ah, bh, 6, 37) FL_MS(ah, bh, 7, 38) FL_MS(ah, bh, 8, 39)
    FL_MF(ah, bh, 9)
    dma_pair((c >> 1) * 9 + I + 3, I % 3);
    if constexpr (I == 3) stage_load(c + 2 < nC ? c + 2 : c);
    if constexpr (I == 7) stage_load(c + 3 < nC ? c + 3 : c);
    FL_SB;
    SLOT(FL_I(40)); FL_SB;
    FL_MS(ah, bh, 10, 41) FL_MS(ah, bh, 11, 42) FL_MS(ah, bh, 12, 43) FL_MS(ah, bh, 13, 44) FL_MS(ah, bh, 14, 45) FL_MS(ah, bh, 15, 46)
    SLOT(FL_I(47)); FL_SB;
#undef FL_MR
#undef FL_MS
#undef FL_MF
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) bh[t4] = bn[t4];
    // vmcnt, oldest first at the end of pair p: DMA(p+2), [halo loads of pair p-1], DMA(p+3), [halo loads of pair p]: DMA(p+2) must
    // have landed; the halo loads issued in pairs 3 / 7 are waited for behind pairs 5 / 0
    if constexpr (I == 3 || I == 7 || I == 4 || I == 8) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW + NST) : "memory");
    } else if constexpr (I == 5 || I == 0) {
      FL_STAGE_WAIT(PPW);
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    FL_SB;
  };
#pragma unroll 1
  for (int c = 0; c < nC; c += 2) {
    pair(FL_I(0), c); pair(FL_I(1), c); pair(FL_I(2), c); pair(FL_I(3), c); pair(FL_I(4), c);
    pair(FL_I(5), c); pair(FL_I(6), c); pair(FL_I(7), c); pair(FL_I(8), c);
  }
#undef FL_I
#undef FL_SB
  // Drain. The last period requested one more (clamped) set of halo loads that nothing consumes: the wait NAMES the staging
  // registers, so that the compiler keeps them allocated until the loads have landed (with a bare s_waitcnt it moved epilogue
  // index arithmetic into one of them above the wait, and a load that landed late overwrote it: DESIGN 7c)
  FL_STAGE_WAIT(0);  // (and no LDS-DMA may outlive the block's LDS allocation)
#undef FL_STAGE_WAIT

  // ---- epilogue (C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg)
  float st1[4], st2[4], bv[4];
  const bool has_act = g.act != FL_ACT_NONE;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    st1[tn] = st2[tn] = 0.f;
    bv[tn] = g.bias ? g.bias[tile_n0 + wc * 64 + tn * 16 + l16] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = tile_m0 + wr * 64 + tm * 16 + 4 * kg + e;
      float* yrow = g.y + (size_t)m * g.Cout + tile_n0 + wc * 64 + l16;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        float v = acc[tm][tn][e] * out_scale + bv[tn];
        if (has_act) v = fl::apply_act(v, g.act, g.act_param);
        yrow[tn * 16] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  h2p_stats_tail(g, st1, st2, tile_m0, tile_n0, wr, wc, l16, kg, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// 4x4 / stride 2 / pad 1 variant (the three down-sampling convs of the encoder, v2:191-213), same math and pipeline.
// The input halo of a 256-output tile is (2 rows + 2) x (2 Wo + 2) pixels - 4x the 3x3 case - so a chunk is EIGHT
// channels and a 16-wide k group pairs two taps: lane half 0 takes tap (ky, 2 kp), half 1 tap (ky, 2 kp + 1), each with
// the chunk's 8 channels (fl_pack_weight_f16x2 with taps = 16 lays the weights out to match). Even and odd input
// columns live in separate LDS planes ([pixel][8 halfs], no padding) so that the 32 lanes of a fragment read - 32
// consecutive output pixels = every second input column - are consecutive 16-byte entries of one plane. 8 k groups
// (12 MFMAs per wave each) per chunk, 4-slot weight ring.
template <int WO>
struct H2Geom4 {
  static constexpr int BM = 256, BN = 128, W = 2 * WO;
  static constexpr int R = BM / WO, RI = R < WO ? R : WO, NI = R / RI, HR = 2 * RI + 2, HC = 2 * WO + 2;
  static constexpr int PXB = 16;
  // plane row pitch: two input rows (= one output row) further must look like "the next WO entries" to the 64 banks
  static constexpr int RPB = WO == 32 ? (WO + 1) * PXB : (WO == 16 ? 384 : 192);
  static constexpr int PAR_B = NI * HR * RPB;      // one column-parity plane
  static constexpr int PLANE_B = 2 * PAR_B;        // h (or l) plane = even + odd columns
  static constexpr int ABUF_B = 2 * PLANE_B;
  static constexpr int NS = 4, SLOT_B = 8192, BOFF = 2 * ABUF_B;
  static constexpr int LDS_B = BOFF + NS * SLOT_B;
  static constexpr int P = NI * HR * HC, ITEMS = P * 2, NT = 512, NST = (ITEMS + NT - 1) / NT;
  static_assert(RPB >= (WO + 1) * PXB && (WO == 32 || (2 * RPB - WO * PXB) % 256 == 0), "row pitch");
  static_assert(LDS_B <= 160 * 1024, "LDS");
};

template <int WO, int PRE>
__global__ __launch_bounds__(512, 2) void conv4x4s2_h2_kernel(const H2Args g) {
  using G = H2Geom4<WO>;
  constexpr int BM = G::BM, BN = G::BN, W = G::W, RI = G::RI, NI = G::NI, HR = G::HR, HC = G::HC, PXB = G::PXB, RPB = G::RPB;
  constexpr int PAR_B = G::PAR_B, PLANE_B = G::PLANE_B, ABUF_B = G::ABUF_B, BOFF = G::BOFF, SLOT_B = G::SLOT_B;
  constexpr int ITEMS = G::ITEMS, NT = G::NT, NST = G::NST;
  static_assert(NST == 5 || NST == 6, "staging passes");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int li = lane & 31, lh = lane >> 5;

  int tile;
  {  // XCD-aware order: blocks with equal id mod 8 share an XCD (speed only) -> give each XCD a contiguous tile range
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m0 = (tile / g.tiles_n) * BM;
  const int tile_n0 = (tile % g.tiles_n) * BN;
  const int img0 = tile_m0 / (WO * WO);
  const int oy0 = (tile_m0 - img0 * (WO * WO)) / WO;
  const int Cin = g.Cin, nC = Cin >> 3;  // 8-channel chunks

  const int ea = g.a_amax ? scale_exp_from_amax(*g.a_amax) : g.a_exp;
  const float sa = __uint_as_float((unsigned)(ea + 127) << 23);
  const float out_scale = reinterpret_cast<const float*>(g.wpack)[1] * __uint_as_float((unsigned)(127 - ea) << 23);

  // ---- staging descriptors: item i = tid + 512 j -> (halo pixel i >> 1, float4 i & 1)
  int goff[NST], sdst[NST], ssoff[NST];
  unsigned vmask = 0u;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    // threads past the last item repeat item i - ITEMS (same bytes to the same place as its owner): branch-free staging
    const int i0 = tid + NT * j, i = i0 < ITEMS ? i0 : i0 - ITEMS;
    const int pix = i >> 1, c4 = i & 1;
    const int nl = pix / (HR * HC), rem = pix - nl * (HR * HC);
    const int hy = rem / HC, hx = rem - hy * HC;
    const int iy = 2 * oy0 - 1 + hy, ix = hx - 1;
    const bool ok = (i < ITEMS) && ((unsigned)iy < (unsigned)W) && ((unsigned)ix < (unsigned)W);
    goff[j] = ok ? (((img0 + nl) * W + iy) * W + ix) * Cin + c4 * 4 : 0;
    sdst[j] = (i < ITEMS) ? (hx & 1) * PAR_B + (nl * HR + hy) * RPB + (hx >> 1) * PXB + c4 * 8 : -1;
    ssoff[j] = (nl < NI ? nl : 0) * Cin + c4 * 4;
    vmask |= (ok ? 1u : 0u) << j;
  }
  float* ss = reinterpret_cast<float*>(smem + G::LDS_B);
  if constexpr (PRE) {
    for (int i = tid; i < NI * Cin; i += NT) {
      const int nl = i / Cin, ch = i - nl * Cin;
      ss[i] = g.pre_scale[(size_t)(img0 + nl) * Cin + ch];
      ss[NI * Cin + i] = g.pre_shift[(size_t)(img0 + nl) * Cin + ch];
    }
    __syncthreads();
  }

  // Halo loads cover TWO chunks (16 channels = 64 contiguous bytes per pixel, two 16-byte loads per thread issued back to back):
  // a chunk's 8 channels alone are 32-byte pieces of a pixel row, and fetching a row's two halves a chunk apart cost down1 1.6x
  // its algorithmic traffic. stg[0][j] = channels 0..7 of the pair (an even chunk), stg[1][j] = channels 8..15 (the odd chunk).
  f32x4v stg[2][NST];
  auto stage_load = [&](int sc) {  // sc: chunk pair
    const float* xc = g.x + sc * 16;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const float* p = xc + goff[j];
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:32"
                   : "=&v"(stg[0][j]), "=&v"(stg[1][j]) : "v"(p) : "memory");
    }
  };
#define FL_STAGE_WAIT4(N)                                                                                                    \
  do {                                                                                                                       \
    if constexpr (NST == 5)                                                                                                  \
      asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                 \
                   : "+v"(stg[0][0]), "+v"(stg[0][1]), "+v"(stg[0][2]), "+v"(stg[0][3]), "+v"(stg[0][4]),                       \
                     "+v"(stg[1][0]), "+v"(stg[1][1]), "+v"(stg[1][2]), "+v"(stg[1][3]), "+v"(stg[1][4])                        \
                   ::"memory");                                                                                              \
    else                                                                                                                     \
      asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                 \
                   : "+v"(stg[0][0]), "+v"(stg[0][1]), "+v"(stg[0][2]), "+v"(stg[0][3]), "+v"(stg[0][4]), "+v"(stg[0][NST - 1]), \
                     "+v"(stg[1][0]), "+v"(stg[1][1]), "+v"(stg[1][2]), "+v"(stg[1][3]), "+v"(stg[1][4]), "+v"(stg[1][NST - 1])  \
                   ::"memory");                                                                                              \
  } while (0)
  auto stage_store = [&](int buf, int c) {  // prologue only: chunk c (0) = the even half of the pair in the staging registers
    char* dst = smem + buf * ABUF_B;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      if (sdst[j] < 0) continue;
      const bool ok = (vmask >> j) & 1u;
      f32x4v v = stg[0][j];
      if constexpr (PRE) {
        const f32x4v sc = *reinterpret_cast<const f32x4v*>(ss + ssoff[j] + c * 8);
        const f32x4v sh = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[j] + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = swish_fast(v[e] * sc[e] + sh[e]);
      }
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float X = ok ? v[e] * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        h[e] = hh;
        l[e] = (_Float16)(X - (float)hh);
      }
      *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = h;
      *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = l;
    }
  };
  // main-loop staging in slots (as in the 3x3 kernel): while chunk c runs, chunk c + 1 = half (c + 1) & 1 of the pair in the
  // registers is converted element by element in the 40 MFMA gaps of k groups 2..6 and written to the other buffer
  f32x4v ssc[NST], ssh[NST];
  f16x4_t sth[NST], stl[NST];
#undef FL_SS_CH
#define FL_SS_CH 8
  auto stage_slot = [&](auto tc, auto pc, auto hc, int buf, int c) {
    constexpr int T = decltype(tc)::value, p = decltype(pc)::value, HALF = decltype(hc)::value;
    if constexpr (T >= 1 && T <= 6) {
      constexpr int G = (T - 2) * 8 + p;
      constexpr int q = G >= 0 ? h2_step_at(G, 4 * NST) : -1;
      if constexpr (PRE) {
#define FL_SSLOAD(jl)                                                                                 \
        if constexpr (jl < NST && h2_ssload_pos(jl < NST ? jl : 0, NST) == G) {                       \
          ssc[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + ssoff[jl < NST ? jl : 0] + c * FL_SS_CH);               \
          ssh[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[jl < NST ? jl : 0] + c * FL_SS_CH);    \
        }
        FL_SSLOAD(0) FL_SSLOAD(1) FL_SSLOAD(2) FL_SSLOAD(3) FL_SSLOAD(4) FL_SSLOAD(5)
#undef FL_SSLOAD
      }
      if constexpr (q >= 0) {
        constexpr int j = q >> 2, e = q & 3;
        float v = stg[HALF][j][e];
        if constexpr (PRE) v = swish_fast(v * ssc[j][e] + ssh[j][e]);
        const float X = ((vmask >> j) & 1u) ? v * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        sth[j][e] = hh;
        stl[j][e] = (_Float16)(X - (float)hh);
        if constexpr (e == 3) {
          char* dst = smem + buf * ABUF_B;
          *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = sth[j];
          *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = stl[j];
        }
      }
    }
  };

  // A fragment address of k group (ky, kp): this lane's tap is (ky, 2 kp + lh) -> column parity lh, entry ox + kp
  int abase[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int r = wr * 64 + tm * 32 + li;
    const int nl = r / (RI * WO), rr = r - nl * (RI * WO);
    const int oyl = rr / WO, ox = rr - oyl * WO;
    abase[tm] = lh * PAR_B + (nl * HR + 2 * oyl) * RPB + ox * PXB;
  }
  const int bfrag = BOFF + wc * 4096 + lane * 16;

  const int nt32 = g.Cout >> 5;
  const char* wsrc = g.wpack + H2_HDR + ((size_t)(tile_n0 >> 5) * 128 + wid * 64 + lane) * 16;
  const size_t g16_stride = (size_t)nt32 * 128 * 16;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma_grp = [&](int c, int kg, int slot) {  // k group (chunk c, kg = 2 ky + kp): fragments [kg * nC + c]
    const int cc = c < nC ? c : nC - 1;
    const char* src = wsrc + (size_t)(kg * nC + cc) * g16_stride;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + BOFF + slot * SLOT_B + wid * 1024), 16, 0, 0);
  };

  struct Frags {
    f16x8_t a[2][2];
    f16x8_t b[2][2];
  };
  auto frag_load = [&](Frags& f, int abuf, int grpoff, int slot) {
    const char* A = smem + abuf * ABUF_B + grpoff;
    const char* Bp = smem + bfrag + slot * SLOT_B;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int p = 0; p < 2; ++p) f.b[tn][p] = *reinterpret_cast<const f16x8_t*>(Bp + tn * 2048 + p * 1024);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int p = 0; p < 2; ++p) f.a[tm][p] = *reinterpret_cast<const f16x8_t*>(A + abase[tm] + p * PLANE_B);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;


  // ---- prologue: chunk pair 0 loaded, its even half staged, k groups 0..3 in the ring, fragments of group 0 in registers
  stage_load(0);
  dma_grp(0, 0, 0);
  dma_grp(0, 1, 1);
  dma_grp(0, 2, 2);
  dma_grp(0, 3, 3);
  FL_STAGE_WAIT4(4);
  stage_store(0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frags fA, fB;
  frag_load(fA, 0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // one k group, MFMA first (see conv3x3_h2_kernel): two MFMAs, then per MFMA one fragment read of the next group and a staging
  // slot, the DMA piece (four groups ahead, into the slot this group's fragments came from) behind MFMA 9 - and, in the last
  // group of an EVEN chunk c, the halo loads of chunks c + 2 | c + 3 (their registers were last read in group 6).
  // vmcnt at the end of group t, oldest first: DMA(t+2), DMA(t+3), DMA(t+4) = 2 may stay; the 2 NST halo loads sit behind
  // DMA(t+4) in group 7 of an even chunk, between DMA(t+3) and DMA(t+4) in group 0 of the odd chunk, and are waited for (with
  // DMA(t+2), which is younger) at the end of its group 1.
  auto grp = [&](auto grp_c, auto par_c, Frags& cur, Frags& nxt, int c) {
    constexpr int T = decltype(grp_c)::value, PAR = decltype(par_c)::value;
    constexpr int T4 = (T + 4) % 8, C4 = (T + 4) / 8;
    constexpr int T1 = (T + 1) % 8, C1 = (T + 1) / 8;
    const char* An = smem + ((c + C1) & 1) * ABUF_B + (T1 >> 1) * RPB + (T1 & 1) * PXB;
    const char* Bn = smem + bfrag + (T1 % 4) * SLOT_B;
    auto MF = [&](auto ic) {
      constexpr int i = decltype(ic)::value, piece = i >> 2, tm = (i >> 1) & 1, tn = i & 1;
      constexpr int pa = piece == 0 ? 1 : 0, pb = piece == 1 ? 1 : 0;
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[tm][pa], cur.b[tn][pb], acc[tm][tn], 0, 0, 0);
    };
    auto RD = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < 4) nxt.b[j >> 1][j & 1] = *reinterpret_cast<const f16x8_t*>(Bn + (j >> 1) * 2048 + (j & 1) * 1024);
      else nxt.a[(j - 4) >> 1][j & 1] = *reinterpret_cast<const f16x8_t*>(An + abase[(j - 4) >> 1] + (j & 1) * PLANE_B);
    };
    const int cn = c + 1 < nC ? c + 1 : c, bn = (c + 1) & 1;
#define FL_I(n) std::integral_constant<int, n>{}
#define FL_SB __builtin_amdgcn_sched_barrier(0)
#define FL_STEP(p) RD(FL_I(p)); FL_SB; stage_slot(grp_c, FL_I(p), FL_I(1 - PAR), bn, cn); FL_SB; MF(FL_I(p + 2)); FL_SB;
    MF(FL_I(0)); FL_SB; MF(FL_I(1)); FL_SB;
    FL_STEP(0) FL_STEP(1) FL_STEP(2) FL_STEP(3) FL_STEP(4) FL_STEP(5) FL_STEP(6) FL_STEP(7)
    dma_grp(c + C4, T4, T % 4);
    if constexpr (T == 7 && PAR == 0) stage_load(c + 2 < nC ? (c + 2) >> 1 : (nC - 1) >> 1);
    FL_SB;
    MF(FL_I(10)); FL_SB; MF(FL_I(11)); FL_SB;
#undef FL_STEP
#undef FL_I
#undef FL_SB
    if constexpr ((T == 7 && PAR == 0) || (T == 0 && PAR == 1)) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NST + 2) : "memory");
    } else if constexpr (T == 1 && PAR == 1) {
      FL_STAGE_WAIT4(2);
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](auto par, int c) {  // 8 groups: the register sets alternate and come back to fA
    grp(std::integral_constant<int, 0>{}, par, fA, fB, c);
    grp(std::integral_constant<int, 1>{}, par, fB, fA, c);
    grp(std::integral_constant<int, 2>{}, par, fA, fB, c);
    grp(std::integral_constant<int, 3>{}, par, fB, fA, c);
    grp(std::integral_constant<int, 4>{}, par, fA, fB, c);
    grp(std::integral_constant<int, 5>{}, par, fB, fA, c);
    grp(std::integral_constant<int, 6>{}, par, fA, fB, c);
    grp(std::integral_constant<int, 7>{}, par, fB, fA, c);
  };
#pragma unroll 1
  for (int c = 0; c < nC; c += 2) {  // nC is even (Cin % 16 == 0)
    chunk(std::integral_constant<int, 0>{}, c);
    chunk(std::integral_constant<int, 1>{}, c + 1);
  }
  // Drain. The last period requested one more (clamped) set of halo loads that nothing consumes: the wait NAMES the staging
  // registers, so that the compiler keeps them allocated until the loads have landed (with a bare s_waitcnt it moved epilogue
  // index arithmetic into one of them above the wait, and a load that landed late overwrote it: DESIGN 7c)
  FL_STAGE_WAIT4(0);
#undef FL_STAGE_WAIT4

  float st1[2], st2[2], bv[2];
  const bool has_act = g.act != FL_ACT_NONE;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    st1[tn] = st2[tn] = 0.f;
    bv[tn] = g.bias ? g.bias[tile_n0 + wc * 64 + tn * 32 + li] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = tile_m0 + wr * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* yrow = g.y + (size_t)m * g.Cout + tile_n0 + wc * 64 + li;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        float v = acc[tm][tn][e] * out_scale + bv[tn];
        if (has_act) v = fl::apply_act(v, g.act, g.act_param);  // wave-uniform; the encoder's convs have none
        yrow[tn * 32] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  h2_stats_tail(g, st1, st2, tile_m0, tile_n0, wr, wc, li, lh, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// The 4x4 / stride 2 convolution on v_mfma_f32_16x16x32_f16 (see conv3x3_h2p_kernel for why that shape). A 32-wide k step is one
// kernel row ky of an 8-channel chunk: lane group g = lane >> 4 takes tap (ky, kx = g) - column parity plane g & 1, entry + (g >> 1),
// a per-lane constant. The weights are read from pack layout 1 AS IT IS (two 16-k groups (ky, kp = 0 | 1) per step: lane (n, g) finds
// its 16 bytes at group g >> 1, lane (g & 1) * 32 + column), so the 32x32x16 kernel above stays usable on the same pack where this
// one does not fit (W_o = 8 with a fused pre-op: ring + coefficients exceed the LDS). Four steps per chunk, one barrier per step,
// fragment sets retired piecewise as in the 3x3 pair kernel; chunk c + 1 is converted in steps 0-1 of chunk c; an even chunk issues
// the halo loads of the next chunk PAIR in step 2 and waits for them behind step 3. Ring: 3 slots of 16 KB, slot index at run time
// (8 steps per loop period is not a multiple of 3).
template <int WO>
struct H2PGeom4 : H2Geom4<WO> {
#ifndef FL_H2P4_PITCH
#define FL_H2P4_PITCH 1  // 0: A/B build with H2Geom4's planes on the 64-wide input stage too
#endif
  // The planes are H2Geom4's. For the 16x16x32 lane map the 64-wide stage's parity plane (9504 B, not a multiple of 256) puts the kg-odd
  // lanes of a ds_read_b128 group on the kg-even lanes' slots: 2-way on every A read (SQ_LDS_BANK_CONFLICT 34 % of down1's LDS cycles,
  // profiles/r03_conv_sq_pmc.md). Row pitch 544 with the plane padded by 192 B clears the reads (brute force of the bank rule): down1
  // 0.2366 -> 0.2332 ms with the pre-op, 0.2178 -> 0.2165 plain (profiles/r03_experiments.md). The 32- and 16-wide stages read
  // conflict-free as they are (their conflicts are staging writes).
  using B4 = H2Geom4<WO>;
  static constexpr int RPB = (FL_H2P4_PITCH && WO == 32) ? 544 : B4::RPB;
  static constexpr int PAR_B = B4::NI * B4::HR * RPB + ((FL_H2P4_PITCH && WO == 32) ? 192 : 0);
  static constexpr int PLANE_B = 2 * PAR_B, ABUF_B = 2 * PLANE_B;
  static constexpr int NS = 3, SLOT_B = 16384, BOFF = 2 * ABUF_B;
  static constexpr int LDS_B = BOFF + NS * SLOT_B;
};

template <int WO, int PRE>
__global__ __launch_bounds__(512, 2) void conv4x4s2_h2p_kernel(const H2Args g) {
  using G = H2PGeom4<WO>;
  constexpr int BM = G::BM, BN = G::BN, W = G::W, RI = G::RI, NI = G::NI, HR = G::HR, HC = G::HC, PXB = G::PXB, RPB = G::RPB;
  constexpr int PAR_B = G::PAR_B, PLANE_B = G::PLANE_B, ABUF_B = G::ABUF_B, BOFF = G::BOFF, SLOT_B = G::SLOT_B;
  constexpr int ITEMS = G::ITEMS, NT = G::NT, NST = G::NST;
  static_assert(NST == 5 || NST == 6, "staging passes");
  constexpr int NGAP = 88;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int l16 = lane & 15, kg = lane >> 4;

  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m0 = (tile / g.tiles_n) * BM;
  const int tile_n0 = (tile % g.tiles_n) * BN;
  const int img0 = tile_m0 / (WO * WO);
  const int oy0 = (tile_m0 - img0 * (WO * WO)) / WO;
  const int Cin = g.Cin, nC = Cin >> 3;  // 8-channel chunks; nC is even

  const int ea = g.a_amax ? scale_exp_from_amax(*g.a_amax) : g.a_exp;
  const float sa = __uint_as_float((unsigned)(ea + 127) << 23);
  const float out_scale = reinterpret_cast<const float*>(g.wpack)[1] * __uint_as_float((unsigned)(127 - ea) << 23);

  int goff[NST], sdst[NST], ssoff[NST];
  unsigned vmask = 0u;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i0 = tid + NT * j, i = i0 < ITEMS ? i0 : i0 - ITEMS;  // (threads past the last item repeat item i - ITEMS)
    const int pix = i >> 1, c4 = i & 1;
    const int nl = pix / (HR * HC), rem = pix - nl * (HR * HC);
    const int hy = rem / HC, hx = rem - hy * HC;
    const int iy = 2 * oy0 - 1 + hy, ix = hx - 1;
    const bool ok = ((unsigned)iy < (unsigned)W) && ((unsigned)ix < (unsigned)W);
    goff[j] = ok ? (((img0 + nl) * W + iy) * W + ix) * Cin + c4 * 4 : 0;
    sdst[j] = (hx & 1) * PAR_B + (nl * HR + hy) * RPB + (hx >> 1) * PXB + c4 * 8;
    ssoff[j] = (nl < NI ? nl : 0) * Cin + c4 * 4;
    vmask |= (ok ? 1u : 0u) << j;
  }
  float* ss = reinterpret_cast<float*>(smem + G::LDS_B);

  f32x4v stg[2][NST];  // [half of the chunk pair][item]
  auto stage_load = [&](int sc) {
    const float* xc = g.x + sc * 16;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const float* p = xc + goff[j];
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:32"
                   : "=&v"(stg[0][j]), "=&v"(stg[1][j]) : "v"(p) : "memory");
    }
  };
#define FL_STAGE_WAIT4(N)                                                                                                    \
  do {                                                                                                                       \
    if constexpr (NST == 5)                                                                                                  \
      asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                 \
                   : "+v"(stg[0][0]), "+v"(stg[0][1]), "+v"(stg[0][2]), "+v"(stg[0][3]), "+v"(stg[0][4]),                       \
                     "+v"(stg[1][0]), "+v"(stg[1][1]), "+v"(stg[1][2]), "+v"(stg[1][3]), "+v"(stg[1][4])                        \
                   ::"memory");                                                                                              \
    else                                                                                                                     \
      asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                 \
                   : "+v"(stg[0][0]), "+v"(stg[0][1]), "+v"(stg[0][2]), "+v"(stg[0][3]), "+v"(stg[0][4]), "+v"(stg[0][NST - 1]), \
                     "+v"(stg[1][0]), "+v"(stg[1][1]), "+v"(stg[1][2]), "+v"(stg[1][3]), "+v"(stg[1][4]), "+v"(stg[1][NST - 1])  \
                   ::"memory");                                                                                              \
  } while (0)
  auto stage_store = [&](int buf, int c) {  // prologue only: chunk c (0) = the even half of the pair in the staging registers
    char* dst = smem + buf * ABUF_B;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const bool ok = (vmask >> j) & 1u;
      f32x4v v = stg[0][j];
      if constexpr (PRE) {
        const f32x4v sc = *reinterpret_cast<const f32x4v*>(ss + ssoff[j] + c * 8);
        const f32x4v sh = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[j] + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = swish_fast(v[e] * sc[e] + sh[e]);
      }
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float X = ok ? v[e] * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        h[e] = hh;
        l[e] = (_Float16)(X - (float)hh);
      }
      *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = h;
      *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = l;
    }
  };
  f32x4v ssc[NST], ssh[NST];
  f16x4_t sth[NST], stl[NST];
  auto stage_slot = [&](auto gc, auto hc, int buf, int c) {  // gap G_ of the conversion window, half HALF of the staged pair
    constexpr int G_ = decltype(gc)::value, HALF = decltype(hc)::value;
    if constexpr (G_ >= -1 && G_ < NGAP) {
      constexpr int q = G_ >= 0 ? h2p_step_at(G_, 4 * NST, NGAP) : -1;
      if constexpr (PRE) {
#define FL_SSLOAD(jl)                                                                                                         \
        if constexpr (jl < NST && h2p_ssload_pos(jl < NST ? jl : 0, NST, NGAP) == G_) {                                        \
          ssc[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + ssoff[jl < NST ? jl : 0] + c * 8);                    \
          ssh[jl < NST ? jl : 0] = *reinterpret_cast<const f32x4v*>(ss + NI * Cin + ssoff[jl < NST ? jl : 0] + c * 8);         \
        }
        FL_SSLOAD(0) FL_SSLOAD(1) FL_SSLOAD(2) FL_SSLOAD(3) FL_SSLOAD(4) FL_SSLOAD(5)
#undef FL_SSLOAD
      }
      if constexpr (q >= 0) {
        constexpr int j = q >> 2, e = q & 3;
        float v = stg[HALF][j][e];
        if constexpr (PRE) v = swish_fast(v * ssc[j][e] + ssh[j][e]);
        const float X = ((vmask >> j) & 1u) ? v * sa : 0.f;
        const _Float16 hh = (_Float16)X;
        sth[j][e] = hh;
        stl[j][e] = (_Float16)(X - (float)hh);
        if constexpr (e == 3) {
          char* dst = smem + buf * ABUF_B;
          *reinterpret_cast<f16x4_t*>(dst + sdst[j]) = sth[j];
          *reinterpret_cast<f16x4_t*>(dst + PLANE_B + sdst[j]) = stl[j];
        }
      }
    }
  };

  // A fragment base of this lane: tap kx = kg -> parity plane kg & 1, entry ox + (kg >> 1); a step adds ky * RPB
  int abase[4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int r = wr * 64 + tm * 16 + l16;
    const int nl = r / (RI * WO), rr = r - nl * (RI * WO);
    const int oyl = rr / WO, ox = rr - oyl * WO;
    abase[tm] = (kg & 1) * PAR_B + (nl * HR + 2 * oyl) * RPB + (ox + (kg >> 1)) * PXB;
  }
  // B fragment of 16-col tile tn, piece p (pack layout 1 in the slot: [kp][32-col tile][piece][lane] x 16 B)
  const int bfrag = BOFF + (kg >> 1) * 8192 + ((kg & 1) * 32 + l16) * 16 + wc * 4096;  // + slot * 16384 + (tn >> 1) * 2048 + (tn & 1) * 256 + p * 1024

  const int nt32 = g.Cout >> 5;
  const char* wsrc = g.wpack + H2_HDR + ((size_t)(tile_n0 >> 5) * 128 + wid * 64 + lane) * 16;
  const size_t g16_stride = (size_t)nt32 * 128 * 16;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma_step = [&](int c, int ky, int slot) {  // both 16-k groups (ky, kp) of chunk c; this wave's KB of each
    const int cc = c < nC ? c : nC - 1;
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      const char* src = wsrc + (size_t)((2 * ky + kp) * nC + cc) * g16_stride;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + BOFF + slot * SLOT_B + kp * 8192 + wid * 1024), 16, 0, 0);
    }
  };

  f32x4v acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
  f16x8_t al[4], ah[4], bl[4], bh[4], bn[4];

  // ---- prologue
  stage_load(0);
  dma_step(0, 0, 0);
  dma_step(0, 1, 1);
  dma_step(0, 2, 2);
  // the pre-op coefficients of the tile's images come in BEHIND the first halo loads and weight pieces (in front of them they cost
  // every tile one global round trip before its first load was even requested)
  if constexpr (PRE) {
    for (int i = tid; i < NI * Cin; i += NT) {
      const int nl = i / Cin, ch = i - nl * Cin;
      ss[i] = g.pre_scale[(size_t)(img0 + nl) * Cin + ch];
      ss[NI * Cin + i] = g.pre_shift[(size_t)(img0 + nl) * Cin + ch];
    }
    __syncthreads();
  }
  FL_STAGE_WAIT4(6);
  stage_store(0, 0);
  stage_slot(std::integral_constant<int, -1>{}, std::integral_constant<int, 1>{}, 1, nC > 1 ? 1 : 0);  // (the first window's coefficient reads)
  asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");  // the weights of steps 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    al[t4] = *reinterpret_cast<const f16x8_t*>(smem + PLANE_B + abase[t4]);
    bh[t4] = *reinterpret_cast<const f16x8_t*>(smem + bfrag + (t4 >> 1) * 2048 + (t4 & 1) * 256);
    bl[t4] = *reinterpret_cast<const f16x8_t*>(smem + bfrag + (t4 >> 1) * 2048 + (t4 & 1) * 256 + 1024);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int sl = 0;  // ring slot of the running step (wave-uniform)
#define FL_I(n) std::integral_constant<int, n>{}
#define FL_SB __builtin_amdgcn_sched_barrier(0)
  auto step = [&](auto kyc, auto parc, int c) {
    constexpr int KY = decltype(kyc)::value, PAR = decltype(parc)::value;  // PAR = c & 1
    constexpr int KYN = (KY + 1) & 3, CN = (KY + 1) >> 2;                  // next step: row KYN of chunk c + CN
    const int sln = sl == 2 ? 0 : sl + 1;
    const char* Ah = smem + PAR * ABUF_B + KY * RPB;                       // this step's a_h (read inside the step)
    const char* Aln = smem + ((PAR + CN) & 1) * ABUF_B + PLANE_B + KYN * RPB;
    const char* Bn = smem + bfrag + sln * SLOT_B;
    const int cs = c + 1 < nC ? c + 1 : c;                                 // the chunk being converted while chunk c runs
    auto SLOT = [&](auto gp) {
      constexpr int gi = decltype(gp)::value;
      if constexpr (KY < 2) stage_slot(FL_I(48 * KY + gi), FL_I(1 - PAR), 1 - PAR, cs);
      else if constexpr (KY == 3 && gi == 47)                              // next chunk's window: its first item's coefficient read
        stage_slot(FL_I(-1), FL_I(PAR), PAR, c + 2 < nC ? c + 2 : (c + 1 < nC ? c + 1 : c));
    };
#define FL_MF(AA, BB, t)                                                                                             \
    acc[(t) >> 2][(t) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AA[(t) >> 2], BB[(t) & 3], acc[(t) >> 2][(t) & 3], 0, 0, 0); FL_SB;
#define FL_MS(AA, BB, t, gi) FL_MF(AA, BB, t) SLOT(FL_I(gi)); FL_SB;
#define FL_MR(AA, BB, t, gi, dst, src)                                                                               \
    FL_MF(AA, BB, t) dst = *reinterpret_cast<const f16x8_t*>(src); FL_SB; SLOT(FL_I(gi)); FL_SB;
    // piece 1: a_l b_h; under it a_h of THIS step
    FL_MF(al, bh, 0) FL_MF(al, bh, 1)
    ah[0] = *reinterpret_cast<const f16x8_t*>(Ah + abase[0]); FL_SB; SLOT(FL_I(0)); FL_SB;
    FL_MF(al, bh, 2)
    ah[1] = *reinterpret_cast<const f16x8_t*>(Ah + abase[1]); FL_SB; SLOT(FL_I(1)); FL_SB;
    FL_MF(al, bh, 3)
    ah[2] = *reinterpret_cast<const f16x8_t*>(Ah + abase[2]); FL_SB; SLOT(FL_I(2)); FL_SB;
    FL_MF(al, bh, 4)
    ah[3] = *reinterpret_cast<const f16x8_t*>(Ah + abase[3]); FL_SB; SLOT(FL_I(3)); FL_SB;
    FL_MS(al, bh, 5, 4) FL_MS(al, bh, 6, 5) FL_MS(al, bh, 7, 6) FL_MS(al, bh, 8, 7) FL_MS(al, bh, 9, 8) FL_MS(al, bh, 10, 9)
    FL_MS(al, bh, 11, 10) FL_MS(al, bh, 12, 11) FL_MS(al, bh, 13, 12) FL_MS(al, bh, 14, 13) FL_MS(al, bh, 15, 14)
    // piece 2: a_h b_l; under it a_l and b_h of the NEXT step
    FL_MS(ah, bl, 0, 15) FL_MS(ah, bl, 1, 16)
    FL_MR(ah, bl, 2, 17, al[0], Aln + abase[0]) FL_MR(ah, bl, 3, 18, al[1], Aln + abase[1])
    FL_MR(ah, bl, 4, 19, al[2], Aln + abase[2]) FL_MR(ah, bl, 5, 20, al[3], Aln + abase[3])
    FL_MR(ah, bl, 6, 21, bn[0], Bn) FL_MR(ah, bl, 7, 22, bn[1], Bn + 256)
    FL_MR(ah, bl, 8, 23, bn[2], Bn + 2048) FL_MR(ah, bl, 9, 24, bn[3], Bn + 2048 + 256)
    FL_MS(ah, bl, 10, 25) FL_MS(ah, bl, 11, 26) FL_MS(ah, bl, 12, 27) FL_MS(ah, bl, 13, 28) FL_MS(ah, bl, 14, 29) FL_MS(ah, bl, 15, 30)
    // piece 3: a_h b_h; under it b_l of the next step, then the weight DMA three steps ahead (+ the halo loads in step 2 of an even chunk)
    FL_MS(ah, bh, 0, 31) FL_MS(ah, bh, 1, 32)
    FL_MR(ah, bh, 2, 33, bl[0], Bn + 1024) FL_MR(ah, bh, 3, 34, bl[1], Bn + 256 + 1024)
    FL_MR(ah, bh, 4, 35, bl[2], Bn + 2048 + 1024) FL_MR(ah, bh, 5, 36, bl[3], Bn + 2048 + 256 + 1024)
    FL_MS(ah, bh, 6, 37) FL_MS(ah, bh, 7, 38) FL_MS(ah, bh, 8, 39)
    FL_MF(ah, bh, 9)
    dma_step(c + ((KY + 3) >> 2), (KY + 3) & 3, sl);
    if constexpr (KY == 2 && PAR == 0) stage_load(c + 2 < nC ? (c + 2) >> 1 : (nC - 1) >> 1);
    FL_SB;
    SLOT(FL_I(40)); FL_SB;
    FL_MS(ah, bh, 10, 41) FL_MS(ah, bh, 11, 42) FL_MS(ah, bh, 12, 43) FL_MS(ah, bh, 13, 44) FL_MS(ah, bh, 14, 45) FL_MS(ah, bh, 15, 46)
    SLOT(FL_I(47)); FL_SB;
#undef FL_MR
#undef FL_MS
#undef FL_MF
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) bh[t4] = bn[t4];
    sl = sln;
    // vmcnt, oldest first at the end of step s: DMA(s+2), [halo loads of step s-1], DMA(s+3), [halo loads of step s]
    if constexpr (KY == 2 && PAR == 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 + 2 * NST) : "memory");
    } else if constexpr (KY == 3 && PAR == 0) {
      FL_STAGE_WAIT4(2);
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    FL_SB;
  };
#pragma unroll 1
  for (int c = 0; c < nC; c += 2) {
    step(FL_I(0), FL_I(0), c); step(FL_I(1), FL_I(0), c); step(FL_I(2), FL_I(0), c); step(FL_I(3), FL_I(0), c);
    step(FL_I(0), FL_I(1), c + 1); step(FL_I(1), FL_I(1), c + 1); step(FL_I(2), FL_I(1), c + 1); step(FL_I(3), FL_I(1), c + 1);
  }
#undef FL_I
#undef FL_SB
  // Drain. The last period requested one more (clamped) set of halo loads that nothing consumes: the wait NAMES the staging
  // registers, so that the compiler keeps them allocated until the loads have landed (with a bare s_waitcnt it moved epilogue
  // index arithmetic into one of them above the wait, and a load that landed late overwrote it: DESIGN 7c)
  FL_STAGE_WAIT4(0);
#undef FL_STAGE_WAIT4

  float st1[4], st2[4], bv[4];
  const bool has_act = g.act != FL_ACT_NONE;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    st1[tn] = st2[tn] = 0.f;
    bv[tn] = g.bias ? g.bias[tile_n0 + wc * 64 + tn * 16 + l16] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = tile_m0 + wr * 64 + tm * 16 + 4 * kg + e;
      float* yrow = g.y + (size_t)m * g.Cout + tile_n0 + wc * 64 + l16;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        float v = acc[tm][tn][e] * out_scale + bv[tn];
        if (has_act) v = fl::apply_act(v, g.act, g.act_param);
        yrow[tn * 16] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  h2p_stats_tail(g, st1, st2, tile_m0, tile_n0, wr, wc, l16, kg, smem);
}

// ---- weight pack: fp32 [N][K] -> header + [16-k group][32-col tile][piece h|l][lane][8 halfs]
__global__ void h2_amax_kernel(const float* __restrict__ w, long long n, unsigned* hdr) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[i]));
  m = fl::wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(hdr, __float_as_uint(m));  // non-negative floats order like their bits
}

// layout 0: k group g16 = 16 consecutive k (3x3 conv: [tap][Cin/16] groups; any K-contiguous operand);
// layout 1 (4x4 taps, K = 16 Cin): group g16 = kg * (Cin/8) + c8 with kg = 2 ky + kp; its k slot 8 lh + j is tap
// (ky, 2 kp + lh), channel 8 c8 + j - the pairing conv4x4s2_h2_kernel reads
__global__ void h2_pack_kernel(const float* __restrict__ w, char* __restrict__ out, int N, int K, int layout) {
  const unsigned* hdr = reinterpret_cast<const unsigned*>(out);
  const int e = scale_exp_from_amax(hdr[0]);
  const float s = __uint_as_float((unsigned)(e + 127) << 23);
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) reinterpret_cast<float*>(out)[1] = __uint_as_float((unsigned)(127 - e) << 23);
  const int nt = N >> 5;
  if (i >= (long long)(K >> 4) * nt * 64) return;
  const int lane = (int)(i & 63), lr = lane & 31, lh = lane >> 5;
  const long long gt = i >> 6;
  const int t = (int)(gt % nt);
  const long long g16 = gt / nt;
  long long k0 = 16 * g16 + 8 * lh;  // k slots of this lane: 8*lh + j
  if (layout == 2) {
    // 3x3 taps, K = 9 Cin, Cin % 64 == 0, for v_mfma_f32_16x16x32_f16 (conv3x3_h2p_kernel): 1-KB unit u = [pair P][16-col tile t][piece]
    // with P = 9 (chunk pair) + i; lane (n = lane & 15, kg = lane >> 4) holds tap s = 2 i + (kg >> 1) of the pair's 18
    // (chunk 2 cp + s / 9, tap s % 9), channels 8 (kg & 1) .. + 7 of that 16-channel chunk
    const int Cin = K / 9, nt16 = N >> 4;
    const long long P = gt / nt16;  // (gt = unit pair index here: the same count of 64-lane groups, N / 16 per pair)
    const int t16 = (int)(gt - P * nt16), kgl = lane >> 4, n = 16 * t16 + (lane & 15);
    const int cp = (int)(P / 9), s2 = 2 * (int)(P - 9ll * cp) + (kgl >> 1);
    const long long kk = (long long)(s2 % 9) * Cin + 16 * (2 * cp + s2 / 9) + 8 * (kgl & 1);
    const float* row2 = w + (long long)n * K + kk;
    const float4 p0 = *reinterpret_cast<const float4*>(row2), p1 = *reinterpret_cast<const float4*>(row2 + 4);
    const float xx[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    f16x8_t h2, l2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float X = xx[j] * s;
      const _Float16 hh = (_Float16)X;
      h2[j] = hh;
      l2[j] = (_Float16)(X - (float)hh);
    }
    f16x8_t* o2 = reinterpret_cast<f16x8_t*>(out + H2_HDR) + (gt * 2) * 64 + lane;
    o2[0] = h2;
    o2[64] = l2;
    return;
  }
  if (layout == 1) {
    const int Cin = K >> 4, nC8 = Cin >> 3;
    const int kg = (int)(g16 / nC8), c8 = (int)(g16 - (long long)kg * nC8);
    k0 = (long long)((kg >> 1) * 4 + 2 * (kg & 1) + lh) * Cin + 8 * c8;
  }
  const float* row = w + (long long)(32 * t + lr) * K + k0;
  const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
  const float x[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
  f16x8_t h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float X = x[j] * s;
    const _Float16 hh = (_Float16)X;
    h[j] = hh;
    l[j] = (_Float16)(X - (float)hh);
  }
  f16x8_t* o = reinterpret_cast<f16x8_t*>(out + H2_HDR) + (gt * 2) * 64 + lane;
  o[0] = h;
  o[64] = l;
}

template <int W>
void h2_set_attr() {
  const int mx = 160 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2_kernel<W, 0, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2_kernel<W, 1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2_kernel<W, 0, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2_kernel<W, 1, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2p_kernel<W, 0, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2p_kernel<W, 1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2p_kernel<W, 0, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_h2p_kernel<W, 1, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_h2p_kernel<W, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_h2p_kernel<W, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_h2_kernel<W, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_h2_kernel<W, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
}

// M-tile of the 3x3 kernel: 256 pixels (8 waves, one block per CU) or 128 (4 waves; two INDEPENDENT blocks per CU, whose
// prologues / epilogues / barriers overlap each other's MFMA clusters). FL_H2_BM overrides the per-shape default.
int h2_bm(int W, int Cin, bool pre, long long M, int Cout) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("FL_H2_BM"); forced = e ? atoi(e) : 0; }
  int bm = forced ? forced : 256;
  // fewer 256-pixel tiles than CUs (batch 64 on the 16-wide stage): 128-pixel tiles fill the chip
  if (!forced && M % 128 == 0 && (M / 256) * (Cout / 128) < 256) bm = 128;
  if (bm == 128) {  // two blocks must fit the CU's LDS
    const int ni = (128 / W) > W ? (128 / W) / W : 1;
    const int lds = (W == 32 ? H2Geom<32, 128>::LDS_B : W == 16 ? H2Geom<16, 128>::LDS_B : H2Geom<8, 128>::LDS_B) + (pre ? 2 * ni * Cin * 4 : 0);
    if (2 * lds > 160 * 1024) bm = 256;
  }
  if (M % bm) bm = (M % 128 == 0) ? 128 : 256;
  return bm;
}


// ---------------------------------------------------------------------------------------------------------------------
// nn.Linear with a huge reduction in f16x2 split math: the frozen encoder's two heads (v2:210-221: 32768 -> 512 each, batch
// 256: 17 GFLOP that the generic split-K kernel ran at 91 TFLOP/s). y[M, N] = x[M, K] W[N, K]^T as split-K slabs:
// block = 256 rows x 128 columns x (K / splits), 8 waves of 64 x 64 like the conv kernels, 12 MFMAs per wave and 16-k group.
// Nothing here is reused across taps, so there is no halo and no DMA ring: the weight fragments (packed once, frozen; the
// block's 16 KB per 32 k) and the activations (split on the way) go global -> registers -> LDS one chunk of two 16-k groups
// at a time, requested three chunks ahead of their use. Plain loads and __syncthreads only: hipcc counts the waits.
struct LinH2Args {
  const float* x;
  const char* wpack;
  float* slabs;              // [splits][M][N]
  const unsigned* a_amax;
  int a_exp, lda, M, N, K, tiles_n, splits, chunks;  // chunks (of 32 k) per split
};

constexpr int LIN_PXB = 48, LIN_PLANE_B = 256 * LIN_PXB, LIN_GROUP_B = 2 * LIN_PLANE_B, LIN_ABUF_B = 2 * LIN_GROUP_B;
constexpr int LIN_BBUF_B = 2 * 8192, LIN_BOFF = 2 * LIN_ABUF_B;  // B: [group][tile 0..3][piece][lane] x 16 B per buffer
constexpr int LIN_LDS_B = LIN_BOFF + 2 * LIN_BBUF_B;

__global__ __launch_bounds__(512) void linear_h2_kernel(const LinH2Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1, li = lane & 31, lh = lane >> 5;
  const int sp = blockIdx.x % g.splits, rest = blockIdx.x / g.splits;
  const int tile_n0 = (rest % g.tiles_n) * 128, tile_m0 = (rest / g.tiles_n) * 256;
  const int ea = g.a_amax ? scale_exp_from_amax(*g.a_amax) : g.a_exp;
  const float sa = __uint_as_float((unsigned)(ea + 127) << 23);
  const float out_scale = reinterpret_cast<const float*>(g.wpack)[1] * __uint_as_float((unsigned)(127 - ea) << 23);
  const int nt32 = g.N >> 5, nch = g.chunks;
  const long long grp0 = (long long)sp * nch * 2;  // first 16-k group of this split

  // A staging items: i = tid + 512 j -> row i >> 3, float4 q = i & 7 of the chunk's 32 floats (group q >> 2, c4 = q & 3)
  const float* asrc[4];
  int adst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 512 * j, row = i >> 3, q = i & 7;
    asrc[j] = g.x + (size_t)(tile_m0 + row) * g.lda + grp0 * 16 + q * 4;
    adst[j] = (q >> 2) * LIN_GROUP_B + row * LIN_PXB + (q & 3) * 8;
  }
  // B staging: the block's fragments of a chunk are 2 groups x 8 KB ([tile 0..3][piece][lane] x 16 B, contiguous in the
  // pack); thread tid moves 16 bytes of each group. Through LDS, so that the four waves of a column do not each pull their
  // own copy through the L1 (64 KB per chunk and CU at 64 B/clk was the bound of the first version: 74 us)
  const char* bsrc = g.wpack + H2_HDR + (((size_t)grp0 * nt32 + (tile_n0 >> 5)) * 2) * 1024 + tid * 16;
  const size_t bgrp = (size_t)nt32 * 2 * 1024;  // bytes from one group's fragments to the next group's
  const int afrag = (wr * 64 + li) * LIN_PXB + 16 * lh;  // + gi * GROUP_B + p * PLANE_B + tm * 32 * PXB
  const int bfrag = LIN_BOFF + wc * 4096 + lane * 16;    // + buf * BBUF_B + gi * 8192 + tn * 2048 + p * 1024

  struct Stg { f32x4v a[4], b[2]; };
  auto load = [&](Stg& st, int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st.a[j] = *reinterpret_cast<const f32x4v*>(asrc[j] + (size_t)c * 32);
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) st.b[gi] = *reinterpret_cast<const f32x4v*>(bsrc + (size_t)(2 * c + gi) * bgrp);
  };
  auto store = [&](const Stg& st, int buf) {
    char* dst = smem + buf * LIN_ABUF_B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float X = st.a[j][e] * sa;
        const _Float16 hh = (_Float16)X;
        h[e] = hh;
        l[e] = (_Float16)(X - (float)hh);
      }
      *reinterpret_cast<f16x4_t*>(dst + adst[j]) = h;
      *reinterpret_cast<f16x4_t*>(dst + LIN_PLANE_B + adst[j]) = l;
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) *reinterpret_cast<f32x4v*>(smem + LIN_BOFF + buf * LIN_BBUF_B + gi * 8192 + tid * 16) = st.b[gi];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  auto compute = [&](int buf) {
    const char* A = smem + buf * LIN_ABUF_B + afrag;
    const char* Bp = smem + bfrag + buf * LIN_BBUF_B;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      f16x8_t a[2][2], b[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a[t][p] = *reinterpret_cast<const f16x8_t*>(A + gi * LIN_GROUP_B + p * LIN_PLANE_B + t * 32 * LIN_PXB);
          b[t][p] = *reinterpret_cast<const f16x8_t*>(Bp + gi * 8192 + t * 2048 + p * 1024);
        }
#define FL_PIECE(pa, pb)                                                                              \
      _Pragma("unroll") for (int tm = 0; tm < 2; ++tm) _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) \
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm][pa], b[tn][pb], acc[tm][tn], 0, 0, 0);
      FL_PIECE(1, 0) FL_PIECE(0, 1) FL_PIECE(0, 0)  // l*h, h*l, h*h: smallest terms first
#undef FL_PIECE
    }
  };

  // software pipeline over chunks: chunk c+3 is requested, chunk c+1 parked in LDS, chunk c contracted; three staging
  // register sets in rotation, two LDS buffers (period 6)
  Stg s0, s1, s2;
  load(s0, 0);
  if (nch > 1) load(s1, 1);
  if (nch > 2) load(s2, 2);
  store(s0, 0);
  __syncthreads();
#define FL_LIN_STEP(c, SNEXT, SFREE)                                        \
  do {                                                                      \
    if ((c) < nch) { /* block-uniform */                                    \
      if ((c) + 1 < nch) store(SNEXT, ((c) + 1) & 1);                       \
      if ((c) + 3 < nch) load(SFREE, (c) + 3);                              \
      compute((c) & 1);                                                     \
      __syncthreads();                                                      \
    }                                                                       \
  } while (0)
  // SNEXT holds chunk c+1; SFREE held chunk c (parked one step earlier) and takes chunk c+3
  for (int c = 0; c < nch; c += 3) {
    FL_LIN_STEP(c + 0, s1, s0);
    FL_LIN_STEP(c + 1, s2, s1);
    FL_LIN_STEP(c + 2, s0, s2);
  }
#undef FL_LIN_STEP

  // slab store (C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
  float* slab = g.slabs + (size_t)sp * g.M * g.N;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = tile_m0 + wr * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* yrow = slab + (size_t)m * g.N + tile_n0 + wc * 64 + li;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) yrow[tn * 32] = acc[tm][tn][e] * out_scale;
    }
  }
}

// y[(col / nb)][row][col % nb] = bias[col] + sum over the split-K slabs (nb = N: plain row-major y)
__global__ void linear_h2_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ y,
                                        int M, int N, int nb, int splits) {
  const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= (long long)M * N) return;
  const int row = (int)(i4 * 4 / N), col = (int)(i4 * 4 - (long long)row * N);
  float4 a = bias ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p = slabs + i4 * 4;
  for (int s = 0; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)s * M * N);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  *reinterpret_cast<float4*>(y + (size_t)(col / nb) * M * nb + (size_t)row * nb + col % nb) = a;
}

}  // namespace

namespace fl {

bool conv3x3_h2_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
  if (Cin % 32 || Cout % 128 || (reinterpret_cast<uintptr_t>(x) & 15) || H != W) return false;
  if ((long long)B * H * W * Cin >= (1ll << 31)) return false;
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && (W == 8 || W == 16 || W == 32)) return ((long long)B * H * W) % 128 == 0;
  return false;
}

bool conv4x4s2_h2_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
  if (Cin % 16 || Cout % 128 || (reinterpret_cast<uintptr_t>(x) & 15) || H != W) return false;  // halo loads take chunk PAIRS
  if ((long long)B * H * W * Cin >= (1ll << 31)) return false;
  if (kh == 4 && kw == 4 && stride == 2 && pad == 1 && (W == 16 || W == 32 || W == 64)) return ((long long)B * (H / 2) * (W / 2)) % 256 == 0;
  return false;
}

// bytes of the pre-op coefficient block ([scale | shift][images per tile][Cin]) behind the kernel's static LDS layout
static int h2_pre_bytes(int ksize, int Wout, int Cin, int bm = 256) {
  const int ni = (bm / Wout) > Wout ? (bm / Wout) / Wout : 1;
  return 2 * ni * Cin * (int)sizeof(float);
}

// the 3x3 convs with Cin % 64 == 0 run on the 16x16x32 pair kernel (weights packed in layout 2: fl_pack_weight_f16x2 and the launch
// apply the same rule)
#ifndef FL_H2_PAIRS
#define FL_H2_PAIRS 1  // 0: timing-only A/B build that keeps every 3x3 conv on the 32x32x16 kernel (tools/build_variant_lib.sh)
#endif
bool conv3x3_h2_pairs(int Cin) { return FL_H2_PAIRS && Cin % 64 == 0; }
// the 4x4 convs run on their 16x16x32 kernel (same pack layout 1) wherever its 48 KB weight ring + the pre-op coefficients fit the LDS
#ifndef FL_H2_PAIRS4
#define FL_H2_PAIRS4 FL_H2_PAIRS
#endif
bool conv4x4s2_h2_pairs(int lds_bytes) { return FL_H2_PAIRS4 && lds_bytes <= 160 * 1024; }

int conv_h2_lds_bytes(int ksize, int Win, int Cin, bool pre) {
  const int Wout = ksize == 3 ? Win : Win / 2;
  int base;
  if (ksize == 3 && conv3x3_h2_pairs(Cin)) base = Wout == 32 ? H2PGeom<32, 256>::LDS_B : (Wout == 16 ? H2PGeom<16, 256>::LDS_B : H2PGeom<8, 256>::LDS_B);
  else if (ksize == 3) base = Wout == 32 ? H2Geom<32>::LDS_B : (Wout == 16 ? H2Geom<16>::LDS_B : H2Geom<8>::LDS_B);
  else {
    const int pb = pre ? h2_pre_bytes(ksize, Wout, Cin) : 0;
    const int basep = Wout == 32 ? H2PGeom4<32>::LDS_B : (Wout == 16 ? H2PGeom4<16>::LDS_B : H2PGeom4<8>::LDS_B);
    if (conv4x4s2_h2_pairs(basep + pb)) return basep + pb;  // the 16x16x32 kernel where it fits
    base = Wout == 32 ? H2Geom4<32>::LDS_B : (Wout == 16 ? H2Geom4<16>::LDS_B : H2Geom4<8>::LDS_B);
  }
  return base + (pre ? h2_pre_bytes(ksize, Wout, Cin) : 0);
}

int launch_conv_h2(const float* x, const void* wpack, const float* bias, float* y, float* stats, const unsigned* a_amax,
                   int a_exp, const float* pre_scale, const float* pre_shift, int B, int H, int W, int Cin, int Cout, int ksize,
                   int act, float act_param, const H2Finish* fin, int* finished, hipStream_t s) {
  H2Args g;
  g.fin = 0;
  g.ln_w = g.ln_b = nullptr;
  g.scale = g.shift = g.mean = g.rstd = nullptr;
  g.eps = 0.f;
  if (finished) *finished = 0;
  g.pre_scale = pre_scale; g.pre_shift = pre_shift;
  g.Cin = Cin; g.Cout = Cout; g.tiles_n = Cout / 128; g.act = act; g.act_param = act_param;
  g.a_amax = a_amax; g.a_exp = a_exp;
  g.x = x; g.wpack = reinterpret_cast<const char*>(wpack); g.bias = bias; g.y = y; g.stats = stats;
  g.M = ksize == 3 ? B * H * W : B * (H / 2) * (W / 2);
  const int bm = ksize == 3 ? h2_bm(W, Cin, pre_scale != nullptr, g.M, Cout) : 256;
  const int pre_b = pre_scale ? h2_pre_bytes(ksize, ksize == 3 ? W : W / 2, Cin, bm) : 0;
  g.ntiles = (g.M / bm) * g.tiles_n;
  if (fin && stats) {
    const int HWo = ksize == 3 ? H * W : (H / 2) * (W / 2);
    g.fin = HWo == 64 ? 1 : (HWo == 256 && bm == 256 ? 2 : 0);
    if (g.fin) {
      g.ln_w = fin->ln_w; g.ln_b = fin->ln_b; g.eps = fin->eps;
      g.scale = fin->scale; g.shift = fin->shift; g.mean = fin->mean; g.rstd = fin->rstd;
      if (finished) *finished = 1;
    }
  }
  const int blocks = g.ntiles;
  static std::once_flag once;
  std::call_once(once, [] { h2_set_attr<32>(); h2_set_attr<16>(); h2_set_attr<8>(); });
  const int prio = g.pre_scale != nullptr;  // template PRE
#define FL_LAUNCH(WW)                                                                                                       \
  do {                                                                                                                      \
    if (bm == 256) {                                                                                                        \
      if (prio) hipLaunchKernelGGL((conv3x3_h2_kernel<WW, 1, 256>), dim3(blocks), dim3(512), (H2Geom<WW, 256>::LDS_B + pre_b), s, g); \
      else hipLaunchKernelGGL((conv3x3_h2_kernel<WW, 0, 256>), dim3(blocks), dim3(512), (H2Geom<WW, 256>::LDS_B + pre_b), s, g);      \
    } else {                                                                                                                \
      if (prio) hipLaunchKernelGGL((conv3x3_h2_kernel<WW, 1, 128>), dim3(blocks), dim3(256), (H2Geom<WW, 128>::LDS_B + pre_b), s, g); \
      else hipLaunchKernelGGL((conv3x3_h2_kernel<WW, 0, 128>), dim3(blocks), dim3(256), (H2Geom<WW, 128>::LDS_B + pre_b), s, g);      \
    }                                                                                                                       \
  } while (0)
#define FL_LAUNCHP(WW)                                                                                                      \
  do {                                                                                                                      \
    if (bm == 256) {                                                                                                        \
      if (prio) hipLaunchKernelGGL((conv3x3_h2p_kernel<WW, 1, 256>), dim3(blocks), dim3(512), (H2PGeom<WW, 256>::LDS_B + pre_b), s, g); \
      else hipLaunchKernelGGL((conv3x3_h2p_kernel<WW, 0, 256>), dim3(blocks), dim3(512), (H2PGeom<WW, 256>::LDS_B + pre_b), s, g);      \
    } else {                                                                                                                \
      if (prio) hipLaunchKernelGGL((conv3x3_h2p_kernel<WW, 1, 128>), dim3(blocks), dim3(256), (H2PGeom<WW, 128>::LDS_B + pre_b), s, g); \
      else hipLaunchKernelGGL((conv3x3_h2p_kernel<WW, 0, 128>), dim3(blocks), dim3(256), (H2PGeom<WW, 128>::LDS_B + pre_b), s, g);      \
    }                                                                                                                       \
  } while (0)
#define FL_LAUNCH4(WW)                                                                                               \
  do {                                                                                                             \
    if (conv4x4s2_h2_pairs(H2PGeom4<WW>::LDS_B + pre_b)) {                                                         \
      if (prio) hipLaunchKernelGGL((conv4x4s2_h2p_kernel<WW, 1>), dim3(blocks), dim3(512), H2PGeom4<WW>::LDS_B + pre_b, s, g);  \
      else hipLaunchKernelGGL((conv4x4s2_h2p_kernel<WW, 0>), dim3(blocks), dim3(512), H2PGeom4<WW>::LDS_B + pre_b, s, g);       \
    } else {                                                                                                       \
      if (prio) hipLaunchKernelGGL((conv4x4s2_h2_kernel<WW, 1>), dim3(blocks), dim3(512), H2Geom4<WW>::LDS_B + pre_b, s, g);  \
      else hipLaunchKernelGGL((conv4x4s2_h2_kernel<WW, 0>), dim3(blocks), dim3(512), H2Geom4<WW>::LDS_B + pre_b, s, g);       \
    }                                                                                                              \
  } while (0)
  if (ksize == 3 && conv3x3_h2_pairs(Cin)) {
    if (W == 32) FL_LAUNCHP(32);
    else if (W == 16) FL_LAUNCHP(16);
    else FL_LAUNCHP(8);
  } else if (ksize == 3) {
    if (W == 32) FL_LAUNCH(32);
    else if (W == 16) FL_LAUNCH(16);
    else FL_LAUNCH(8);
  } else {
    if (W == 64) FL_LAUNCH4(32);
    else if (W == 32) FL_LAUNCH4(16);
    else FL_LAUNCH4(8);
  }
#undef FL_LAUNCH
#undef FL_LAUNCHP
#undef FL_LAUNCH4
  return fl::check_launch("conv_h2_kernel");
}

size_t pack_weight_f16x2_bytes(int N, int K) { return (size_t)H2_HDR + (size_t)N * K * 4; }

int pack_weight_f16x2(const float* w, void* out, int N, int K, int layout, hipStream_t s) {
  if (hipMemsetAsync(out, 0, H2_HDR, s) != hipSuccess) return fl::check_launch("pack_weight_f16x2 memset");
  const long long n = (long long)N * K;
  int blocks = fl::cdiv(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(h2_amax_kernel, dim3(blocks), dim3(256), 0, s, w, n, reinterpret_cast<unsigned*>(out));
  const long long total = (long long)(K / 16) * (N / 32) * 64;
  hipLaunchKernelGGL(h2_pack_kernel, dim3(fl::cdiv(total, 256)), dim3(256), 0, s, w, reinterpret_cast<char*>(out), N, K, layout);
  return fl::check_launch("h2_pack_kernel");
}

// one atomic per block (memory-side atomics to one address serialise at ~25 ns each: a per-wave atomic from 2048 blocks
// cost 94 us on a 33 MB tensor); float4 loads, n % 4 handled by the scalar tail of block 0
__global__ __launch_bounds__(256) void h2_amax_kernel_pub(const float* __restrict__ x, long long n, unsigned* out) {
  __shared__ float red[4];
  float m = 0.f;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(x[i]));
  m = fl::wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));  // non-negative floats order like their bits
}

}  // namespace fl

// =====================================================================================
extern "C" {

size_t fl_pack_weight_f16x2_bytes(int N, int K) { return fl::pack_weight_f16x2_bytes(N, K); }

int fl_conv2d_f16x2_weight_layout(int kh, int Cin) { return kh == 4 ? 1 : (fl::conv3x3_h2_pairs(Cin) ? 2 : 0); }

// w: [N][K] fp32 (a packed conv weight [Cout][kh][kw][Cin]), N % 32 == 0, K % 16 == 0
int fl_pack_weight_f16x2(const float* w, void* out, int N, int K, int layout, fl_stream_t stream) {
  FL_REQUIRE(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "fl_pack_weight_f16x2: N=%d (mult. of 32), K=%d (mult. of 16)", N, K);
  FL_REQUIRE(layout == 0 || (layout == 1 && K % 128 == 0) || (layout == 2 && K % 576 == 0 && N % 16 == 0),
             "fl_pack_weight_f16x2: layout %d (0 = 16-k groups, 1 = 4x4 tap pairs, K = 16 Cin, Cin %% 8 == 0, 2 = 3x3 tap pairs of 32 k, K = 9 Cin, Cin %% 64 == 0)", layout);
  FL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0, "fl_pack_weight_f16x2: 16-byte alignment");
  return fl::pack_weight_f16x2(w, out, N, K, layout, reinterpret_cast<hipStream_t>(stream));
}

// splits: the largest divisor of the chunk count (32 k each) that keeps the grid within one block per CU
static int lin_h2_splits(int M, int N, int K) {
  const int tiles = (M / 256) * (N / 128), nchunks = K / 32;
  int best = 0;
  for (int s = 1; s <= nchunks; ++s) {
    if (nchunks % s) continue;
    if (tiles * s <= 256 || best == 0) best = s;
  }
  return best;
}

int fl_linear_fwd_f16x2_applicable(int M, int N, int K) {
  return M > 0 && M % 256 == 0 && N % 128 == 0 && K % 32 == 0 && lin_h2_splits(M, N, K) > 0;
}

size_t fl_linear_fwd_f16x2_workspace_bytes(int M, int N, int K) {
  if (!fl_linear_fwd_f16x2_applicable(M, N, K)) return 0;
  return (size_t)lin_h2_splits(M, N, K) * M * N * sizeof(float);
}

// y = x W^T + bias, W packed by fl_pack_weight_f16x2 ([N][K], layout 0). y is written as N / n_block consecutive [M][n_block]
// matrices (n_block = N: one row-major [M][N]); n_block % 4 == 0.
int fl_linear_fwd_f16x2(const float* x, int lda, const void* wpack, const float* bias, float* y, int n_block, int M, int N, int K,
                        const float* x_amax, int x_exp, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(x && wpack && y && fl_linear_fwd_f16x2_applicable(M, N, K), "fl_linear_fwd_f16x2: M=%d (mult. of 256) N=%d (mult. of 128) K=%d", M, N, K);
  FL_REQUIRE(lda >= K && lda % 4 == 0 && n_block > 0 && N % n_block == 0 && n_block % 4 == 0, "fl_linear_fwd_f16x2: lda=%d n_block=%d", lda, n_block);
  FL_REQUIRE(ws && ws_bytes >= fl_linear_fwd_f16x2_workspace_bytes(M, N, K), "fl_linear_fwd_f16x2: workspace too small");
  FL_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpack) | reinterpret_cast<uintptr_t>(y) |
               reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0, "fl_linear_fwd_f16x2: 16-byte alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  LinH2Args g;
  g.x = x; g.wpack = reinterpret_cast<const char*>(wpack); g.slabs = reinterpret_cast<float*>(ws);
  g.a_amax = reinterpret_cast<const unsigned*>(x_amax); g.a_exp = x_exp; g.lda = lda; g.M = M; g.N = N; g.K = K;
  g.tiles_n = N / 128; g.splits = lin_h2_splits(M, N, K); g.chunks = K / 32 / g.splits;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LIN_LDS_B);
  });
  hipLaunchKernelGGL(linear_h2_kernel, dim3((M / 256) * g.tiles_n * g.splits), dim3(512), LIN_LDS_B, s, g);
  const long long n4 = (long long)M * N / 4;
  hipLaunchKernelGGL(linear_h2_reduce_kernel, dim3(fl::cdiv(n4, 256)), dim3(256), 0, s, g.slabs, bias, y, M, N, n_block, g.splits);
  return fl::check_launch("linear_h2_kernel");
}

// out[0] = max|x| (a device float; 0 for n == 0): the range information fl_conv2d_fwd_f16x2 takes as x_amax when the
// producer of x keeps no bound (training activations and gradients). Order-independent (atomic max of non-negative floats).
int fl_absmax(const float* x, int64_t n, float* out, fl_stream_t stream) {
  FL_REQUIRE(out && (x || n == 0) && n >= 0, "fl_absmax: bad args");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, 4, s) != hipSuccess) return fl::check_launch("fl_absmax memset");
  if (n == 0) return FL_OK;
  FL_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "fl_absmax: x must be 16-byte aligned");
  int blocks = fl::cdiv(n, 256 * 16);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(fl::h2_amax_kernel_pub, dim3(blocks), dim3(256), 0, s, x, (long long)n, reinterpret_cast<unsigned*>(out));
  return fl::check_launch("fl_absmax");
}

int fl_conv2d_fwd_f16x2_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
  return (fl::conv3x3_h2_applicable(x, B, H, W, Cin, Cout, kh, kw, stride, pad) ||
          fl::conv4x4s2_h2_applicable(x, B, H, W, Cin, Cout, kh, kw, stride, pad)) ? 1 : 0;
}

int fl_conv2d_fwd_f16x2(const float* x, const void* wpack, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                        int kh, int kw, int stride, int pad, int act, float act_param, const float* ln_w, const float* ln_b,
                        float eps, float* scale, float* shift, float* mean, float* rstd, const float* x_amax, int x_exp,
                        const float* pre_scale, const float* pre_shift, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(x && wpack && y, "fl_conv2d_fwd_f16x2: null operand");
  FL_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr), "fl_conv2d_fwd_f16x2: pre_scale / pre_shift come together");
  FL_REQUIRE(fl::conv_h2_lds_bytes(kh, W, Cin, pre_scale != nullptr) <= 160 * 1024, "fl_conv2d_fwd_f16x2: Cin %d too large for the fused pre-op", Cin);
  FL_REQUIRE(fl_conv2d_fwd_f16x2_applicable(x, B, H, W, Cin, Cout, kh, kw, stride, pad),
             "fl_conv2d_fwd_f16x2: unsupported shape (B=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d s=%d p=%d)", B, H, W, Cin, Cout, kh, kw, stride, pad);
  FL_REQUIRE(x_exp >= -100 && x_exp <= 100, "fl_conv2d_fwd_f16x2: x_exp %d out of range", x_exp);
  const int Ho = kh == 3 ? H : H / 2, Wo = kh == 3 ? W : W / 2;
  const int HW = Ho * Wo;
  const bool stats = scale != nullptr;
  if (stats) {
    FL_REQUIRE(shift && act == FL_ACT_NONE && HW % 64 == 0, "fl_conv2d_fwd_f16x2: fused statistics need shift, act NONE, Ho*Wo %% 64 == 0");
    FL_REQUIRE(ws && ws_bytes >= fl_conv2d_fwd_instnorm_workspace_bytes(B, Ho, Wo, Cout), "fl_conv2d_fwd_f16x2: workspace too small");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* part = stats ? reinterpret_cast<float*>(ws) : nullptr;
  fl::H2Finish fin;
  fin.ln_w = ln_w; fin.ln_b = ln_b; fin.eps = eps; fin.scale = scale; fin.shift = shift; fin.mean = mean; fin.rstd = rstd;
  int finished = 0;
  const int rc = fl::launch_conv_h2(x, wpack, bias, y, part, reinterpret_cast<const unsigned*>(x_amax), x_exp, pre_scale, pre_shift, B, H,
                                    W, Cin, Cout, kh, act, act_param, stats ? &fin : nullptr, &finished, s);
  if (rc || !stats || finished) return rc;  // (finished: the conv kernel wrote scale / shift itself: its tiles hold whole images)
  return fl::launch_conv_inorm_finalize(part, ln_w, ln_b, eps, scale, shift, mean, rstd, B, HW, Cout, HW / 64, s);
}

}  // extern "C"
