// 3x3 / stride 1 / pad 1 convolution with pre-split weights, "halo tile" variant of the implicit-GEMM conv in gemm.hip
// (the six ResidualBlock convs of the frozen encoder, v2:159-178, two thirds of the DDPM step's matrix work).
//
// The generic kernel stages one [128 pixels x 32 channels] A tile per TAP: nine global->LDS round trips (loads, zero
// selects, ds_writes, a block barrier) per 32-channel chunk. Its timing ablations (profiles/r01_conv_kernel_pmc.md) put
// ~10 % of the kernel on that staging. Here the tile's input HALO ((rows+2) x (W+2) pixels x 32 channels, <= 29 KB) is
// staged ONCE per channel chunk and all nine taps read their MFMA A fragments from it at shifted LDS addresses (the tap
// shift is an immediate offset): one staging pass and one barrier per 9 x 48 MFMAs instead of per 48.
//
// Same math as gemm_kernel<.., B_FRAGS, .., SPLIT=6>: fp32 activations split exactly into three bf16 pieces in
// registers, weights pre-split once (fl_split_pack_weight, identical buffer), six piece products per k-group on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation; same 128x128 block / 64x64 wave tiling, same epilogue (bias, activation,
// row-major store, optional LayerNorm2d column sums per 64-row slab). Only the ORDER of the K loop differs
// (channel chunk outer, tap inner), i.e. fp32 re-association.
//
// Specialisation (host-checked): kh = kw = 3, stride 1, pad 1, H == W in {8, 16, 32}, Cin % 32 == 0, Cout % 128 == 0,
// B*H*W % 128 == 0. A 128-pixel tile is 128/W whole output rows (W = 32: 4 rows, 16: 8 rows) or two whole 8x8 images.
#include "fl_common.h"
#include <mutex>
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct Split3 {
  u32x4_t h, m, l;
};
// x = h + m + l exactly, pieces by truncation (see gemm.hip); 8 values = this lane's k-slots of one 16-k group
__device__ __forceinline__ Split3 split3(const float4& q0, const float4& q1) {
  const float x[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
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

struct HaloArgs {
  const float* x;
  const u32x4_t* wsplit;
  const float* bias;
  float* y;
  float* stats;
  int Cin, Cout, M, tiles_n, act;
  float act_param;
};

constexpr int LROW = 36;  // floats per staged pixel: 32 channels + 4 pad (conflict-free ds_read_b128 over consecutive pixels)

// dynamic LDS of conv3x3_halo_kernel<W> (must mirror the constants inside the kernel)
constexpr int halo3_lds_bytes(int W) {  // independent of TM
  const int R = 128 / W, RI = R < W ? R : W, NI = R / RI, HR = RI + 2, HC = W + 2;
  const int RPAD = W == 32 ? 0 : ((W - HC) * LROW % 64 + 64) % 64;
  return 2 * NI * HR * (HC * LROW + RPAD) * 4;
}

// TM = 2: 4 waves (2 x 2), wave tile 64 x 64, ~220 VGPRs, 2 blocks per CU - the fastest on an otherwise idle chip.
// TM = 1: 8 waves (4 x 2), wave tile 32 x 64, ~160 VGPRs, 1 block per CU - the same 2 waves per SIMD, but ~190 VGPRs
// per SIMD and > 90 KB of LDS stay free, so the waves of a concurrently running stream (the denoiser's small dependent
// kernels in the pipelined DDPM step) can start NEXT to the conv waves instead of waiting for a conv block to retire.
template <int W, int TM>
__global__ __launch_bounds__(TM == 2 ? 256 : 512, TM == 2 ? 2 : 1) void conv3x3_halo_kernel(const HaloArgs g) {
  constexpr int BM = 128, BN = 128, TN = 2, WM = 32 * TM, WN = 64;
  constexpr int NT = 64 * (BM / WM) * 2;    // threads per block
  constexpr int R = BM / W;                 // output rows per tile
  constexpr int RI = R < W ? R : W;         // ... per image
  constexpr int NI = R / RI;                // images per tile (2 for 8x8)
  constexpr int HR = RI + 2, HC = W + 2;    // halo rows / cols per image
  constexpr int P = NI * HR * HC;           // staged pixels
  constexpr int ITEMS = P * 8;              // float4 per chunk
  constexpr int NST = (ITEMS + NT - 1) / NT;  // float4 per thread per chunk
  // halo row pitch (floats). The 32 lanes of a fragment read are 32 consecutive output pixels; for W < 32 they span several
  // output rows, and the jump between rows must look like "the next W pixels" to the LDS banks (64 x 4 B), i.e.
  // RP == W * LROW (mod 64): then every lane hits the banks it would hit in the conflict-free W = 32 case (PMC: 46-63 %
  // of the LDS cycles were bank conflicts with the dense pitch HC * LROW).
  constexpr int RPAD = W == 32 ? 0 : ((W - HC) * LROW % 64 + 64) % 64;
  constexpr int RP = HC * LROW + RPAD;
  constexpr int TILE_F = NI * HR * RP;
  static_assert(2 * TILE_F * 4 <= 80 * 1024, "two halo buffers, two blocks per CU (160 KB of LDS)");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * TILE_F floats (halo3_lds_bytes; may exceed 64 KB)

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order (as gemm.hip): each XCD owns a contiguous range of tiles -> halo rows shared through its L2
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m0 = (tile / g.tiles_n) * BM;
  const int tile_n0 = (tile % g.tiles_n) * BN;
  const int img0 = tile_m0 / (W * W);
  const int oy0 = (tile_m0 - img0 * (W * W)) / W;  // 0 when the tile holds whole images
  const int Cin = g.Cin, nC = Cin >> 5;

  // ---- staging descriptors: item i = tid + 256 j -> (halo pixel i >> 3, float4 i & 7); invariant over channel chunks
  int goff[NST], sdst[NST];
  unsigned vmask = 0u;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i = tid + NT * j;
    const int pix = i >> 3, c4 = i & 7;
    const int nl = pix / (HR * HC), rem = pix - nl * (HR * HC);
    const int hy = rem / HC, hx = rem - hy * HC;
    const int iy = oy0 - 1 + hy, ix = hx - 1;
    const bool ok = (i < ITEMS) && ((unsigned)iy < (unsigned)W) && ((unsigned)ix < (unsigned)W);
    goff[j] = ok ? (((img0 + nl) * W + iy) * W + ix) * Cin + c4 * 4 : 0;
    sdst[j] = (nl * HR + hy) * RP + hx * LROW + c4 * 4;
    vmask |= (ok ? 1u : 0u) << j;
  }

  float4 stg[NST];
  auto stage_load = [&](int c) {
    const float* xc = g.x + c * 32;
#pragma unroll
    for (int j = 0; j < NST; ++j) stg[j] = *reinterpret_cast<const float4*>(xc + goff[j]);
  };
  auto stage_store = [&](int buf) {
    float* dst = smem + buf * TILE_F;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      if ((j + 1) * NT > ITEMS && tid + NT * j >= ITEMS) continue;  // only the last pass can run past the halo
      const bool ok = (vmask >> j) & 1u;
      float4 v = stg[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;  // padding ring
      *reinterpret_cast<float4*>(dst + sdst[j]) = v;
    }
  };

  // ---- A fragment addresses: tile row r -> halo pixel of tap (0,0); tap (dy,dx) adds dy*RP + dx*LROW floats
  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = wr * WM + tm * 32 + li;
    const int nl = r / (RI * W), rr = r - nl * (RI * W);
    const int oyl = rr / W, ox = rr - oyl * W;
    abase[tm] = (nl * HR + oyl) * RP + ox * LROW + 4 * lh;
  }
  // ---- B fragments: [16-k group][32-col tile][piece][lane] x 16 B, straight from global / L2
  const int nt32 = g.Cout >> 5;
  const u32x4_t* bbase = g.wsplit + (size_t)((tile_n0 + wc * WN) >> 5) * (3 * 64) + lane;
  const int gpt = Cin >> 4;  // 16-k groups per tap
  auto load_bp = [&](int g16, u32x4_t (&bp)[TN][3]) {
    const u32x4_t* b = bbase + (size_t)g16 * nt32 * (3 * 64);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[tn][p] = b[(tn * 3 + p) * 64];
  };
  // quarters 2h, 2h+1 of tap `tap` from LDS buffer S
  auto frag_load = [&](const float* S, int tapoff, int h, float4 (&f0)[TM], float4 (&f1)[TM]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float* p = S + abase[tm] + tapoff + 16 * h;
      f0[tm] = *reinterpret_cast<const float4*>(p);
      f1[tm] = *reinterpret_cast<const float4*>(p + 8);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // 24 MFMAs on pieces split earlier, interleaved 1 : 8 with the VALU split of the NEXT half-step's fragments
  auto mfma_and_split = [&](const Split3 (&sa)[TM], const u32x4_t (&bp)[TN][3], Split3 (&sn)[TM], const float4 (&n0)[TM],
                            const float4 (&n1)[TM]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) sn[tm] = split3(n0[tm], n1[tm]);
#define FL_PIECE(pa, pi)                                                                                       \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)         \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, sa[tm].pa),        \
                                                              __builtin_bit_cast(bf16x8_t, bp[tn][pi]), acc[tm][tn], 0, 0, 0);
    FL_PIECE(l, 0) FL_PIECE(h, 2) FL_PIECE(m, 1) FL_PIECE(m, 0) FL_PIECE(h, 1) FL_PIECE(h, 0)  // smallest terms first
#undef FL_PIECE
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
    for (int i = 0; i < 6 * TM * TN - 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };

  Split3 saA[TM], saB[TM];
  u32x4_t bpA[TN][3], bpB[TN][3];
  float4 f0[TM], f1[TM];

  // ---- prologue: chunk 0 staged, first fragments split
  stage_load(0);
  load_bp(0, bpA);
  stage_store(0);
  __syncthreads();
  frag_load(smem, 0, 0, f0, f1);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) saA[tm] = split3(f0[tm], f1[tm]);

  // one tap = two half-steps. STAGE: the next chunk's halo loads are issued right AFTER this tap's first fragment
  // loads - vmcnt retires in order, so a long-latency load issued BEFORE them would be waited for at once; issued
  // after, it has a whole half-step (24 MFMAs) before the next wait reaches it.
  auto tap_body = [&](auto stage_tag, int tap, const float* S, int tapoff, int tapnext, int gc, int cn) {
    // half 0 runs on (saA, bpA); prefetch half 1 of this tap
    load_bp(tap * gpt + gc + 1, bpB);
    frag_load(S, tapoff, 1, f0, f1);
    if constexpr (decltype(stage_tag)::value) stage_load(cn);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saA, bpA, saB, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    // half 1 runs on (saB, bpB); prefetch half 0 of the next tap
    load_bp((tap + 1) * gpt + gc, bpA);
    frag_load(S, tapnext, 0, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saB, bpB, saA, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int c = 0; c < nC; ++c) {
    const float* S = smem + (c & 1) * TILE_F;
    const int cn = c + 1 < nC ? c + 1 : c;  // clamped: the last chunk re-stages itself (never read)
    const int gc = 2 * c;                    // first 16-k group of this chunk inside a tap
    tap_body(std::true_type{}, 0, S, 0, LROW, gc, cn);  // next halo in flight under the remaining 8 taps
    int tapoff = LROW;
#pragma unroll 1
    for (int tap = 1; tap < 8; ++tap) {
      const int dyn = (tap + 1) / 3, dxn = (tap + 1) - 3 * dyn;
      const int tapnext = dyn * RP + dxn * LROW;
      tap_body(std::false_type{}, tap, S, tapoff, tapnext, gc, cn);
      tapoff = tapnext;
    }
    // tap 8 (dy = dx = 2): half 0 as above
    load_bp(8 * gpt + gc + 1, bpB);
    frag_load(S, tapoff, 1, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saA, bpA, saB, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    // half 1: the next chunk's halo goes to the other buffer (last read before the previous barrier), one barrier,
    // then the first fragments of the next chunk are prefetched under this chunk's last 24 MFMAs
    stage_store((c + 1) & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_bp(2 * cn, bpA);
    frag_load(smem + ((c + 1) & 1) * TILE_F, 0, 0, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saB, bpB, saA, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
  float st1[TN], st2[TN], bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    st1[tn] = st2[tn] = 0.f;
    bv[tn] = g.bias ? g.bias[tile_n0 + wc * WN + tn * 32 + li] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = tile_m0 + wr * WM + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* yrow = g.y + (size_t)m * g.Cout + tile_n0 + wc * WN + li;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const float v = fl::apply_act(acc[tm][tn][e] + bv[tn], g.act, g.act_param);
        yrow[tn * 32] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  // fused LayerNorm2d statistics: this wave's 64 rows lie in one image (H*W % 64 == 0): one slab of (sum y, sum y^2)
  if (g.stats) {
    const int grp = (tile_m0 + wr * WM) / WM;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const float a = st1[tn] + __shfl_xor(st1[tn], 32, 64);
      const float b = st2[tn] + __shfl_xor(st2[tn], 32, 64);
      const int col = tile_n0 + wc * WN + tn * 32 + li;
      if (lh == 0) {
        g.stats[((size_t)grp * 2) * g.Cout + col] = a;
        g.stats[((size_t)grp * 2 + 1) * g.Cout + col] = b;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// 4x4 / stride 2 / pad 1 variant (the three down-sampling convs of the encoder, v2:191-213). The input halo of a
// 128-output tile is (2 rows + 2) x (2 Wo + 2) pixels - ~4x the 3x3 case - so it is staged per SIXTEEN channels
// ([pixel][16 + 4 pad] floats, even / odd columns in separate planes, <= 60 KB, single buffer, 2 blocks per CU): one 16-k group per tap and chunk, sixteen taps
// = sixteen half-steps between stagings. With one buffer a chunk boundary needs two barriers (all reads done -> store ->
// stores visible); the next chunk's global loads are still in flight under the whole chunk.
constexpr int LROW2 = 20;

template <int WO, int TM>  // TM as in conv3x3_halo_kernel: 2 = 4 waves x (64 x 64), 1 = 8 waves x (32 x 64), one block per CU
__global__ __launch_bounds__(TM == 2 ? 256 : 512, TM == 2 ? 2 : 1) void conv4x4s2_halo_kernel(const HaloArgs g) {
  constexpr int BM = 128, BN = 128, TN = 2, WM = 32 * TM, WN = 64;
  constexpr int NT = 64 * (BM / WM) * 2;      // threads per block
  constexpr int W = 2 * WO;                   // input height = width
  constexpr int R = BM / WO;                  // output rows per tile
  constexpr int RI = R < WO ? R : WO;         // ... per image
  constexpr int NI = R / RI;
  constexpr int HR = 2 * RI + 2, HC = 2 * WO + 2;
  constexpr int P = NI * HR * HC;
  constexpr int ITEMS = P * 4;                // float4 per 16-channel chunk
  constexpr int NST = (ITEMS + NT - 1) / NT;
  // LDS image: even and odd input columns in separate planes, so that the 32 lanes of a fragment read (32 consecutive
  // output pixels = input columns 2 ox + dx) are CONSECUTIVE pixels of one plane: pitch 20 floats = an odd multiple of
  // 4 banks -> 16 lanes cover the 64 banks exactly once. Row pitch RP2 makes the jump to the next output row (two
  // input rows) look like "the next WO pixels" to the banks (2 RP2 == WO * 20 mod 64), as in the 3x3 kernel.
  constexpr int HCP = WO + 1;                 // pixels per plane row
  constexpr int RP2 = WO == 32 ? HCP * LROW2 : (WO == 16 ? 352 : 208);
  static_assert(RP2 >= HCP * LROW2 && (WO == 32 || (2 * RP2 - WO * LROW2) % 64 == 0), "row pitch");
  constexpr int PLANE_F = NI * HR * RP2;
  constexpr int TILE_F = 2 * PLANE_F;
  static_assert(TILE_F * 4 <= 64 * 1024, "the halo buffer fits the static LDS limit");
  __shared__ __attribute__((aligned(16))) float smem[TILE_F];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
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
  const int Cin = g.Cin, nC = Cin >> 4;

  int goff[NST], sdst[NST];
  unsigned vmask = 0u;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i = tid + NT * j;
    const int pix = i >> 2, c4 = i & 3;
    const int nl = pix / (HR * HC), rem = pix - nl * (HR * HC);
    const int hy = rem / HC, hx = rem - hy * HC;
    const int iy = 2 * oy0 - 1 + hy, ix = hx - 1;
    const bool ok = (i < ITEMS) && ((unsigned)iy < (unsigned)W) && ((unsigned)ix < (unsigned)W);
    goff[j] = ok ? (((img0 + nl) * W + iy) * W + ix) * Cin + c4 * 4 : 0;
    sdst[j] = (hx & 1) * PLANE_F + (nl * HR + hy) * RP2 + (hx >> 1) * LROW2 + c4 * 4;
    vmask |= (ok ? 1u : 0u) << j;
  }

  float4 stg[NST];
  auto stage_load = [&](int c) {
    const float* xc = g.x + c * 16;
#pragma unroll
    for (int j = 0; j < NST; ++j) stg[j] = *reinterpret_cast<const float4*>(xc + goff[j]);
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      if ((j + 1) * NT > ITEMS && tid + NT * j >= ITEMS) continue;
      const bool ok = (vmask >> j) & 1u;
      float4 v = stg[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(smem + sdst[j]) = v;
    }
  };

  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = wr * WM + tm * 32 + li;
    const int nl = r / (RI * WO), rr = r - nl * (RI * WO);
    const int oyl = rr / WO, ox = rr - oyl * WO;
    abase[tm] = (nl * HR + 2 * oyl) * RP2 + ox * LROW2 + 4 * lh;  // tap (dy, dx): + (dx & 1) PLANE_F + dy RP2 + (dx >> 1) LROW2
  }
  const int nt32 = g.Cout >> 5;
  const u32x4_t* bbase = g.wsplit + (size_t)((tile_n0 + wc * WN) >> 5) * (3 * 64) + lane;
  const int gpt = Cin >> 4;
  auto load_bp = [&](int g16, u32x4_t (&bp)[TN][3]) {
    const u32x4_t* b = bbase + (size_t)g16 * nt32 * (3 * 64);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[tn][p] = b[(tn * 3 + p) * 64];
  };
  auto frag_load = [&](int tapoff, float4 (&f0)[TM], float4 (&f1)[TM]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float* p = smem + abase[tm] + tapoff;
      f0[tm] = *reinterpret_cast<const float4*>(p);
      f1[tm] = *reinterpret_cast<const float4*>(p + 8);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  auto mfma_and_split = [&](const Split3 (&sa)[TM], const u32x4_t (&bp)[TN][3], Split3 (&sn)[TM], const float4 (&n0)[TM],
                            const float4 (&n1)[TM]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) sn[tm] = split3(n0[tm], n1[tm]);
#define FL_PIECE(pa, pi)                                                                                       \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)         \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, sa[tm].pa),        \
                                                              __builtin_bit_cast(bf16x8_t, bp[tn][pi]), acc[tm][tn], 0, 0, 0);
    FL_PIECE(l, 0) FL_PIECE(h, 2) FL_PIECE(m, 1) FL_PIECE(m, 0) FL_PIECE(h, 1) FL_PIECE(h, 0)
#undef FL_PIECE
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
    for (int i = 0; i < 6 * TM * TN - 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };

  Split3 saA[TM], saB[TM];
  u32x4_t bpA[TN][3], bpB[TN][3];
  float4 f0[TM], f1[TM];

  stage_load(0);
  load_bp(0, bpA);
  stage_store();
  __syncthreads();
  frag_load(0, f0, f1);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) saA[tm] = split3(f0[tm], f1[tm]);

  // taps 2 tp (on saA, bpA) and 2 tp + 1 (on saB, bpB): tap t = (dy, dx) = (t >> 2, t & 3) -> 2 tp = (tp >> 1, 2 (tp & 1))
  auto pair_body = [&](auto stage_tag, int tp, int c, int cn) {
    const int off0 = (tp >> 1) * RP2 + (tp & 1) * LROW2;               // tap 2 tp: even plane
    const int tn2 = tp + 1, off2 = (tn2 >> 1) * RP2 + (tn2 & 1) * LROW2;
    load_bp((2 * tp + 1) * gpt + c, bpB);
    frag_load(off0 + PLANE_F, f0, f1);                                 // tap 2 tp + 1: same shift, odd plane
    if constexpr (decltype(stage_tag)::value) stage_load(cn);  // after the fragment loads: see tap_body above
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saA, bpA, saB, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    load_bp((2 * tp + 2) * gpt + c, bpA);
    frag_load(off2, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saB, bpB, saA, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int c = 0; c < nC; ++c) {
    const int cn = c + 1 < nC ? c + 1 : c;
    pair_body(std::true_type{}, 0, c, cn);
#pragma unroll 1
    for (int tp = 1; tp < 7; ++tp) pair_body(std::false_type{}, tp, c, cn);
    // taps 14, 15
    load_bp(15 * gpt + c, bpB);
    frag_load(PLANE_F + 3 * RP2 + LROW2, f0, f1);  // tap 15 = (3, 3): odd plane, shift 1
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saA, bpA, saB, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    // every read of this chunk has been issued: drain, let everyone finish, replace the halo, publish it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage_store();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_bp(cn, bpA);
    frag_load(0, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_and_split(saB, bpB, saA, f0, f1);
    __builtin_amdgcn_sched_barrier(0);
  }

  float st1[TN], st2[TN], bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    st1[tn] = st2[tn] = 0.f;
    bv[tn] = g.bias ? g.bias[tile_n0 + wc * WN + tn * 32 + li] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = tile_m0 + wr * WM + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* yrow = g.y + (size_t)m * g.Cout + tile_n0 + wc * WN + li;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const float v = fl::apply_act(acc[tm][tn][e] + bv[tn], g.act, g.act_param);
        yrow[tn * 32] = v;
        st1[tn] += v;
        st2[tn] += v * v;
      }
    }
  }
  if (g.stats) {
    const int grp = (tile_m0 + wr * WM) / WM;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const float a = st1[tn] + __shfl_xor(st1[tn], 32, 64);
      const float b = st2[tn] + __shfl_xor(st2[tn], 32, 64);
      const int col = tile_n0 + wc * WN + tn * 32 + li;
      if (lh == 0) {
        g.stats[((size_t)grp * 2) * g.Cout + col] = a;
        g.stats[((size_t)grp * 2 + 1) * g.Cout + col] = b;
      }
    }
  }
}

}  // namespace

namespace fl {

bool conv3x3_halo_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
  if (Cin % 32 || Cout % 128 || (reinterpret_cast<uintptr_t>(x) & 15) || H != W) return false;
  if ((long long)B * H * W * Cin >= (1ll << 31)) return false;
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && (W == 8 || W == 16 || W == 32)) return ((long long)B * H * W) % 128 == 0;
  if (kh == 4 && kw == 4 && stride == 2 && pad == 1 && (W == 16 || W == 32 || W == 64)) return ((long long)B * (H / 2) * (W / 2)) % 128 == 0;
  return false;
}

// 3x3 s1 p1 (W in 8/16/32) or 4x4 s2 p1 (input W in 16/32/64); the caller checked conv3x3_halo_applicable
int launch_conv3x3_halo(const float* x, const void* wsplit, const float* bias, float* y, float* stats, int B, int H, int W,
                        int Cin, int Cout, int act, float act_param, hipStream_t s) {
  return launch_conv_halo(x, wsplit, bias, y, stats, B, H, W, Cin, Cout, 3, act, act_param, 64, s);
}

// rows per statistics slab (= rows of a wave tile) the halo kernels use in `mode` (0: 64-row wave tiles, 1: the
// co-residency-friendly 32-row ones)
int conv_halo_slab_rows(int ksize, int mode) { return (mode == 2 || (mode == 1 && ksize == 3)) ? 32 : 64; }  // 1: 3x3 only, 2: both

int launch_conv_halo(const float* x, const void* wsplit, const float* bias, float* y, float* stats, int B, int H, int W,
                     int Cin, int Cout, int ksize, int act, float act_param, int slab_rows, hipStream_t s) {
  HaloArgs g;
  g.x = x; g.wsplit = reinterpret_cast<const u32x4_t*>(wsplit); g.bias = bias; g.y = y; g.stats = stats;
  g.Cin = Cin; g.Cout = Cout; g.tiles_n = Cout / 128; g.act = act; g.act_param = act_param;
  if (ksize == 3) {
    g.M = B * H * W;
    const int blocks = (g.M / 128) * g.tiles_n;
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, halo3_lds_bytes(32));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, halo3_lds_bytes(16));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, halo3_lds_bytes(8));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, halo3_lds_bytes(32));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, halo3_lds_bytes(16));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, halo3_lds_bytes(8));
    });
    if (slab_rows == 32) {
      if (W == 32) hipLaunchKernelGGL((conv3x3_halo_kernel<32, 1>), dim3(blocks), dim3(512), halo3_lds_bytes(32), s, g);
      else if (W == 16) hipLaunchKernelGGL((conv3x3_halo_kernel<16, 1>), dim3(blocks), dim3(512), halo3_lds_bytes(16), s, g);
      else hipLaunchKernelGGL((conv3x3_halo_kernel<8, 1>), dim3(blocks), dim3(512), halo3_lds_bytes(8), s, g);
    } else {
      if (W == 32) hipLaunchKernelGGL((conv3x3_halo_kernel<32, 2>), dim3(blocks), dim3(256), halo3_lds_bytes(32), s, g);
      else if (W == 16) hipLaunchKernelGGL((conv3x3_halo_kernel<16, 2>), dim3(blocks), dim3(256), halo3_lds_bytes(16), s, g);
      else hipLaunchKernelGGL((conv3x3_halo_kernel<8, 2>), dim3(blocks), dim3(256), halo3_lds_bytes(8), s, g);
    }
  } else {
    const int Wo = W / 2;
    g.M = B * Wo * Wo;
    const int blocks = (g.M / 128) * g.tiles_n;
    if (slab_rows == 32) {
      if (Wo == 32) hipLaunchKernelGGL((conv4x4s2_halo_kernel<32, 1>), dim3(blocks), dim3(512), 0, s, g);
      else if (Wo == 16) hipLaunchKernelGGL((conv4x4s2_halo_kernel<16, 1>), dim3(blocks), dim3(512), 0, s, g);
      else hipLaunchKernelGGL((conv4x4s2_halo_kernel<8, 1>), dim3(blocks), dim3(512), 0, s, g);
    } else {
      if (Wo == 32) hipLaunchKernelGGL((conv4x4s2_halo_kernel<32, 2>), dim3(blocks), dim3(256), 0, s, g);
      else if (Wo == 16) hipLaunchKernelGGL((conv4x4s2_halo_kernel<16, 2>), dim3(blocks), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((conv4x4s2_halo_kernel<8, 2>), dim3(blocks), dim3(256), 0, s, g);
    }
  }
  return fl::check_launch("conv_halo_kernel");
}

}  // namespace fl
