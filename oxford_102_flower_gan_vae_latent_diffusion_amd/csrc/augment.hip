// Training-image transform on the device: transform_train / transform_test of v2/model_train_test.py:24-35
// applied to a pre-resized uint8 NHWC image cache that lives in HBM (the 1020-image train split is 12.5 MB), replacing
// DataLoader(num_workers=2) + PIL at v2:1303. Byte/integer work, HBM-bound: 12 KB read + 48 KB written per image.
//
// One workgroup per image, 1024 lanes x 4 consecutive pixels (16 waves per CU hide the gather latency at B = 256 = one
// block per CU; each lane stores one float4 per channel plane); the whole 64x64x3 image stays in registers between the steps:
//   gather (h-flip folded into the rotation's nearest-neighbour inverse map, Pillow's 16.16 fixed point)
//   -> up to three ImageEnhance blends in the drawn order (contrast needs the image-wide grey mean: one block reduction)
//   -> ToTensor: CHW float32 = u8 / 255 (correctly rounded quotients from a 256-entry LDS table), coalesced per plane.
// Every step reproduces Pillow's integer / float32 arithmetic exactly (oracle/augment_oracle.py, pinned against Pillow).
#include "fl_common.h"

namespace {

constexpr int IMG = 64, PIX = IMG * IMG, THREADS = 1024, WAVES = THREADS / 64, PPT = PIX / THREADS, REC = 16;

__device__ __forceinline__ int grey_of(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Pillow Blend.c on one uint8 band: float32 multiply and add (no contraction), truncating store, clip only when extrapolating.
// (HIP's __fmul_rn/__fadd_rn are plain * and + and would be contracted into one fma: the pragma keeps two roundings.)
__device__ __forceinline__ int blend1(int in1, int in2, float alpha, bool inside) {
#pragma clang fp contract(off)
  const float prod = alpha * (float)(in2 - in1);
  const float t = (float)in1 + prod;
  if (inside) return (int)t;
  return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

__device__ __forceinline__ int block_sum_i32(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // `red` may still be read from a previous reduction
  if (lane == 0) red[wid] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(THREADS) void augment_kernel(const uint8_t* __restrict__ cache, const int64_t* __restrict__ labels,
                                                      const int64_t* __restrict__ index, const int32_t* __restrict__ params,
                                                      float* __restrict__ out, int64_t* __restrict__ labels_out, int N) {
  __shared__ int red[WAVES];
  __shared__ float unit[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 256) unit[tid] = __fdiv_rn((float)tid, 255.0f);  // ToTensor's u8/255, one IEEE division per level
  long long src = index ? index[b] : b;
  src = src < 0 ? 0 : (src >= N ? N - 1 : src);  // indices are device data: clamp rather than fault
  if (labels_out && tid == 0) labels_out[b] = labels[src];
  const uint8_t* img = cache + (size_t)src * PIX * 3;
  const int32_t* rec = params ? params + (size_t)b * REC : nullptr;
  const bool plain = !rec || rec[14] != 0;
  int a0 = 65536, a1 = 0, a2 = 32768, a3 = 0, a4 = 65536, a5 = 32768, flip = 0;
  if (!plain) {
    flip = rec[0];
    a0 = rec[1]; a1 = rec[2]; a2 = rec[3]; a3 = rec[4]; a4 = rec[5]; a5 = rec[6];
  }
  int r[PPT], g[PPT], bl[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = tid * PPT + k, y = p >> 6, x = p & 63;
    const int xin = (a2 + y * a1 + x * a0) >> 16, yin = (a5 + y * a4 + x * a3) >> 16;
    const bool ok = xin >= 0 && xin < IMG && yin >= 0 && yin < IMG;
    const int xs = flip ? IMG - 1 - xin : xin;
    const uint8_t* px = img + (ok ? (yin * IMG + xs) * 3 : 0);
    const int v0 = px[0], v1 = px[1], v2 = px[2];
    r[k] = ok ? v0 : 0; g[k] = ok ? v1 : 0; bl[k] = ok ? v2 : 0;
  }
  if (!plain) {
    for (int j = 0; j < 4; ++j) {
      const int op = rec[7 + j];
      if (op < 0 || op > 2) continue;  // 3 = hue: disabled in the reference's ColorJitter (v2:28)
      const float f = __int_as_float(rec[11 + op]);
      if (f == 1.0f) continue;         // blend returns the image itself
      const bool inside = f >= 0.0f && f <= 1.0f;
      int mean = 0;
      if (op == 1) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) s += grey_of(r[k], g[k], bl[k]);
        s = block_sum_i32(s, red);
        mean = (int)((double)s / (double)PIX + 0.5);
      }
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int d = op == 0 ? 0 : (op == 1 ? mean : grey_of(r[k], g[k], bl[k]));
        if (f == 0.0f) {
          r[k] = g[k] = bl[k] = d;
        } else {
          r[k] = blend1(d, r[k], f, inside);
          g[k] = blend1(d, g[k], f, inside);
          bl[k] = blend1(d, bl[k], f, inside);
        }
      }
    }
  }
  __syncthreads();  // unit[] is complete (block-uniform control flow above)
  static_assert(PPT == 4, "one float4 per lane and plane");
  float* o = out + (size_t)b * 3 * PIX + tid * PPT;
  *reinterpret_cast<float4*>(o) = make_float4(unit[r[0]], unit[r[1]], unit[r[2]], unit[r[3]]);
  *reinterpret_cast<float4*>(o + PIX) = make_float4(unit[g[0]], unit[g[1]], unit[g[2]], unit[g[3]]);
  *reinterpret_cast<float4*>(o + 2 * PIX) = make_float4(unit[bl[0]], unit[bl[1]], unit[bl[2]], unit[bl[3]]);
}

}  // namespace

int fl_augment_batch(const uint8_t* cache, const int64_t* labels, const int64_t* index, const int32_t* params, float* out,
                     int64_t* labels_out, int B, int N, int H, int W, fl_stream_t stream) {
  FL_REQUIRE(H == IMG && W == IMG, "fl_augment_batch: the image cache is %dx%d (v2:21), got %dx%d", IMG, IMG, H, W);
  FL_REQUIRE(B >= 0 && N > 0 && cache, "fl_augment_batch: bad args (B=%d N=%d)", B, N);
  if (B == 0) return FL_OK;
  FL_REQUIRE(out && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "fl_augment_batch: out must be a 16-byte aligned buffer");
  FL_REQUIRE(index || B <= N, "fl_augment_batch: B=%d images requested from a cache of %d without an index", B, N);
  FL_REQUIRE(!labels_out || labels, "fl_augment_batch: labels_out needs labels");
  hipLaunchKernelGGL(augment_kernel, dim3(B), dim3(THREADS), 0, reinterpret_cast<hipStream_t>(stream), cache, labels, index,
                     params, out, labels_out, N);
  return fl::check_launch("augment_kernel");
}
