// Residual-block tail of the VAE (v2:170-178): LayerNorm2d -> CALayer -> SpatialAttention
// -> (+residual) -> Swish, as two HBM-streaming passes over the conv output instead of the
// ~20 eager kernels the reference launches:
//   pass 1 (res_tail_map): u = s[b,c] * (y*scale[b,c] + shift[b,c]); per-pixel channel
//           mean and max of u -> 2-plane map (B,2,H,W)                 (v2:76-78)
//   pass 2 (res_tail_out): att = sigmoid(conv7x7(map)) ; out = swish(u*att + x)   (v2:79-81,176-177)
// NHWC makes the channel reduction a contiguous per-pixel read; a pixel is owned by a
// group of LPP = min(64, C/4) lanes (float4 per lane) and reduced with wave shuffles.
#include "fl_common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPP>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// u for 4 channels
__device__ __forceinline__ float4 tail_u(const float* y, const float* scale, const float* shift, const float* s,
                                         long long pix_off, long long bc_off, long long s_off) {
  float4 v = ld4(y + pix_off);
  if (scale) {
    const float4 sc = ld4(scale + bc_off), sh = ld4(shift + bc_off);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  }
  if (s) {
    const float4 ss = ld4(s + s_off);
    v.x *= ss.x; v.y *= ss.y; v.z *= ss.z; v.w *= ss.w;
  }
  return v;
}

template <int LPP>
__global__ void res_tail_map_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                    const float* __restrict__ shift, const float* __restrict__ s,
                                    float* __restrict__ map, int B, int HW, int C, int s_per_image) {
  constexpr int PPW = 64 / LPP;  // pixels per wave
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gl = lane % LPP, gp = lane / LPP;
  const long long npix = (long long)B * HW;
  const int C4 = C >> 2;
  const long long wave0 = ((long long)blockIdx.x * (blockDim.x >> 6) + wid) * PPW;
  const long long wstride = (long long)gridDim.x * (blockDim.x >> 6) * PPW;
  for (long long pw = wave0; pw < npix; pw += wstride) {
    const long long pix = pw + gp;
    const bool ok = pix < npix;
    const long long pp = ok ? pix : 0;
    const int b = (int)(pp / HW);
    float sum = 0.f, mx = -INFINITY;
    for (int c4 = gl; c4 < C4; c4 += LPP) {
      const float4 u = tail_u(y, scale, shift, s, pp * C + c4 * 4, (long long)b * C + c4 * 4, (s_per_image ? (long long)b * C : 0) + c4 * 4);
      sum += (u.x + u.y) + (u.z + u.w);
      mx = fmaxf(fmaxf(mx, fmaxf(u.x, u.y)), fmaxf(u.z, u.w));
    }
    sum = group_sum<LPP>(sum);
    mx = group_max<LPP>(mx);
    if (ok && gl == 0) {
      const int p = (int)(pp - (long long)b * HW);
      map[((long long)b * 2) * HW + p] = sum / C;
      map[((long long)b * 2 + 1) * HW + p] = mx;
    }
  }
}

template <int LPP>
__global__ void res_tail_out_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                    const float* __restrict__ shift, const float* __restrict__ s,
                                    const float* __restrict__ map, const float* __restrict__ saw,
                                    const float* __restrict__ xres, const float* __restrict__ xs,
                                    const float* __restrict__ xh, float* __restrict__ out, float* __restrict__ att_out,
                                    int B, int H, int W, int C, int final_swish, int s_per_image,
                                    const float* __restrict__ att_in) {
  constexpr int PPW = 64 / LPP;
  __shared__ float wsm[98];
  if (map) {
    for (int i = threadIdx.x; i < 98; i += blockDim.x) wsm[i] = saw[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gl = lane % LPP, gp = lane / LPP;
  const int HW = H * W;
  const long long npix = (long long)B * HW;
  const int C4 = C >> 2;
  const long long wave0 = ((long long)blockIdx.x * (blockDim.x >> 6) + wid) * PPW;
  const long long wstride = (long long)gridDim.x * (blockDim.x >> 6) * PPW;
  for (long long pw = wave0; pw < npix; pw += wstride) {
    const long long pix = pw + gp;
    const bool ok = pix < npix;
    const long long pp = ok ? pix : 0;
    const int b = (int)(pp / HW);
    const int p = (int)(pp - (long long)b * HW);
    float att = 1.f;
    if (att_in) {  // the spatial-attention gate was computed by sa_att_kernel: this pass only streams
      att = att_in[pp];
    } else if (map) {
      const int py = p / W, px = p - py * W;
      float a = 0.f;
      for (int t = gl; t < 98; t += LPP) {
        const int ch = t / 49, r = t - ch * 49;
        const int dy = r / 7 - 3, dx = r % 7 - 3;
        const int yy = py + dy, xx = px + dx;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
          a += wsm[t] * map[((long long)b * 2 + ch) * HW + yy * W + xx];
      }
      a = group_sum<LPP>(a);
      att = fl::sigmoid_acc(a);
      if (att_out && ok && gl == 0) att_out[pp] = att;
    }
    if (!ok) continue;
    for (int c4 = gl; c4 < C4; c4 += LPP) {
      const long long off = pp * C + c4 * 4;
      float4 u = tail_u(y, scale, shift, s, off, (long long)b * C + c4 * 4, (s_per_image ? (long long)b * C : 0) + c4 * 4);
      u.x *= att; u.y *= att; u.z *= att; u.w *= att;
      if (xres) {
        float4 r = ld4(xres + off);
        if (xs) {  // the residual before its LayerNorm2d-apply + Swish (the fused inference path: fast Swish, as in conv_h2.hip)
          const float4 a = ld4(xs + (long long)b * C + c4 * 4), d = ld4(xh + (long long)b * C + c4 * 4);
          r.x = fl::swish_fast(r.x * a.x + d.x); r.y = fl::swish_fast(r.y * a.y + d.y);
          r.z = fl::swish_fast(r.z * a.z + d.z); r.w = fl::swish_fast(r.w * a.w + d.w);
        }
        u.x += r.x; u.y += r.y; u.z += r.z; u.w += r.w;
      }
      if (final_swish) {
        if (xs) {  // fused inference path: the fast Swish everywhere (this pass is otherwise ALU-bound on expf)
          u.x = fl::swish_fast(u.x); u.y = fl::swish_fast(u.y); u.z = fl::swish_fast(u.z); u.w = fl::swish_fast(u.w);
        } else {
          u.x = fl::swish_acc(u.x); u.y = fl::swish_acc(u.y); u.z = fl::swish_acc(u.z); u.w = fl::swish_acc(u.w);
        }
      }
      st4(out + off, u);
    }
  }
}

// The streaming form of res_tail_out_kernel for the case the encoder runs it in (gate precomputed by sa_att_kernel, residual given):
// a lane group walks RUN CONSECUTIVE pixels of one image, so the five per-(image, channel) coefficient vectors are loaded once per
// run instead of once per pixel (they were 5 of the 7 loads of every pixel: cache hits, but each a full pass through the texture
// path), and the run's loads go out UNR pixels at a time before the first of them is used. Same arithmetic as res_tail_out_kernel
// (the two agree to 1-3 ulp: hipcc contracts the multiply-adds of the two bodies differently; tools/res_tail_equal.py).
// Stage 1 / 2 / 3 of the encoder at batch 256: gate + tail 97.6 -> 85.6 us at stage 1, encoder graph 2.236 -> 2.205 ms.
template <int LPP, int RUN>
__global__ __launch_bounds__(256) void res_tail_stream_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ s,
                                                              const float* __restrict__ xres, const float* __restrict__ xs,
                                                              const float* __restrict__ xh, float* __restrict__ out,
                                                              const float* __restrict__ att_in, long long npix, int HW, int C,
                                                              int final_swish, int s_per_image) {
  constexpr int PPW = 64 / LPP, UNR = 4;
  static_assert(RUN % UNR == 0, "run length");
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gl = lane % LPP, gp = lane / LPP;
  const int C4 = C >> 2;
  const long long pix0 = (((long long)blockIdx.x * (blockDim.x >> 6) + wid) * PPW + gp) * RUN;  // HW % RUN == 0: one image per run
  if (pix0 >= npix) return;
  const int b = (int)(pix0 / HW);
  for (int c4 = gl; c4 < C4; c4 += LPP) {
    const long long bc = (long long)b * C + c4 * 4;
    const float4 sc = scale ? ld4(scale + bc) : make_float4(1.f, 1.f, 1.f, 1.f), sh = scale ? ld4(shift + bc) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ss = s ? ld4(s + (s_per_image ? (long long)b * C : 0) + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 a = xs ? ld4(xs + bc) : make_float4(1.f, 1.f, 1.f, 1.f), d = xs ? ld4(xh + bc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int i0 = 0; i0 < RUN; i0 += UNR) {
      float4 v[UNR], r[UNR];
      float att[UNR];
#pragma unroll
      for (int i = 0; i < UNR; ++i) {
        const long long off = (pix0 + i0 + i) * C + c4 * 4;
        v[i] = ld4(y + off);
        r[i] = ld4(xres + off);
        att[i] = att_in[pix0 + i0 + i];
      }
#pragma unroll
      for (int i = 0; i < UNR; ++i) {
        float4 u = v[i];
        if (scale) { u.x = u.x * sc.x + sh.x; u.y = u.y * sc.y + sh.y; u.z = u.z * sc.z + sh.z; u.w = u.w * sc.w + sh.w; }
        if (s) { u.x *= ss.x; u.y *= ss.y; u.z *= ss.z; u.w *= ss.w; }
        u.x *= att[i]; u.y *= att[i]; u.z *= att[i]; u.w *= att[i];
        float4 q = r[i];
        if (xs) {
          q.x = fl::swish_fast(q.x * a.x + d.x); q.y = fl::swish_fast(q.y * a.y + d.y);
          q.z = fl::swish_fast(q.z * a.z + d.z); q.w = fl::swish_fast(q.w * a.w + d.w);
        }
        u.x += q.x; u.y += q.y; u.z += q.z; u.w += q.w;
        if (final_swish) {
          if (xs) {
            u.x = fl::swish_fast(u.x); u.y = fl::swish_fast(u.y); u.z = fl::swish_fast(u.z); u.w = fl::swish_fast(u.w);
          } else {
            u.x = fl::swish_acc(u.x); u.y = fl::swish_acc(u.y); u.z = fl::swish_acc(u.z); u.w = fl::swish_acc(u.w);
          }
        }
        st4(out + (pix0 + i0 + i) * C + c4 * 4, u);
      }
    }
  }
}

// att[b, p] = sigmoid(conv7x7([mean | max] map, saw)) (v2:76-79): one thread per pixel on the L2-resident 2-plane map.
// Inside res_tail_out_kernel the 98-tap stencil (lanes splitting the taps, a shuffle tree, the sigmoid) sat in front of every
// pixel's channel stream and cost 59 of its 135 us at the 32x32x128 stage (tools/res_tail_probe.py).
__global__ __launch_bounds__(256) void sa_att_kernel(const float* __restrict__ map, const float* __restrict__ saw,
                                                     float* __restrict__ att, int B, int H, int W) {
  // one image per block: its two map planes with a zero halo of 3 in LDS, then 98 unconditional LDS reads per pixel
  extern __shared__ float sm[];  // [98 weights][2][(H + 6)][(W + 6)]
  float* wsm = sm;
  float* pm = sm + 98;
  const int HW = H * W, Wp = W + 6, plane = (H + 6) * Wp;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < 98; i += blockDim.x) wsm[i] = saw[i];
  for (int i = threadIdx.x; i < 2 * plane; i += blockDim.x) {
    const int ch = i / plane, r = i - ch * plane;
    const int yy = r / Wp - 3, xx = r - (r / Wp) * Wp - 3;
    pm[i] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? map[((long long)b * 2 + ch) * HW + yy * W + xx] : 0.f;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int py = p / W, px = p - py * W;
    // 98 taps summed in fp64 (exact products, one rounding at the end): the gate's own gradient (v2:76-79 backward) is
    // ill-conditioned enough that the summation order of an fp32 stencil shows in it (tests/test_golden_r2_gpu.py)
    double a = 0.0;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const float* m = pm + ch * plane + py * Wp + px;  // window origin (py - 3, px - 3) in padded coordinates
#pragma unroll
      for (int dy = 0; dy < 7; ++dy)
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) a += (double)wsm[ch * 49 + dy * 7 + dx] * (double)m[dy * Wp + dx];
    }
    att[(long long)b * HW + p] = fl::sigmoid_acc((float)a);
  }
}

static int tail_blocks(long long npix, int ppw) {
  long long waves = (npix + ppw - 1) / ppw;
  long long blocks = (waves + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" {

// map (B,2,H*W): plane 0 = channel mean, plane 1 = channel max of u = s*(y*scale+shift)
// (scale/shift/s may be NULL = identity).  Replaces v2:76-78 (+ v2:151-155, v2:67 fused in).
int fl_res_tail_map(const float* y, const float* scale, const float* shift, const float* s, int s_per_image, float* map, int B, int HW,
                    int C, fl_stream_t stream) {
  FL_REQUIRE(y && map && C % 4 == 0, "fl_res_tail_map: bad args");
  FL_REQUIRE((scale == nullptr) == (shift == nullptr), "fl_res_tail_map: scale/shift must come together");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long npix = (long long)B * HW;
  if (npix == 0) return FL_OK;
  const int C4 = C / 4;
#define LAUNCH_MAP(L) hipLaunchKernelGGL(res_tail_map_kernel<L>, dim3(tail_blocks(npix, 64 / L)), dim3(256), 0, st, y, scale, shift, s, map, B, HW, C, s_per_image)
  // lanes per pixel: 8 (C = 128: four float4 per lane in flight, three shuffle steps per reduction) or 16 - one lane per float4 and a
  // five / six step shuffle tree per pixel pair made the pass shuffle-bound (3.6 TB/s on the 32x32x128 stage; its neighbour 5.2)
  if (C4 >= 64) LAUNCH_MAP(16);
  else if (C4 >= 8) LAUNCH_MAP(8);
  else LAUNCH_MAP(1);
#undef LAUNCH_MAP
  return fl::check_launch("res_tail_map_kernel");
}

// out = [swish]( u * sigmoid(conv7x7(map, saw)) + xres ), u as above. map NULL -> att = 1.
// att_out (B,H*W) optional (saved for backward). Replaces v2:79-81 and v2:176-177.
int fl_res_tail_out_pre(const float* y, const float* scale, const float* shift, const float* s, int s_per_image, const float* map,
                        const float* saw, const float* xres, const float* xres_scale, const float* xres_shift, float* out,
                        float* att_out, int B, int H, int W, int C, int final_swish, fl_stream_t stream) {
  FL_REQUIRE(y && out && C % 4 == 0, "fl_res_tail_out: bad args");
  FL_REQUIRE((xres_scale == nullptr) == (xres_shift == nullptr) && (!xres_scale || xres), "fl_res_tail_out_pre: xres_scale / xres_shift / xres");
  FL_REQUIRE((map == nullptr) || (saw != nullptr), "fl_res_tail_out: map needs conv weights");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long npix = (long long)B * H * W;
  if (npix == 0) return FL_OK;
  const int C4 = C / 4;
  const float* att_in = nullptr;
  if (map && att_out && (98 + 2 * (H + 6) * (W + 6)) * (int)sizeof(float) <= 48 * 1024) {  // gate first (its own small kernel, into the buffer the caller keeps anyway), then the streaming pass
    hipLaunchKernelGGL(sa_att_kernel, dim3(B), dim3(256), (98 + 2 * (H + 6) * (W + 6)) * sizeof(float), st, map, saw, att_out, B, H, W);
    att_in = att_out;
    att_out = nullptr;
    map = nullptr;
  }
  // the streaming form: gate precomputed, residual given, whole runs of 8 pixels inside an image (FL_TAIL_STREAM=0: the general kernel)
  static const bool stream_ok = !(getenv("FL_TAIL_STREAM") && getenv("FL_TAIL_STREAM")[0] == '0');
  if (stream_ok && att_in && xres && !map && !att_out && (H * W) % 8 == 0 && C4 >= 16) {
    const long long runs = npix / 8;
#define LAUNCH_STREAM(L) hipLaunchKernelGGL((res_tail_stream_kernel<L, 8>), dim3((unsigned)((runs + 4 * (64 / L) - 1) / (4 * (64 / L)))), dim3(256), 0, st, y, scale, shift, s, xres, xres_scale, xres_shift, out, att_in, npix, H * W, C, final_swish, s_per_image)
    if (C4 >= 64) LAUNCH_STREAM(64);
    else if (C4 >= 32) LAUNCH_STREAM(32);
    else LAUNCH_STREAM(16);
#undef LAUNCH_STREAM
    return fl::check_launch("res_tail_stream_kernel");
  }
#define LAUNCH_OUT(L) hipLaunchKernelGGL(res_tail_out_kernel<L>, dim3(tail_blocks(npix, 64 / L)), dim3(256), 0, st, y, scale, shift, s, map, saw, xres, xres_scale, xres_shift, out, att_out, B, H, W, C, final_swish, s_per_image, att_in)
  if (C4 >= 64) LAUNCH_OUT(64);
  else if (C4 >= 32) LAUNCH_OUT(32);
  else if (C4 >= 16) LAUNCH_OUT(16);
  else if (C4 >= 8) LAUNCH_OUT(8);
  else LAUNCH_OUT(1);
#undef LAUNCH_OUT
  return fl::check_launch("res_tail_out_kernel");
}

int fl_res_tail_out(const float* y, const float* scale, const float* shift, const float* s, int s_per_image, const float* map,
                    const float* saw, const float* xres, float* out, float* att_out, int B, int H, int W, int C,
                    int final_swish, fl_stream_t stream) {
  return fl_res_tail_out_pre(y, scale, shift, s, s_per_image, map, saw, xres, nullptr, nullptr, out, att_out, B, H, W, C, final_swish,
                             stream);
}

}  // extern "C"
