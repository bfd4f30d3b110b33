/*
 * flowers_hip.h — C ABI of libflowers_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary underneath the v2 nn.Module surface of
 * ynyeh0221/Oxford-102-Flower-GAN-VAE-latent-diffusion. The
 * reference has no FFI of its own: every entry point below replaces the ATen
 * ops that a given reference line executes; the line it replaces is cited as
 * `v2:N` = /root/reference/v2/model_train_test.py:N.
 *
 * Conventions (all entry points):
 *   - plain device pointers (tensor.data_ptr()), sizes as ints; no torch types;
 *   - the caller owns every buffer incl. the workspace `ws`; the library never
 *     allocates or frees device memory;
 *   - all work is enqueued on `stream`; no implicit synchronisation, no
 *     default-stream use (hipGraph-capturable);
 *   - return FL_OK (0) or a negative fl_status; fl_last_error_string() has text;
 *   - activations are fp32, NHWC ("channels-last") in HBM; 3-channel images are
 *     stored as NHWC with C padded to 4 (pad channel = 0);
 *   - conv weights are "packed": [Cout][kh][kw][Cin] fp32 (fl_pack_conv_weight).
 */
#ifndef FLOWERS_HIP_H
#define FLOWERS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* fl_stream_t; /* == hipStream_t */

typedef enum fl_status {
  FL_OK = 0,
  FL_ERR_BAD_ARG = -1,
  FL_ERR_WORKSPACE = -2,
  FL_ERR_LAUNCH = -3,
  FL_ERR_UNSUPPORTED = -4
} fl_status;

enum { FL_ACT_NONE = 0, FL_ACT_SWISH = 1, FL_ACT_RELU = 2, FL_ACT_LRELU = 3, FL_ACT_SIGMOID = 4 };
/* operand layouts for fl_gemm_f32: KC = reduction index contiguous, KS = strided */
enum { FL_LAYOUT_KC = 0, FL_LAYOUT_KS = 1 };

int fl_version(void);
const char* fl_last_error_string(void);



/* ------------------------------------------------------------------------ *
 * Dense contraction core (fp32-in / fp32-accumulate MFMA, v_mfma_f32_32x32x2_f32)
 *   C[m][n] = epi( sum_k A(m,k) * B(n,k) )
 *   A: KC -> A[m*lda + k]   KS -> A[k*lda + m]
 *   B: KC -> B[n*ldb + k]   KS -> B[k*ldb + n]
 *   epi(v) : v += bias_scale*bias[n]; if (R && r_pre) v += r_scale*R[m*ldr+n];
 *            v = act(v); if (R && !r_pre) v += r_scale*R[m*ldr+n]
 * Replaces nn.Linear forward/backward (v2:211-220, v2:246-252, v2:311-320,
 * v2:405-431, v2:509-531) and, through the conv entry points, nn.Conv2d /
 * nn.ConvTranspose2d. K must be a multiple of 4; lda/ldb multiples of 4.
 * Split-K is used automatically when the tile count is small and `ws` allows.
 * ------------------------------------------------------------------------ */
size_t fl_gemm_workspace_bytes(int M, int N, int K);
/* How the GEMM core multiplies fp32 operands (all GEMM / conv entry points; set before capturing graphs):
 *   0  v_mfma_f32_32x32x2_f32 on the fp32 matrix path (157 TFLOP/s peak);
 *   1  (default; env FL_GEMM_MATH=fp32 selects 0) every operand is split EXACTLY into three bf16 pieces
 *      x = h + m + l and the product is assembled from 6 of the 9 piece products on v_mfma_f32_32x32x16_bf16
 *      (piece products exact, fp32 accumulation); dropped terms m*l, l*m, l*l < 2^-21 |x*y| per product, below the
 *      rounding noise of an fp32 accumulation over K >= 64 terms. Unaligned operands always take mode 0. */
int fl_set_gemm_math(int mode);
int fl_get_gemm_math(void);
/* Split-K policy of the GEMM launcher: a GEMM whose grid has fewer tiles than `tiles` is split along K (+ one reduce
 * launch) so that it fills the chip. Default 160. 0 disables split-K. A stream that shares the chip with another one
 * (the denoiser next to the frozen encoder, steps.DDPMTrainer.capture_pipelined) lowers it while it captures its graph:
 * results are the same up to fp32 summation order. Process-wide, read at launch time. */
int fl_set_gemm_splitk_tiles(int tiles);
int fl_get_gemm_splitk_tiles(void);
/* Variant of the 3x3 halo-tile conv kernel behind fl_conv2d_fwd_presplit. 0 (default): 4 waves with 64x64 wave tiles,
 * ~220 VGPRs, two blocks per CU - fastest on an otherwise idle chip. 1: 8 waves with 32x64 wave tiles, ~160 VGPRs, one
 * block per CU - the same two conv waves per SIMD, but registers and LDS are left for the waves of a concurrently running
 * stream (the denoiser next to the frozen encoder in the pipelined DDPM step). Same results up to fp32 summation order
 * of the fused LayerNorm2d statistics (32- instead of 64-row slabs). Process-wide, read at launch time. */
int fl_set_conv_halo_mode(int mode);
int fl_get_conv_halo_mode(void);
int fl_gemm_f32(const float* A, int64_t lda, int a_layout,
                const float* B, int64_t ldb, int b_layout,
                float* C, int64_t ldc, int M, int N, int K,
                const float* bias, float bias_scale,
                const float* R, int64_t ldr, float r_scale, int r_pre,
                int act, float act_param,
                void* ws, size_t ws_bytes, fl_stream_t stream);

/* Implicit-GEMM NHWC convolution, y = act(conv(x, w) + bias) (+ residual R, same
 * shape as y).  x:(B,H,W,Cin) wp:[Cout][kh][kw][Cin]  y:(B,Ho,Wo,Cout) or, when
 * out_nchw != 0, (B,Cout,Ho,Wo).  Cin % 4 == 0.
 * Replaces nn.Conv2d forward: v2:162,165 (3x3 s1 p1), v2:186, v2:192,198,204
 * (4x4 s2 p1), v2:273,276, VGG16 convs (v2:965), Discriminator64 (v2:987-1002).
 * INPUT RANGE of the RGB stem (3 -> 64 channels, 3x3 s1 p1, Cin padded to 4) in the default "split" GEMM math: that shape runs
 * on the f16x2 stem kernel (conv_first.hip), which scales the taps by a fixed 2^6 before splitting them into fp16 pieces: it is
 * exact-to-fp32 for |x| < 1023 (images, normalised or not, are far inside) and returns NaN - loudly, never a wrong number - for
 * larger inputs. FL_STEM_FP32=1 or fl_set_gemm_math(FL_GEMM_FP32) select the fp32-MFMA stem, which has no range limit.
 * The same holds for fl_conv2d_fwd_instnorm. Every other shape takes x_amax / x_exp explicitly (fl_conv2d_fwd_f16x2). */
int fl_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y,
                  int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                  int act, float act_param, const float* R, int out_nchw,
                  void* ws, size_t ws_bytes, fl_stream_t stream);
/* Same convolution with the LayerNorm2d statistics of its output (v2:152-153) accumulated in the GEMM epilogue:
 * emits y plus per-(image,channel) scale/shift/mean/rstd (norm(y) = y*scale + shift) — no separate statistics pass. */
size_t fl_conv2d_fwd_instnorm_workspace_bytes(int B, int Ho, int Wo, int Cout);
int fl_conv2d_fwd_instnorm(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                           int Cout, int kh, int kw, int stride, int pad, const float* ln_w, const float* ln_b,
                           float eps, float* scale, float* shift, float* mean, float* rstd, void* ws, size_t ws_bytes,
                           fl_stream_t stream);
/* Forward conv with a weight that was split into its three bf16 pieces once (fl_split_pack_weight of the packed
 * [Cout][kh*kw*Cin] weight; fl_split_pack_weight_bytes(N,K) = 6*N*K): the weight fragments go from HBM/L2 straight into
 * the MFMA operand registers and the VALU-bound inner loop only splits the activations. Needs Cin % 32 == 0 and
 * Cout % 128 == 0; always uses the split GEMM math. scale != NULL adds the fused LayerNorm2d statistics of
 * fl_conv2d_fwd_instnorm (same outputs, same workspace, act must be NONE). */
size_t fl_split_pack_weight_bytes(int N, int K);
int fl_split_pack_weight(const float* w, void* out, int N, int K, fl_stream_t stream);
int fl_conv2d_fwd_presplit(const float* x, const void* wsplit, const float* bias, float* y, int B, int H, int W, int Cin,
                           int Cout, int kh, int kw, int stride, int pad, int act, float act_param, const float* ln_w,
                           const float* ln_b, float eps, float* scale, float* shift, float* mean, float* rstd, void* ws,
                           size_t ws_bytes, fl_stream_t stream);
/* 3x3 / stride 1 / pad 1 and 4x4 / stride 2 / pad 1 forward conv in "f16x2" split math (csrc/conv_h2.hip; the
 * ResidualBlock and down-sampling convs of the frozen encoder, v2:159-178, v2:191-213). Every fp32 operand is scaled by a power of two and split by round-to-nearest into two fp16
 * pieces x*2^e = h + l (+ <= 2^-22 |x| 2^e); a product is 3 of the 4 piece products on v_mfma_f32_32x32x16_f16 (piece
 * products exact, fp32 accumulation): error vs fp64 0.5-1.7e-6 of the output's max for K = 1024..4608 (2-4x the native fp32
 * MFMA kernel's accumulation error, the class of the bf16x3 scheme) at half the matrix instructions of the bf16x3 scheme. fl_pack_weight_f16x2 scales (per tensor, amax -> [2^14,2^15)), splits
 * and lays out the packed weight [Cout][kh*kw*Cin] once (fl_pack_weight_f16x2_bytes(N,K) = 256 + 4*N*K bytes;
 * layout 1 for the 4x4 kernel, whose 16-wide k groups pair two taps x 8 channels; for the 3x3 kernel layout 2 when Cin % 64 == 0
 * (32-wide k groups pairing two taps x 16 channels for v_mfma_f32_16x16x32_f16: the shape that sustains the higher clock on
 * LDS-fed random operands, profiles/r03_mfma_shape_probe.txt) and layout 0 otherwise - fl_conv2d_fwd_f16x2 picks its kernel by
 * the same rule, fl_conv2d_f16x2_weight_layout(kh, Cin) returns it).
 * Activations are scaled by 2^x_exp, or - when x_amax != NULL - by the power of two that maps *x_amax (a device float
 * >= max|x|, kept by the kernel that produced x) into [2^14,2^15). |x| 2^e >= 65520 yields NaN outputs (loud).
 * Needs Cout % 128 == 0 and: 3x3 H == W in {8,16,32}, Cin % 32 == 0, B*H*W % 128 == 0 (256-pixel tiles, one 8-wave block per
 * CU; 128-pixel tiles with 4 waves when only that divides - FL_H2_BM=128 forces them: measured equal); 4x4 H == W in
 * {16,32,64}, Cin % 16 == 0, B*Ho*Wo % 256 == 0 (fl_conv2d_fwd_f16x2_applicable).
 * scale != NULL adds the fused LayerNorm2d statistics of fl_conv2d_fwd_instnorm (same outputs and workspace).
 * pre_scale / pre_shift ([B][Cin], both or neither): the conv consumes swish(x * pre_scale[b,c] + pre_shift[b,c]) - the
 * producer's LayerNorm2d-apply + Swish (v2:163-164, v2:193-194) fused into the halo staging, so that pass never
 * touches HBM; x_amax / x_exp then bound the transformed values. */
size_t fl_pack_weight_f16x2_bytes(int N, int K);
int fl_conv2d_f16x2_weight_layout(int kh, int Cin);
int fl_pack_weight_f16x2(const float* w, void* out, int N, int K, int layout, fl_stream_t stream);
int fl_conv2d_fwd_f16x2_applicable(const float* x, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
/* nn.Linear forward with a long reduction in f16x2 split math (the frozen encoder's heads, v2:210-221: K = 32768): y = x W^T +
 * bias with W packed by fl_pack_weight_f16x2 ([N][K], layout 0) and x scaled as for the convs (x_amax / x_exp). Split-K over
 * as many blocks as fill the chip (slabs in `ws`), then one reduce pass that also adds the bias. y is written as N / n_block
 * consecutive [M][n_block] matrices (n_block = N: plain row-major) - two concatenated heads come out as two tensors.
 * M % 256 == 0, N % 128 == 0, K % 32 == 0 (fl_linear_fwd_f16x2_applicable). */
int fl_linear_fwd_f16x2_applicable(int M, int N, int K);
size_t fl_linear_fwd_f16x2_workspace_bytes(int M, int N, int K);
int fl_linear_fwd_f16x2(const float* x, int lda, const void* wpack, const float* bias, float* y, int n_block, int M, int N, int K,
                        const float* x_amax, int x_exp, void* ws, size_t ws_bytes, fl_stream_t stream);
/* out[0] = max|x| over n floats: x_amax of fl_conv2d_fwd_f16x2 for inputs whose producer keeps no bound (the training paths'
 * activations and gradients, v2:1060-1134). */
int fl_absmax(const float* x, int64_t n, float* out, fl_stream_t stream);
int fl_conv2d_fwd_f16x2(const float* x, const void* wpack, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                        int kh, int kw, int stride, int pad, int act, float act_param, const float* ln_w, const float* ln_b,
                        float eps, float* scale, float* shift, float* mean, float* rstd, const float* x_amax, int x_exp,
                        const float* pre_scale, const float* pre_shift, void* ws, size_t ws_bytes, fl_stream_t stream);
/* dx = conv_transpose(dy, w): input gradient of the conv above (stride 1 any
 * kernel; stride 2 only for 4x4 p1).  dy:(B,Ho,Wo,Cout) -> dx:(B,H,W,Cin).
 * wp is the SAME packed weight as the forward. */
int fl_conv2d_dgrad(const float* dy, const float* wp, float* dx,
                    int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                    void* ws, size_t ws_bytes, fl_stream_t stream);
/* dwp[Cout][kh][kw][Cin] = sum_pixels dy (x) x  (packed layout, see fl_unpack_conv_weight) */
int fl_conv2d_wgrad(const float* dy, const float* x, float* dwp,
                    int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                    void* ws, size_t ws_bytes, fl_stream_t stream);
size_t fl_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
/* nn.ConvTranspose2d(k=4,s=2,p=1) forward (v2:256,262,268) as 4 sub-pixel 2x2
 * convolutions (no zero insertion).  x:(B,H,W,Cin) -> y:(B,2H,2W,Cout).
 * wtp = fl_pack_conv_weight of the (Cin,Cout,4,4) parameter = [Cin][4][4][Cout]. */
int fl_convT2d_fwd(const float* x, const float* wtp, const float* bias, float* y,
                   int B, int H, int W, int Cin, int Cout, int act, float act_param,
                   void* ws, size_t ws_bytes, fl_stream_t stream);

/* (O,I,kh,kw) -> [O][kh][kw][Ipad] (zero-filled pad channels) and back. */
int fl_pack_conv_weight(const float* w, float* wp, int O, int I, int kh, int kw, int Ipad, fl_stream_t stream);
int fl_unpack_conv_weight(const float* wp, float* w, int O, int I, int kh, int kw, int Ipad, fl_stream_t stream);


/* ------------------------------------------------------------------------ *
 * Normalisation family (NHWC, fp32).  `ws` >= fl_norm_workspace_bytes(B,HW,C).
 * The *_stats entry points run the streaming statistics pass and emit per-(image,
 * channel) scale/shift such that  norm(x) = x*scale + shift ; fl_affine_act then
 * applies  y = act(x*scale + shift) (+R)  in one pass.
 *   fl_instnorm_*  : LayerNorm2d, v2:144-156 (per-(n,c) stats over H*W, biased var)
 *   fl_groupnorm_* : nn.GroupNorm, v2:257,263,269,274
 *   fl_batchnorm_* : nn.BatchNorm2d, v2:991,995,999 (training stats + running update)
 * scale/shift are (B,C) for instance/group norm (per_image=1) and (C) for batch norm.
 * ------------------------------------------------------------------------ */
size_t fl_norm_workspace_bytes(int B, int HW, int C);
int fl_instnorm_stats(const float* x, const float* w, const float* b, float eps, float* scale, float* shift,
                      float* mean, float* rstd, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream);
int fl_groupnorm_stats(const float* x, const float* w, const float* b, float eps, int G, float* scale, float* shift,
                       float* mean, float* rstd, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream);
int fl_batchnorm_stats(const float* x, const float* w, const float* b, float eps, float momentum, float* running_mean,
                       float* running_var, int training, float* scale, float* shift, float* mean, float* rstd, int B,
                       int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream);
int fl_affine_act(const float* x, const float* scale, const float* shift, int per_image, const float* R, float* y, int B,
                  int HW, int C, int act, float act_param, fl_stream_t stream);
/* out[0] = mult * max|w| + max|b| (+ prev[0] when prev != NULL; w NULL = 1, b NULL = 0): a rigorous bound on the
 * magnitude of a LayerNorm2d / LayerNorm output (|x - mean| * rstd <= sqrt(n) = mult) after its affine map; Swish and
 * gates in (0,1) only shrink it, a residual adds its own bound (prev). Computed once per weight version and handed to
 * fl_conv2d_fwd_f16x2 as x_amax. */
int fl_affine_bound(const float* w, const float* b, int C, float mult, const float* prev, float* out, fl_stream_t stream);
/* backward of y = act(norm(x)); coef is scratch: 3*B*C floats (3*C for batch norm) */
int fl_instnorm_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                    const float* mean, const float* rstd, int act, float act_param, float* dx, float* dw, float* db,
                    float* coef, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream);
int fl_groupnorm_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                     const float* mean, const float* rstd, int G, int act, float act_param, float* dx, float* dw,
                     float* db, float* coef, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream);
int fl_batchnorm_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                     const float* mean, const float* rstd, int act, float act_param, float* dx, float* dw, float* db,
                     float* coef, int B, int HW, int C, void* ws, size_t ws_bytes, fl_stream_t stream);
/* Synchronised batch norm for the data-parallel VAE-GAN step (the reference's nn.BatchNorm2d layers of
 * Discriminator64, v2:591-603, under torch's SyncBatchNorm rule): the per-channel reductions are split where
 * ranks exchange them. sums = double[2*C]: {sum x, sum x^2} (dy NULL) or {sum g, sum g*x}, g = dy*act'(norm(x)).
 * The caller all-reduces `sums` (SUM) and finishes with the *_from_sums call, n_total = world * B * HW.
 * dx uses the global sums, dw/db this rank's local ones. coef is scratch: 8*C floats. */
int fl_batchnorm_sums(const float* x, const float* dy, const float* scale, const float* shift, int act,
                      float act_param, double* sums, int B, int HW, int C, void* ws, size_t ws_bytes,
                      fl_stream_t stream);
int fl_batchnorm_stats_from_sums(const double* sums, int64_t n_total, const float* w, const float* b, float eps,
                                 float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                 float* mean, float* rstd, int C, fl_stream_t stream);
int fl_batchnorm_bwd_from_sums(const float* dy, const float* x, const float* scale, const float* shift, const float* w,
                               const float* mean, const float* rstd, int act, float act_param,
                               const double* sums_global, const double* sums_local, int64_t n_total, float* dx,
                               float* dw, float* db, float* coef, int B, int HW, int C, fl_stream_t stream);
/* per-image channel sums: out1[b,c] = sum_pix a, out2[b,c] = sum_pix a*x, a = dy (or x if dy NULL) — v2:56,65 */
int fl_chan_sums(const float* x, const float* dy, float* out1, float* out2, int B, int HW, int C, void* ws,
                 size_t ws_bytes, fl_stream_t stream);
/* nn.LayerNorm over the last dim (v2:212,218,248,251,313,317,521,525,530):
 * y = act(mask * (xhat*w + b)) + R; mask (rows,D) optional = dropout scale (v2:522);
 * R optional residual (v2:548).  Backward: dx = LN'(dy) + R. */
int fl_layernorm_fwd(const float* x, const float* w, const float* b, const float* mask, const float* R, float* y,
                     float* mean, float* rstd, int rows, int D, float eps, int act, float act_param, fl_stream_t stream);
size_t fl_layernorm_bwd_workspace_bytes(int rows, int D);
int fl_layernorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* mask,
                     const float* mean, const float* rstd, const float* R, float* dx, float* dw, float* db, int rows, int D,
                     int act, float act_param, void* ws, size_t ws_bytes, fl_stream_t stream);

/* Fused multi-head self-attention over the HW tokens of a feature map (UNetAttentionBlock, v2:446-457): one launch for
 * softmax(scale * Q K^T) V of every (image, head), no (HW x HW) tensor in HBM. qkv (B,HW,3C): channel = which*C + head*d + i
 * (NHWC output of the 1x1 qkv conv); out (B,HW,C): channel = i*heads + head (the d-major merge of v2:456-457); P optional
 * (B,heads,HW,HW): the normalised weights, kept only when a backward will need them. Exact-fp32 MFMAs; the score tile is
 * built transposed so that the row softmax is in-lane arithmetic + wave shuffles and the accumulator tile feeds the PV MFMA as
 * it stands. Two kernels: ONE pass over the keys (online softmax, v_mfma_f32_16x16x4_f32, K / V^T staged in LDS per block, exp2
 * on the transcendental pipe) when P == NULL, d = C / heads is a multiple of 4 up to 64, HW % 64 == 0 and qkv is 16-byte
 * aligned; otherwise the two-pass kernel on v_mfma_f32_32x32x2_f32 (HW % 32 == 0, d <= 32). stats optional (B,heads,HW,2):
 * (row max of scale*s, sum of exp) - what fl_attention_bwd (d <= 32) recomputes the weights from. */
int fl_attention_fwd(const float* qkv, float* out, float* P, float* stats, int B, int HW, int C, int heads, float scale,
                     fl_stream_t stream);
/* stats optional (B,heads,HW,2): per query the max of scale*s and the sum of exp - all the backward needs to recompute P.
 * Backward of the same core (v2:446-457) in two launches, no (HW x HW) tensor in HBM, no atomics: a dQ kernel (one wave = 32
 * queries) and a dK / dV kernel (one wave = 32 keys), both recomputing the probabilities from Q, K and `stats`; merged / dmerged
 * in the forward's d-major channel order, dqkv in the qkv layout (every element written). Workspace: B*heads*HW floats.
 * With d % 4 == 0, HW % 64 == 0 and 16-byte aligned qkv / dqkv both run as LDS-staged v_mfma_f32_16x16x4_f32 kernels (K, V, Q,
 * dO and their transposes staged once per block; every accumulator tile is the next contraction's B operand as it stands);
 * otherwise (HW % 32 == 0) the 32x32x2 kernels with operands from global memory. d <= 32. */
size_t fl_attention_bwd_workspace_bytes(int B, int HW, int heads);
int fl_attention_bwd(const float* qkv, const float* merged, const float* dmerged, const float* stats, float* dqkv, int B, int HW,
                     int C, int heads, float scale, void* ws, size_t ws_bytes, fl_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Residual-block tail: CALayer (v2:53-67) x SpatialAttention (v2:69-81) x residual+Swish
 * (v2:176-177) fused into two streaming passes.  u = s[b,c]*(y*scale[b,c]+shift[b,c]).
 * ------------------------------------------------------------------------ */
int fl_res_tail_map(const float* y, const float* scale, const float* shift, const float* s, int s_per_image, float* map,
                    int B, int HW, int C, fl_stream_t stream);
/* With map AND att_out the spatial-attention gate sigmoid(conv7x7(map)) is computed by its own one-thread-per-pixel kernel into
 * att_out first and the channel pass only streams (the 98-tap stencil inside the streaming pass cost 59 of 135 us at 32x32x128);
 * without att_out the gate is computed inside the pass.
 * fl_res_tail_out_pre: as fl_res_tail_out, with the residual given before its LayerNorm2d-apply + Swish:
 * xres := swish(xres * xres_scale[b,c] + xres_shift[b,c]) (the block input is never materialised, engine.encoder_infer). */
int fl_res_tail_out_pre(const float* y, const float* scale, const float* shift, const float* s, int s_per_image, const float* map,
                        const float* saw, const float* xres, const float* xres_scale, const float* xres_shift, float* out,
                        float* att_out, int B, int H, int W, int C, int final_swish, fl_stream_t stream);
int fl_res_tail_out(const float* y, const float* scale, const float* shift, const float* s, int s_per_image,
                    const float* map,
                    const float* saw, const float* xres, float* out, float* att_out, int B, int H, int W, int C,
                    int final_swish, fl_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Elementwise / gather / diffusion / RNG
 * ------------------------------------------------------------------------ */
int fl_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int Cpad, fl_stream_t stream);
int fl_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, int Cpad, fl_stream_t stream);
int fl_act_fwd(const float* x, float* y, int64_t n, int act, float p, fl_stream_t stream);          /* Swish v2:48-50 */
int fl_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, float p, fl_stream_t stream);
int fl_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n, fl_stream_t stream);
/* dst[i][:] = a*src[i][:] + b*dst[i][:] for `entries` independent (src, dst, n) triples in as few launches as the
 * kernel-argument table allows (96 per launch): per-parameter gradients -> flat gradient buffer (v2:1127, v2:1262).
 * src/dst/n are HOST arrays. */
int fl_multi_axpby(const void* const* src, void* const* dst, const int64_t* n, int entries, float a, float b,
                   fl_stream_t stream);
/* out[i][c] = alpha[i] * sum_m x[i][m][c] for `entries` independent row-major [M][N] matrices in one launch (64 per
 * launch): the bias gradients of the denoiser's nn.Linear layers (v2:1262), collected at the end of the backward.
 * x/out/M/N/alpha are HOST arrays. */
int fl_multi_colsum(const void* const* x, void* const* out, const int* M, const int* N, const float* alpha, int entries,
                    fl_stream_t stream);
/* attention-dropout of nn.MultiheadAttention on a length-1 sequence (v2:514,550-551) */
int fl_scale_heads(const float* v, const float* head_mask, float* out, int B, int D, int heads, fl_stream_t stream);
int fl_add_rows(const float* x, const float* y, int y_rows, float* out, int B, int D, fl_stream_t stream);
/* freqs[n_ch/2] = exp(-i*ln(1e4)/(n_ch/2-1)), the constant table of v2:411-412 */
int fl_time_sinusoid(const int64_t* t, const float* freqs, float* out, int Bt, int n_ch, fl_stream_t stream); /* v2:410-414 */
int fl_embedding_fwd(const float* table, const int64_t* idx, float* out, int n, int D, fl_stream_t stream); /* v2:430 */
int fl_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, int n, int D, int rows, fl_stream_t stream);
int fl_q_sample(const float* x0, const float* eps, const int64_t* t, const float* alpha_bar, float* xt, int B, int D,
                fl_stream_t stream);                                                                  /* v2:574-578 */
int fl_p_sample_update(const float* xt, const float* eps_theta, const float* noise, const int32_t* t_ptr,
                       const float* alpha, const float* alpha_bar, const float* beta, float* out, int64_t n,
                       fl_stream_t stream);                                                           /* v2:584-592 */
int fl_step_counter_add(int32_t* t_ptr, int64_t* t64_ptr, int delta, fl_stream_t stream);
int fl_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, float* logvar_clamped,
                      int64_t n, fl_stream_t stream);                                                 /* v2:339-343,352 */
/* Philox4x32-10; mode 0 normal, 1 dropout scale {0,1/(1-p)}, 2 uniform. counter = offset + dev_offset[0] */
int fl_philox_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, const int64_t* dev_offset, int mode, float p,
                   fl_stream_t stream);
int fl_philox_randint(int64_t* out, int64_t n, uint64_t seed, uint64_t offset, const int64_t* dev_offset, int hi,
                      fl_stream_t stream);
int fl_advance_offset(int64_t* dev_offset, int64_t delta, fl_stream_t stream);
/* The tail of BOTH encoder heads in one launch (csrc/latent_prep.hip; v2:210-221: fc_mu / fc_logvar are Linear -> LayerNorm ->
 * Swish -> Linear; this is everything behind the first Linear): g = [2][B][K] pre-LayerNorm activations (head 0 = mu, 1 = logvar,
 * bias included), K in {256, 512}, L % 32 == 0 (fl_encoder_heads_applicable), exact-fp32 MFMA. Outputs mu / logvar [B][L] (either
 * may be NULL) and, when z != NULL, reparameterize (v2:339-343): z = mu + eps * exp(0.5 * clamp(logvar, -2, 10)) with eps the
 * N(0,1) numbers fl_philox_fill(mode 0) produces at counter (dev_offset ? *dev_offset : 0). With dev_offset the kernel also moves
 * the counter on by (B*L + 3) / 4 when its last block finishes: done_counter is a zero-initialised uint32 word it uses (and
 * re-zeroes) for that. Replaces 9 launches of the frozen-encoder step (2 x [LayerNorm, GEMM, split-K reduce], fill, bump, reparam). */
int fl_encoder_heads_applicable(int B, int K, int L);
int fl_encoder_heads_fwd(const float* g, const float* ln_w_mu, const float* ln_b_mu, const float* ln_w_logvar,
                         const float* ln_b_logvar, float ln_eps, const float* w_mu, const float* b_mu, const float* w_logvar,
                         const float* b_logvar, float* mu, float* logvar, float* z, uint64_t seed, int64_t* dev_offset,
                         void* done_counter, int B, int K, int L, fl_stream_t stream);
/* The draws of one latent-DDPM training step and q_sample in one launch (v2:1260-1261, v2:574-578): t[b] = the number
 * fl_philox_randint(hi = n_steps) gives at counter base + b, noise = fl_philox_fill(mode 0) at counter base + B (base = offset +
 * *dev_offset), x_t = sqrt(alpha_bar[t]) z + sqrt(1 - alpha_bar[t]) noise. masks != NULL adds a second, independent draw: n_masks
 * dropout scales {0, 1/(1-p)} = fl_philox_fill(mode 1) of the stream (mask_seed, mask_offset + *mask_dev_offset). No counter is
 * moved: the caller bumps them (B + (B*L + 3)/4 and (n_masks + 3)/4), e.g. through fl_program_run's post hook. L % 4 == 0. */
int fl_diffuse_prep(const float* z, const float* alpha_bar, int n_steps, uint64_t seed, uint64_t offset, const int64_t* dev_offset,
                    int64_t* t, float* noise, float* xt, int B, int L, float* masks, int64_t n_masks, uint64_t mask_seed,
                    uint64_t mask_offset, const int64_t* mask_dev_offset, float p, fl_stream_t stream);
size_t fl_colsum_workspace_bytes(int M, int N);
/* out = beta*out + alpha * column sums of x (M,N): bias gradients */
int fl_colsum(const float* x, int64_t ldx, float* out, int M, int N, float alpha, float beta, void* ws, size_t ws_bytes,
              fl_stream_t stream);
/* out = a + sigmoid(w[0])*b (learned residual gate v2:533,561; w stays on the device); a may be NULL */
int fl_gated_add(const float* a, const float* b, const float* w, float* out, int64_t n, fl_stream_t stream);
/* out[0] = sigmoid'(w[0]) * sum(x*y): gradient of the gate parameter */
int fl_gate_grad(const float* x, const float* y, const float* w, float* out, int64_t n, fl_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Losses + optimizer tail (device-resident scalars; no host syncs)
 * ------------------------------------------------------------------------ */
int fl_euclid_loss_fwd(const float* x, const float* y, float* norms, float* loss, int B, int64_t D, int reduction,
                       fl_stream_t stream);                                                           /* v2:293-302 */
int fl_euclid_loss_bwd(const float* x, const float* y, const float* norms, float gscale, const float* gdev, float* dx,
                       float* dy, int B, int64_t D, int reduction, fl_stream_t stream);
size_t fl_sqnorm_workspace_bytes(void);
int fl_sqnorm(const float* x, int64_t n, float* out, int accumulate, void* ws, size_t ws_bytes, fl_stream_t stream);
int fl_clip_coef(const float* sqnorm, float max_norm, float* coef, float* norm_out, fl_stream_t stream); /* v2:1128,1263 */
int fl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, const int32_t* step_dev,
                 float b1, float b2, float eps, float wd, const float* coef_dev, float gscale, int decoupled,
                 const float* betas_dev, fl_stream_t stream);                                /* v2:1026,1028,1247 */
/* betas_dev != NULL: (beta1, beta2) are read from device memory instead of b1, b2 - a scheduler that cycles beta1 every batch
 * (torch's OneCycleLR, v2:1031-1037) then also drives a step that was captured in a hipGraph */
int fl_kl_fwd(const float* mu, const float* logvar, float* kl_row, float* musq_row, float* loss, int B, int D,
              fl_stream_t stream);                                                                    /* v2:380-386 */
int fl_kl_bwd(const float* mu, const float* logvar, const float* kl_row, float gscale, const float* gdev, float* dmu,
              float* dlv, int B, int D, float reg_scale, fl_stream_t stream);
/* reg_scale multiplies the gradient of the batch-SUM term 1e-4*sum(mu^2) (v2:385): 1 on one GPU; the world size in a
 * data-parallel step, where shard gradients are averaged although that term adds up over the global batch */
int fl_bce_fwd(const float* p, float target, float* loss, int n, fl_stream_t stream);                /* v2:1029 */
int fl_bce_bwd(const float* p, float target, float gscale, const float* gdev, float* dp, int n, fl_stream_t stream);
int fl_cross_entropy_fwd(const float* logits, const int64_t* labels, float* row_loss, float* probs, float* loss, int B,
                         int C, fl_stream_t stream);                                                  /* v2:1093 */
int fl_cross_entropy_bwd(const float* probs, const int64_t* labels, float gscale, const float* gdev, float* dlogits,
                         int B, int C, fl_stream_t stream);
/* y = softmax(scale*x) per row: attention weights of UNetAttentionBlock (v2:452-454) */
int fl_softmax_rows(const float* x, float* y, int rows, int D, float scale, fl_stream_t stream);
/* dx = scale * y * (dy - sum_j dy_j y_j): backward of the row softmax above (y = its output) */
int fl_softmax_rows_bwd(const float* y, const float* dy, float* dx, int rows, int D, float scale, fl_stream_t stream);
int fl_gather_rows(const float* table, const int64_t* idx, float* out, int B, int D, fl_stream_t stream); /* v2:363 */
int fl_update_centers(float* centers, const float* z, const int64_t* labels, float momentum, int B, int D, int classes,
                      fl_stream_t stream);                                                            /* v2:369-378 */
int fl_balance_losses(const float* terms, const float* weights, float* coeff, float* total, fl_stream_t stream); /* v2:1107-1125 */

/* Training-image transform on the device. Replaces transform_train (v2:24-30) / transform_test
 * (v2:32-35) + DataLoader(..., num_workers=2) (v2:1303) for a pre-resized uint8 image cache resident in HBM:
 *   cache  (N,64,64,3) uint8 NHWC - Resize((64,64)) already applied when the cache was built
 *   labels (N) int64 or NULL; index (B) int64 rows of the cache for this batch, or NULL = rows 0..B-1
 *   params (B,16) int32 or NULL. NULL / word 14 != 0 = transform_test (ToTensor only). Otherwise per image:
 *     [0] horizontal flip; [1..6] Image.rotate's inverse affine map in Pillow's 16.16 fixed point (a0 a1 a2 a3 a4 a5,
 *     half-pixel offsets folded into a2/a5), nearest neighbour, zero fill; [7..10] ColorJitter op order
 *     (0 brightness, 1 contrast, 2 saturation, 3 none); [11..13] float32 bits of the three factors.
 *   out (B,3,64,64) float32 NCHW in [0,1] (ToTensor); labels_out (B) int64 or NULL.
 * Bit-exact with Pillow's integer / float32 arithmetic (oracle/augment_oracle.py). Out-of-range indices are clamped. */
int fl_augment_batch(const uint8_t* cache, const int64_t* labels, const int64_t* index, const int32_t* params, float* out,
                     int64_t* labels_out, int B, int N, int H, int W, fl_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Backward of the residual-block tail (v2:173-177) and VAE-GAN-only helpers.
 *   A: dw = dout*swish'(u*att + x) ; da = att(1-att) * sum_c dw*u
 *   conv: dmap = conv7x7^T(da), dsaw = sum da (x) map
 *   C: dn = s * (dw*att + dmap0/C + [argmax] dmap1)   (feeds fl_instnorm_bwd)
 * ------------------------------------------------------------------------ */
int fl_res_tail_bwd_a(const float* dout, const float* y, const float* scale, const float* shift, const float* s,
                      int s_per_image, const float* att, const float* xres, int final_swish, float* dw, float* da,
                      int B, int HW, int C, fl_stream_t stream);
int fl_res_tail_bwd_c(const float* dw, const float* y, const float* scale, const float* shift, const float* s,
                      int s_per_image, const float* att, const float* dmap, float* dn, int B, int HW, int C,
                      fl_stream_t stream);
int fl_sa_conv_bwd(const float* da, const float* map, const float* saw, float* dmap, float* dsaw, int B, int H, int W,
                   fl_stream_t stream);
/* ds[c] = (w_ln*dw_ln + b_ln*db_ln)/s : gradient of the batch-shared channel-attention gate (v2:65-67) */
int fl_ca_gate_grad(const float* lnw, const float* lnb, const float* dlnw, const float* dlnb, const float* s, float* ds,
                    int C, fl_stream_t stream);
int fl_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, fl_stream_t stream);    /* VGG16, v2:965 */
int fl_maxpool2_bwd(const float* dy, const float* x, float* dx, int B, int H, int W, int C, fl_stream_t stream);
int fl_mul(const float* a, const float* b, float* out, int64_t n, fl_stream_t stream);            /* nn.Dropout v2:315,319 */
int fl_clamp_bwd(const float* d, const float* x, float lo, float hi, float* out, int64_t n, fl_stream_t stream); /* v2:352 */
int fl_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, int64_t n, fl_stream_t stream); /* v2:339-343 */

/* ------------------------------------------------------------------------ *
 * Persistent phase-program executor ("persistent fused fwd"): the
 * ConditionalUNet forward (v2:534-561) is ~35 dependent M=B GEMMs + row
 * LayerNorms, each far too small to amortise a kernel launch (>= 4.5 us per
 * graph node on MI355X). A *program* is a list of phases; a phase is a list of
 * independent jobs; one persistent grid (one 256-thread block per CU) executes
 * the phases in order with a device-wide barrier in between.
 *
 * Job kinds
 *   FL_PJOB_GEMM     C[M,N] = epi( sum_s A_s[M,K_s] * W_s[N,K_s]^T )  (nn.Linear layout; up to 3 concatenated-K
 *                    segments, e.g. h + time_proj(tc) of v2:543-544 as ONE contraction); epilogue:
 *                    v = acc + sum_s bias_scale_s*bias_s (all * (1+sigmoid(*gate)) when gate != 0); v = act(v);
 *                    v *= colmask[row, col/mask_group]; v += R[row, col].   N % 32 == 0, K_s % 32 == 0; lda may be 0
 *                    (one shared row, the (1,) timestep of v2:583).
 *   FL_PJOB_ROWNORM  y1 = act(mask * LN(x; W[0], bias[0], eps)) + R + sigmoid(*gate) * aux[1];  C = y1;
 *                    if W[1]: aux[0] = LN(y1; W[1], bias[1], eps2).   N = row width, N % 256 == 0, N <= 1024.
 *   FL_PJOB_SINUSOID C[M,N] = [sin(t*f) | cos(t*f)], t = (int64*)A[0] with element stride lda[0], f = W[0] (N/2 freqs).
 *   FL_PJOB_GATHER   C[M,N] = table A[0] rows picked by (int64*)W[0]. K[0] > 0 = the table has K[0] rows: an index outside
 *                    [0, K[0]) is never dereferenced and voids the run (abort word = 2; Program.check raises).
 *   FL_PJOB_PSAMPLE  the DDPM reverse update (v2:585-592) on C = A[0] = x_t (in place allowed), A[1] = eps_theta:
 *                    x <- (x - (1-a_t)/sqrt(1-ab_t)*eps)/sqrt(a_t) + [t>0] sqrt(beta_t)*z, t = *(int64*)aux[2], tables
 *                    W[0..2] = alpha, alpha_bar, beta; z = A[2] if given, else Philox4x32-10 N(0,1) with seed
 *                    (K[0] | K[1]<<32) and counter *(int64*)aux[0] + i/4 (the stream fl_philox_fill would produce).
 *   Training kinds (the denoiser's backward, loss and optimizer: v2:1260-1264 as two persistent launches):
 *   FL_PJOB_GEMM flags (field `flags`): FL_PG_TA / FL_PG_TW = operand A / W is stored with the contraction index as
 *                    its memory ROW: A_s[k*lda + m], W_s[k*ldw + n]  (dgrad: dx = dy * W is {A = dy, W = weight, TW};
 *                    wgrad: dW = dy^T x is {A = dy, W = x, TA|TW, K = batch}); FL_PG_COLSUM (with TA): the tiles of
 *                    column 0 also emit aux[0][m] = alpha * (1 + sigmoid(*gate) if gate) * sum_k A[k][m] (the bias
 *                    gradient rides on the weight gradient); FL_PG_ACTBWD: v *= act'(aux[1][row*ld2 + col]) after the
 *                    bias (act names the activation whose derivative is taken; no forward activation is applied);
 *                    aux[3] != 0: the pre-activation value is stored there too (row stride ldc). FL_PG_T64 (with TA|TW,
 *                    one segment, M, N % 64 == 0, no epilogue other than FL_PG_COLSUM): 64x64 output tiles from one
 *                    LDS stage - the weight-gradient form.
 *   FL_PJOB_LNBWD    row LayerNorm backward: x = A[0], dy = A[1]; u = mask*(xhat*W[0]+bias[0]); gl = dy*act'(u)*mask;
 *                    C = rstd*(gl*w - mean(gl*w) - xhat*mean(gl*w*xhat)) + R; aux[0] = gl (optional), aux[1] = [M][2]
 *                    (mean, rstd) recomputed from x exactly as FL_PJOB_ROWNORM computes them. Two-stage form (A[2] != 0,
 *                    the backward of ROWNORM's second LayerNorm chained into its first): stage 1 = plain LN backward with
 *                    W[1], bias[1], eps2 on x = A[0], dy = A[1], + R -> aux[2] (stats -> aux[3]); stage 2 = the above on
 *                    x = A[2] with that result as dy (no R).
 *   FL_PJOB_LNCOL    C[n] = sum_m gl[m][n]*(x[m][n]-mean_m)*rstd_m, aux[0][n] = sum_m gl[m][n]; gl = A[0], x = A[1],
 *                    stats = aux[1]  (LayerNorm weight / bias gradients, 32 columns per tile).
 *   FL_PJOB_LOSS     rows of euclidean_distance_loss (v2:293-302) and its gradient: d = A[0]-A[1]; aux[0][m] =
 *                    sqrt(sum d^2 + 1e-8); C = alpha*d/norm; aux[1][m] = sum_n C[m][n]*A[2][m][n].
 *   FL_PJOB_FIN      one tile: C[0] = mean(A[0][0..M)); aux[0][0] = s(1-s)*sum(A[1][0..M)), s = sigmoid(*gate).
 *   FL_PJOB_EWADD    C = A[0] + A[1] (+ A[2]).
 *   FL_PJOB_EMBBWD   C[r][:] = sum over b with idx[b] == r of A[0][b][:]  (idx = (int64*)W[0], K[0] = table rows, M = batch).
 *   FL_PJOB_SQNORM   C[t] = sum of squares of tile t's slice of A[0][0..K[0])  (256 tiles).
 *   FL_PJOB_ADAM     clip_grad_norm_ + Adam/AdamW (torch semantics) on the flat buffers: every tile sums the 256 partials
 *                    W[0] -> norm = alpha*sqrt(.), coef = min(1, bias_scale[2]/(norm+1e-6)); g' = g*alpha*coef; step =
 *                    *(int*)W[2] + 1 (bump it with fl_program_post); p = C, g = A[0], m = aux[0], v = aux[1], lr = *W[1],
 *                    b1, b2 = bias_scale[0..1] (or, when bias[0] != 0, the two device floats it points to), eps, weight decay = eps2 (decoupled iff flags & FL_PG_DECOUPLED);
 *                    tile 0 stores coef -> aux[2], norm -> aux[3].
 * GEMM jobs may pick bias[0] from a table: when aux[2] != 0 the row *(int64*)aux[2] of stride bias_row_stride is
 * used (time-dependent terms of the sampler, precomputed for every timestep).
 * Pointer fields hold a device address, 0 (none), or a small integer 1..FL_PROG_MAX_IO naming slot (value-1) of the
 * `io` array handed to fl_program_run (inputs/outputs that change per call while the program stays resident).
 * ------------------------------------------------------------------------ */
#define FL_PROG_MAX_IO 16
#define FL_PROG_MAX_SEG 3
enum { FL_PJOB_GEMM = 0, FL_PJOB_ROWNORM = 1, FL_PJOB_SINUSOID = 2, FL_PJOB_GATHER = 3, FL_PJOB_PSAMPLE = 4,
       FL_PJOB_LNBWD = 5, FL_PJOB_LNCOL = 6, FL_PJOB_LOSS = 7, FL_PJOB_FIN = 8, FL_PJOB_EWADD = 9, FL_PJOB_EMBBWD = 10,
       FL_PJOB_SQNORM = 11, FL_PJOB_ADAM = 12 };
enum { FL_PG_TA = 1, FL_PG_TW = 2, FL_PG_COLSUM = 4, FL_PG_ACTBWD = 8, FL_PG_DECOUPLED = 16, FL_PG_T64 = 32,
       FL_PG_DMA = 64,     /* set by fl_program_job_layout: a GEMM job whose operands go global -> LDS by LDS-DMA (not TA, every
                              K_s % 128 == 0, operands 16-byte aligned): 4-stage ring of 128-wide k chunks */
       FL_PG_WCONST = 128, /* set by the builder: no job of the program writes this GEMM job's W operands (weights), so a block
                              may request their first chunks BEFORE the barrier that precedes the job's phase */
       FL_PG_WPACK = 256   /* (with FL_PG_DMA, not TW) every W_s points to fl_program_pack_weight's image of the [N][K_s] weight
                              (ldw_s unused): the 16 KB a (32-column tile, 128-k chunk) needs are one contiguous run in LDS image
                              order. Frozen weights only (inference programs): the row-major walk costs 1.5x the time per chunk
                              (tools/dma_rate_probe.hip, variants 4 / 7) */ };
typedef struct fl_pjob {
  uint64_t A[FL_PROG_MAX_SEG], W[FL_PROG_MAX_SEG], bias[FL_PROG_MAX_SEG];
  uint64_t R, C, mask, gate;
  uint64_t aux[4];
  int32_t kind, M, N, nseg;
  int32_t K[FL_PROG_MAX_SEG], lda[FL_PROG_MAX_SEG], ldw[FL_PROG_MAX_SEG];
  int32_t ldr, ldc, act, mask_group;
  int32_t tile_begin, ntiles; /* tiles of this job inside its phase: [tile_begin, tile_begin + ntiles) */
  float bias_scale[FL_PROG_MAX_SEG];
  float eps, eps2;
  int32_t bias_row_stride, flags;
  float alpha;
  int32_t ld2;
  int32_t nrt, per; /* filled by fl_program_job_layout: the job's tiles are nrt row tiles x per tiles each */
} fl_pjob; /* 256 bytes (one 32-bit word per lane of a wave), no implicit padding */
/* Counters bumped once per run, after the last phase, by the last block to leave (the device-resident DDPM step
 * counter in its int32 and int64 forms, the Philox offset): ptr == 0 entries are skipped. */
typedef struct fl_program_post {
  uint64_t ptr64[2];
  int64_t delta64[2];
  uint64_t ptr32;
  int32_t delta32, pad_;
} fl_program_post;
#define FL_PPHASE_ACQUIRE 1 /* flags: invalidate L1/L2 before this phase (it re-reads a buffer that an earlier phase of
                              the same run overwrote AFTER some block had read it; a run starts with clean caches) */
#define FL_PPHASE_LOCAL 2   /* flags: every job of the phase is row-local (row tile rt of each job only needs row tile rt of
                              what earlier phases produced): with FL_PROG_XCD_LOCAL the row tiles are dealt to the XCDs
                              (rt % 8, by HW_REG_XCC_ID), outputs are stored with plain stores (they stay in that XCD's L2)
                              and two consecutive local phases are separated by a barrier among that XCD's blocks only
                              (a flag line in L2); a local phase followed by a device-wide phase ends in an L2 write-back
                              + the device-wide barrier */
#define FL_PPHASE_RINGSAFE 4 /* flags: no job of the phase touches the GEMM tiles' LDS rings (only ROWNORM, SINUSOID, GATHER, PSAMPLE,
                              LNBWD, LOSS, EWADD jobs): a block may stream the weights of its first GEMM tile of the phase AFTER
                              this one into LDS while it runs this phase's tiles */
typedef struct fl_pphase {
  int32_t job_begin, job_end, ntiles, flags;
} fl_pphase;
/* sizeof(fl_pjob) / sizeof(fl_pphase) as compiled: the host-side builder asserts its packing against these */
int fl_program_job_bytes(void);
int fl_program_phase_bytes(void);
/* tiles a job occupies = ceil(M/32) row tiles x tiles per row tile (GEMM: N/32 (FL_PG_T64: M/64 x N/64 in all); ROWNORM: 32 (one row per tile, a quarter row per wave); SINUSOID: ceil(32*N/2/256);
 * GATHER, PSAMPLE, EWADD: ceil(32*N/4/256); LNBWD, LOSS: 8; LNCOL: N/32, EMBBWD: K[0], FIN: 1, SQNORM, ADAM: 256 in all);
 * negative fl_status if the job violates the shape rules above. `job` is a HOST pointer. */
int fl_program_job_tiles(const void* job);
/* The same check; also fills job->nrt / job->per and sets or clears FL_PG_DMA in job->flags (`job` is a writable HOST
 * pointer). Every job handed to fl_program_run must have been through this call. Returns the tile count. */
int fl_program_job_layout(void* job);
/* out[((ct * (K/128) + c) * 32 + r) * 128 + 4 * pb + e] = w[(32 ct + r) * ldw + 128 c + 4 * (pb ^ (r & 15)) + e]: the weight [N][K]
 * (row pitch ldw floats; N % 32 == 0, K % 128 == 0) as FL_PG_WPACK reads it; out = N * K floats. */
int fl_program_pack_weight(const float* w, float* out, int N, int K, int ldw, fl_stream_t stream);
/* Bytes of the barrier area `sync` (arrival counters spread over memory channels, departures, abort flag). */
int fl_program_sync_bytes(void);
#define FL_PROG_SYNC_ABORT_WORD (8 * 1024 + 16) /* uint32 index inside `sync`: != 0 after a run = the run is void (1: a barrier timed out, 2: a checked gather index was out of range, 3: the grid's blocks were not spread evenly over the XCDs (FL_PROG_XCD_LOCAL)) */
/* Run a device-resident program (n_phases*16 + n_jobs*256 bytes <= 29.5 KiB: it is copied to LDS). `io`: HOST array of
 * n_io device pointers (read at call time). `sync`: fl_program_sync_bytes() zeroed device bytes owned by the caller,
 * reused across runs (abort word != 0 afterwards: the run is void and the caller must zero `sync`). max_blocks: 0 =
 * one block per CU. Concurrent runs must not share `sync`; two programs running at once on one device may each wait
 * for CUs the other holds until the barrier timeout fires. `post`: HOST pointer to an fl_program_post, or NULL. */
#define FL_PROG_XCD_LOCAL 1 /* flags: honour FL_PPHASE_LOCAL (without it every phase is device-wide: write-through stores,
                              device-wide barriers, any block runs any tile) */
int fl_program_run(const void* phases, const void* jobs, int n_phases, int n_jobs, const void* const* io, int n_io,
                   void* sync, int max_blocks, int flags, const void* post, const void* guard_sync, fl_stream_t stream);
/* guard_sync: NULL, or the `sync` area of ANOTHER program whose outputs this one consumes (the optimizer program behind the
 * training program): while that area's abort word is set this launch does nothing either - a void gradient is never applied. */

#ifdef __cplusplus
}
#endif
#endif /* FLOWERS_HIP_H */
