// Losses (v2:293-302, v2:380-386, v2:362-367, BCE v2:1029, CE v2:1093) and the optimizer
// tail (global-norm clip v2:1128/1263 + Adam/AdamW v2:1026-1028/1247) on flat fp32 buffers.
// Everything that the reference turns into a host sync (.item(), clip_grad_norm_) stays on the
// device: norms, clip coefficient, step count and learning rate are read from device scalars.
#include "fl_common.h"

namespace {

#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

static int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// per-sample sqrt(sum (x-y)^2 + 1e-8): grid = (B), block = 256
__global__ void euclid_norm_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ norms,
                                   long long D) {
  __shared__ float red[8];
  const long long b = blockIdx.x;
  const float* xp = x + b * D;
  const float* yp = y + b * D;
  float a = 0.f;
  for (long long i = threadIdx.x; i < D; i += blockDim.x) {
    const float d = xp[i] - yp[i];
    a += d * d;
  }
  a = fl::block_sum(a, red);
  if (threadIdx.x == 0) norms[b] = sqrtf(a + 1e-8f);
}
// out[0] = mean or sum of v[0..n)   (single block; n is a batch size)
__global__ void reduce_vec_kernel(const float* __restrict__ v, float* __restrict__ out, int n, int mean) {
  __shared__ float red[8];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i];
  a = fl::block_sum(a, red);
  if (threadIdx.x == 0) out[0] = mean ? a / n : a;
}
// dx = g * (x-y)/norm_b ; g = gscale * (gdev ? gdev[0] : 1) * (mean ? 1/B : 1)
__global__ void euclid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                  const float* __restrict__ norms, float gscale, const float* __restrict__ gdev,
                                  float* __restrict__ dx, float* __restrict__ dyv, long long D, long long total) {
  const float g = gscale * (gdev ? gdev[0] : 1.f);
  GRID_STRIDE(i, total) {
    const long long b = i / D;
    const float v = g * (x[i] - y[i]) / norms[b];
    if (dx) dx[i] = v;
    if (dyv) dyv[i] = -v;
  }
}

// two-stage sum of squares
__global__ void sqnorm_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ partial) {
  __shared__ float red[8];
  float a = 0.f;
  GRID_STRIDE(i, n) a += x[i] * x[i];
  a = fl::block_sum(a, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
__global__ void sqnorm_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out, int accumulate) {
  __shared__ float red[8];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) a += partial[i];
  a = fl::block_sum(a, red);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + a : a;
}
// coef = min(1, max_norm / (sqrt(sq) + 1e-6))  — torch.nn.utils.clip_grad_norm_ semantics
__global__ void clip_coef_kernel(const float* sq, float max_norm, float* coef, float* norm_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float nrm = sqrtf(sq[0]);
    const float c = max_norm / (nrm + 1e-6f);
    coef[0] = c < 1.f ? c : 1.f;
    if (norm_out) norm_out[0] = nrm;
  }
}

// Adam / AdamW, torch single-tensor semantics. step (int32) and lr (float) live on the device.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, const float* __restrict__ lr_dev,
                            const int* __restrict__ step_dev, float b1, float b2, float eps, float wd,
                            const float* __restrict__ coef_dev, float gscale, int decoupled, const float* __restrict__ betas_dev) {
  if (betas_dev) {
    b1 = betas_dev[0];
    b2 = betas_dev[1];
  }
  const float lr = lr_dev[0];
  const float step = (float)step_dev[0];
  const float bc1 = 1.f - powf(b1, step);
  const float bc2s = sqrtf(1.f - powf(b2, step));
  const float cs = gscale * (coef_dev ? coef_dev[0] : 1.f);
  const float step_size = lr / bc1;
  GRID_STRIDE(i, n) {
    float gi = g[i] * cs;
    float pi = p[i];
    if (decoupled) pi *= (1.f - lr * wd);
    else gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// KL (v2:380-386): per-sample -0.5*sum(1 + lv - mu^2 - exp(lv)), clamp [0,100], batch mean, + 1e-4*sum(mu^2)
__global__ void kl_rows_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                               float* __restrict__ kl_row, float* __restrict__ musq_row, int D) {
  __shared__ float red[8];
  const long long b = blockIdx.x;
  float a = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float m = fminf(fmaxf(mu[b * D + i], -10.f), 10.f);
    const float lv = fminf(fmaxf(logvar[b * D + i], -2.f), 10.f);
    a += 1.f + lv - m * m - expf(lv);
    q += m * m;
  }
  a = fl::block_sum(a, red);
  q = fl::block_sum(q, red);
  if (threadIdx.x == 0) {
    kl_row[b] = -0.5f * a;
    musq_row[b] = q;
  }
}
__global__ void kl_final_kernel(const float* kl_row, const float* musq_row, float* out, int B) {
  __shared__ float red[8];
  float a = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    a += fminf(fmaxf(kl_row[i], 0.f), 100.f);
    q += musq_row[i];
  }
  a = fl::block_sum(a, red);
  q = fl::block_sum(q, red);
  if (threadIdx.x == 0) out[0] = a / B + 1e-4f * q;
}
__global__ void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                              const float* __restrict__ kl_row, float gscale, const float* __restrict__ gdev,
                              float* __restrict__ dmu, float* __restrict__ dlv, int B, int D, float reg_scale) {
  const float g = gscale * (gdev ? gdev[0] : 1.f);
  const long long n = (long long)B * D;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / D);
    const float mr = mu[i], lr = logvar[i];
    const bool m_in = mr >= -10.f && mr <= 10.f, l_in = lr >= -2.f && lr <= 10.f;
    const float m = fminf(fmaxf(mr, -10.f), 10.f), lv = fminf(fmaxf(lr, -2.f), 10.f);
    const float k = kl_row[b];
    const float gate = (k >= 0.f && k <= 100.f) ? 1.f : 0.f;  // clamp passes gradient inside [min,max]
    const float gm = (gate * (1.f / B) * m + reg_scale * 2e-4f * m) * (m_in ? 1.f : 0.f);
    const float gl = gate * (1.f / B) * (-0.5f) * (1.f - expf(lv)) * (l_in ? 1.f : 0.f);
    dmu[i] = g * gm;
    dlv[i] = g * gl;
  }
}

// BCE on probabilities (nn.BCELoss, mean): loss = -mean(t*log(p) + (1-t)*log(1-p)), log clamped at -100
__global__ void bce_kernel(const float* __restrict__ p, float target, float* __restrict__ out, int n) {
  __shared__ float red[8];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float lp = fmaxf(logf(p[i]), -100.f), l1p = fmaxf(logf(1.f - p[i]), -100.f);
    a += -(target * lp + (1.f - target) * l1p);
  }
  a = fl::block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a / n;
}
// dL/dp for the mean BCE, times g
__global__ void bce_bwd_kernel(const float* __restrict__ p, float target, float gscale, const float* __restrict__ gdev,
                               float* __restrict__ dp, int n) {
  const float g = gscale * (gdev ? gdev[0] : 1.f) / n;
  GRID_STRIDE(i, n) {
    const float pi = p[i];
    // d/dp [-(t log p + (1-t) log(1-p))] with torch's clamp of the logs at -100 (zero grad when clamped)
    const float a = logf(pi) > -100.f ? -target / pi : 0.f;
    const float b = logf(1.f - pi) > -100.f ? (1.f - target) / (1.f - pi) : 0.f;
    dp[i] = g * (a + b);
  }
}

// softmax cross-entropy (mean over batch): one block per row
__global__ void ce_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                               float* __restrict__ row_loss, float* __restrict__ probs, int C) {
  __shared__ float red[8];
  const long long b = blockIdx.x;
  const float* lr = logits + b * C;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < C; i += blockDim.x) mx = fmaxf(mx, lr[i]);
  mx = fl::block_max(mx, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += blockDim.x) s += expf(lr[i] - mx);
  s = fl::block_sum(s, red);
  const float lse = mx + logf(s);
  for (int i = threadIdx.x; i < C; i += blockDim.x) probs[b * C + i] = expf(lr[i] - lse);
  if (threadIdx.x == 0) row_loss[b] = lse - lr[labels[b]];
}
__global__ void ce_bwd_kernel(const float* __restrict__ probs, const long long* __restrict__ labels, float gscale,
                              const float* __restrict__ gdev, float* __restrict__ dlogits, int B, int C) {
  const float g = gscale * (gdev ? gdev[0] : 1.f) / B;
  const long long n = (long long)B * C;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    const int b = (int)(i / C);
    dlogits[i] = g * (probs[i] - (labels[b] == c ? 1.f : 0.f));
  }
}

// row softmax of scale*x (attention weights of UNetAttentionBlock, v2:453-454): one block per row
__global__ void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int D, float scale) {
  __shared__ float red[8];
  const long long r = blockIdx.x;
  const float* xr = x + r * D;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < D; i += blockDim.x) mx = fmaxf(mx, xr[i] * scale);
  mx = fl::block_max(mx, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) s += expf(xr[i] * scale - mx);
  s = fl::block_sum(s, red);
  const float inv = 1.f / s;
  for (int i = threadIdx.x; i < D; i += blockDim.x) y[r * D + i] = expf(xr[i] * scale - mx) * inv;
}

// backward of y = softmax(scale*x) per row: dx = scale * y * (dy - sum_j dy_j y_j)   (UNetAttentionBlock training)
__global__ void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                        int D, float scale) {
  __shared__ float red[8];
  const long long r = blockIdx.x;
  const float* yr = y + r * D;
  const float* gr = dy + r * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) s += gr[i] * yr[i];
  s = fl::block_sum(s, red);
  for (int i = threadIdx.x; i < D; i += blockDim.x) dx[r * D + i] = scale * yr[i] * (gr[i] - s);
}

// gather rows: out[b,:] = table[idx[b],:]
__global__ void gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx,
                                   float* __restrict__ out, int B, int D) {
  const long long n = (long long)B * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    const int b = (int)(i / D);
    out[i] = table[idx[b] * D + d];
  }
}
// class-centre EMA (v2:369-378): for each class present, centre = mom*centre + (1-mom)*mean(z of class)
__global__ void update_centers_kernel(float* __restrict__ centers, const float* __restrict__ z,
                                      const long long* __restrict__ labels, float momentum, int B, int D, int classes) {
  const long long n = (long long)classes * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    const int c = (int)(i / D);
    float a = 0.f;
    int cnt = 0;
    for (int b = 0; b < B; ++b)
      if (labels[b] == c) { a += z[(long long)b * D + d]; ++cnt; }
    if (cnt > 0) centers[i] = momentum * centers[i] + (1.f - momentum) * (a / cnt);
  }
}

// loss balancing (v2:1107-1125) on the device: scales[k] = min(1, recon/(term_k + 1e-8)) etc.
// in: terms = {recon, percep, kl, adv, cls, center}; weights = {l_recon, l_vgg, klw*klf, l_gan, l_cls*clsf, l_center*cf}
// out: coeff[6] multiplicative factor of each term in total_loss, total[0]
__global__ void balance_losses_kernel(const float* __restrict__ terms, const float* __restrict__ weights,
                                      float* __restrict__ coeff, float* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double recon = terms[0], percep = terms[1], kl = terms[2], adv = terms[3];
  double ps = 1.0, ks = 1.0, gs = 1.0;
  if (recon > 1e-8) {
    ps = fmin(1.0, recon / (percep + 1e-8));
    ks = kl > 0.0 ? fmin(1.0, recon / (kl + 1e-8)) : 1.0;
    gs = fmin(1.0, recon / (adv + 1e-8));
  }
  const double c[6] = {(double)weights[0], weights[1] * ps, weights[2] * ks, weights[3] * gs, (double)weights[4], (double)weights[5]};
  double t = 0.0;
  for (int i = 0; i < 6; ++i) {
    coeff[i] = (float)c[i];
    t += c[i] * terms[i];
  }
  total[0] = (float)t;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

// per-sample norms + reduced loss (reduction: 0 none, 1 mean, 2 sum). Replaces v2:293-302.
int fl_euclid_loss_fwd(const float* x, const float* y, float* norms, float* loss, int B, int64_t D, int reduction,
                       fl_stream_t stream) {
  FL_REQUIRE(x && y && norms, "fl_euclid_loss_fwd: null arg");
  if (B == 0) return FL_OK;
  hipLaunchKernelGGL(euclid_norm_kernel, dim3(B), dim3(256), 0, ST(stream), x, y, norms, (long long)D);
  if (reduction != 0 && loss)
    hipLaunchKernelGGL(reduce_vec_kernel, dim3(1), dim3(256), 0, ST(stream), norms, loss, B, reduction == 1 ? 1 : 0);
  return fl::check_launch("euclid_loss_fwd");
}
int fl_euclid_loss_bwd(const float* x, const float* y, const float* norms, float gscale, const float* gdev, float* dx,
                       float* dy, int B, int64_t D, int reduction, fl_stream_t stream) {
  const long long total = (long long)B * D;
  if (!total) return FL_OK;
  const float g = reduction == 1 ? gscale / B : gscale;
  hipLaunchKernelGGL(euclid_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), x, y, norms, g, gdev, dx, dy, (long long)D, total);
  return fl::check_launch("euclid_loss_bwd");
}

size_t fl_sqnorm_workspace_bytes(void) { return 1024 * sizeof(float); }
// out[0] (+)= sum x^2
int fl_sqnorm(const float* x, int64_t n, float* out, int accumulate, void* ws, size_t ws_bytes, fl_stream_t stream) {
  FL_REQUIRE(ws && ws_bytes >= fl_sqnorm_workspace_bytes(), "fl_sqnorm: workspace too small");
  long long nb = (n + 256 * 8 - 1) / (256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  float* partial = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((int)nb), dim3(256), 0, ST(stream), x, (long long)n, partial);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, ST(stream), partial, (int)nb, out, accumulate);
  return fl::check_launch("sqnorm");
}
int fl_clip_coef(const float* sqnorm, float max_norm, float* coef, float* norm_out, fl_stream_t stream) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, ST(stream), sqnorm, max_norm, coef, norm_out);
  return fl::check_launch("clip_coef");
}
// decoupled=1: AdamW (v2:1026, v2:1247); 0: Adam with L2 (v2:1028). step_dev holds the 1-based step count.
int fl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, const int32_t* step_dev,
                 float b1, float b2, float eps, float wd, const float* coef_dev, float gscale, int decoupled, const float* betas_dev,
                 fl_stream_t stream) {
  FL_REQUIRE(p && g && m && v && lr_dev && step_dev, "fl_adam_step: null arg");
  if (!n) return FL_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), p, g, m, v, (long long)n, lr_dev, step_dev, b1, b2, eps, wd, coef_dev, gscale, decoupled, betas_dev);
  return fl::check_launch("adam_step");
}

int fl_kl_fwd(const float* mu, const float* logvar, float* kl_row, float* musq_row, float* loss, int B, int D,
              fl_stream_t stream) {
  if (!B) return FL_OK;
  hipLaunchKernelGGL(kl_rows_kernel, dim3(B), dim3(256), 0, ST(stream), mu, logvar, kl_row, musq_row, D);
  hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(256), 0, ST(stream), kl_row, musq_row, loss, B);
  return fl::check_launch("kl_fwd");
}
int fl_kl_bwd(const float* mu, const float* logvar, const float* kl_row, float gscale, const float* gdev, float* dmu,
              float* dlv, int B, int D, float reg_scale, fl_stream_t stream) {
  const long long n = (long long)B * D;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), mu, logvar, kl_row, gscale, gdev, dmu, dlv, B, D, reg_scale);
  return fl::check_launch("kl_bwd");
}
int fl_bce_fwd(const float* p, float target, float* loss, int n, fl_stream_t stream) {
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, ST(stream), p, target, loss, n);
  return fl::check_launch("bce_fwd");
}
int fl_bce_bwd(const float* p, float target, float gscale, const float* gdev, float* dp, int n, fl_stream_t stream) {
  if (!n) return FL_OK;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), p, target, gscale, gdev, dp, n);
  return fl::check_launch("bce_bwd");
}
int fl_cross_entropy_fwd(const float* logits, const int64_t* labels, float* row_loss, float* probs, float* loss, int B,
                         int C, fl_stream_t stream) {
  if (!B) return FL_OK;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(B), dim3(128), 0, ST(stream), logits, (const long long*)labels, row_loss, probs, C);
  hipLaunchKernelGGL(reduce_vec_kernel, dim3(1), dim3(256), 0, ST(stream), row_loss, loss, B, 1);
  return fl::check_launch("cross_entropy_fwd");
}
int fl_cross_entropy_bwd(const float* probs, const int64_t* labels, float gscale, const float* gdev, float* dlogits,
                         int B, int C, fl_stream_t stream) {
  const long long n = (long long)B * C;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), probs, (const long long*)labels, gscale, gdev, dlogits, B, C);
  return fl::check_launch("cross_entropy_bwd");
}
int fl_softmax_rows_bwd(const float* y, const float* dy, float* dx, int rows, int D, float scale, fl_stream_t stream) {
  FL_REQUIRE(y && dy && dx && D > 0, "fl_softmax_rows_bwd: bad args");
  if (rows == 0) return FL_OK;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(rows), dim3(D >= 256 ? 256 : 64), 0, ST(stream), y, dy, dx, D, scale);
  return fl::check_launch("softmax_rows_bwd");
}
int fl_softmax_rows(const float* x, float* y, int rows, int D, float scale, fl_stream_t stream) {
  if (!rows) return FL_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(D >= 256 ? 256 : 64), 0, ST(stream), x, y, D, scale);
  return fl::check_launch("softmax_rows");
}
int fl_gather_rows(const float* table, const int64_t* idx, float* out, int B, int D, fl_stream_t stream) {
  const long long n = (long long)B * D;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), table, (const long long*)idx, out, B, D);
  return fl::check_launch("gather_rows");
}
int fl_update_centers(float* centers, const float* z, const int64_t* labels, float momentum, int B, int D, int classes,
                      fl_stream_t stream) {
  const long long n = (long long)classes * D;
  if (!n) return FL_OK;
  hipLaunchKernelGGL(update_centers_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), centers, z, (const long long*)labels, momentum, B, D, classes);
  return fl::check_launch("update_centers");
}
int fl_balance_losses(const float* terms, const float* weights, float* coeff, float* total, fl_stream_t stream) {
  hipLaunchKernelGGL(balance_losses_kernel, dim3(1), dim3(64), 0, ST(stream), terms, weights, coeff, total);
  return fl::check_launch("balance_losses");
}

}  // extern "C"
