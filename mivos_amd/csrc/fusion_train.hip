// Training step of the difference-aware FusionNet on gfx950 (reference: model/fusion_model.py:54-131 do_pass,
// model/losses.py:21-41 BootstrappedCE / :44-76 LossComputer, model/aggregate.py:39-53 aggregate_wbg_channel,
// train.py:27,96-124; torch.optim.Adam).  The forward pass and every data gradient (dgrad = a 3x3 convolution with the
// transposed, 180-degree-rotated weights) run on the library's convolution kernels; this file holds what those cannot do:
//
//   fusion_wgrad3x3_kernel    dW[n][tap][c] = sum_pixels g[p][n] * x[p + tap][c],  db[n] = sum_pixels g[p][n]
//                             a [32 x 9*32] x (pixels) GEMM with a reduction over ~10^6 pixels: exact fp32 MFMA
//                             (v_mfma_f32_32x32x2_f32: A = g (one pixel pair), B = the nine shifted x rows), nine 32x32
//                             accumulators per wave, waves walk image rows, fixed-order reduction over waves / workgroups
//                             (deterministic: the all-reduce across ranks is the only other place gradients are summed)
//   fusion_loss_kernel        sigmoid * selector -> aggregate_wbg_channel (logits, softmax) -> per-pixel cross-entropy
//   fusion_kth_loss_kernel    BootstrappedCE's top-p selection: exact k-th largest per-pixel loss of one sample by a 4-pass
//                             radix select on the float bits (+ count and sum of the losses above it)
//   fusion_loss_grad_kernel   d total_loss / d (FusionNet logits of the two objects)
//   mul_positive_kernel       g *= (y > 0): ReLU backward
//   adam_kernel               torch.optim.Adam's update on the flat parameter vector
#include <math.h>
#include <string.h>

#include "common.h"

namespace mivos {

// ---- weight / bias gradient of a 3x3, pad 1, stride 1 convolution ---------------------------------------------------
constexpr int WG_PART = 9 * 32 * 32 + 32;       // floats of one partial: [tap][n][c] + db[n]

template <int CX, int CG>   // channels of x (16 / 32), channels of g (32 / 1)
__global__ __launch_bounds__(256) void fusion_wgrad3x3_kernel(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ partial,
                                                              int N, int H, int W) {
  __shared__ float red[WG_PART];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, k = lane >> 5;       // A: channel n = col of g; B: channel c = col of x; k: pixel of the pair
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const int units = N * H, stride = gridDim.x * 4;
  for (int u = blockIdx.x * 4 + wave; u < units; u += stride) {
    const int img = u / H, y = u - img * H;
    const float *grow = g + ((long long)img * H + y) * W * CG;
    const float *xrow[3];
    bool rok[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      rok[dy] = (unsigned)yy < (unsigned)H;
      xrow[dy] = x + ((long long)img * H + (rok[dy] ? yy : y)) * W * CX;
    }
    for (int x0 = 0; x0 < W; x0 += 2) {
      const int xc = x0 + k;
      const float a = (xc < W && col < CG) ? grow[(long long)xc * CG + col] : 0.f;
      bsum += a;
      float b[9];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = xc + dx - 1;
          const bool ok = rok[dy] && (unsigned)xx < (unsigned)W && xc < W && col < CX;
          b[dy * 3 + dx] = ok ? xrow[dy][(long long)xx * CX + col] : 0.f;
        }
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
    }
  }
  // db: lanes (n, 0) and (n, 1) hold the two pixel parities
  bsum += __shfl_xor(bsum, 32);
  // reduction over the four waves in wave order (((wave 0 + wave 1) + wave 2) + wave 3) through one LDS buffer, then one partial
  // per workgroup
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(t * 32 + mfma32_row(r, lane)) * 32 + col] = acc[t][r];
      if (k == 0) red[9 * 32 * 32 + col] = bsum;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 32 + mfma32_row(r, lane)) * 32 + col];
      if (k == 0) bsum += red[9 * 32 * 32 + col];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float *out = partial + (long long)blockIdx.x * WG_PART;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(t * 32 + mfma32_row(r, lane)) * 32 + col] = acc[t][r];
    if (k == 0) out[9 * 32 * 32 + col] = bsum;
  }
}

// partial [n_part][WG_PART] -> dw OHWI [CG][9][CX], db [CG]; partials summed in ascending order
__global__ void fusion_wgrad_reduce_kernel(const float *__restrict__ partial, int n_part, float *__restrict__ dw, float *__restrict__ db, int CX, int CG) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= WG_PART) return;
  float s = 0.f;
  for (int p = 0; p < n_part; ++p) s += partial[(long long)p * WG_PART + e];
  if (e >= 9 * 32 * 32) {
    const int n = e - 9 * 32 * 32;
    if (n < CG) db[n] = s;
    return;
  }
  const int t = e / 1024, n = (e >> 5) & 31, c = e & 31;
  if (n < CG && c < CX) dw[((long long)n * 9 + t) * CX + c] = s;
}

// ---- loss -----------------------------------------------------------------------------------------------------------
// One thread per (sample, pixel).  z1 / z2: FusionNet logits of object 1 / 2 [B][P]; selector [B][2]; cls_gt [B][P] int32.
// prob = sigmoid(z) * selector (fusion_model.py:84-86); aggregate_wbg_channel (aggregate.py:39-53): raw = [prod(1 - prob), prob]
// clamped to [1e-7, 1 - 1e-7], logits = log(raw / (1 - raw)), mask = softmax over the 3 classes.  Per-pixel cross-entropy over
// all 3 classes when selector[b][1] > 0.5, over classes {0, 1} otherwise (losses.py:56-60).
struct LossPix {
  float p1, p2, raw[3], logit[3];
};
__device__ __forceinline__ LossPix loss_forward_pixel(float z1, float z2, float s1, float s2) {
  LossPix L;
  L.p1 = (1.f / (1.f + expf(-z1))) * s1;
  L.p2 = (1.f / (1.f + expf(-z2))) * s2;
  L.raw[0] = (1.f - L.p1) * (1.f - L.p2);
  L.raw[1] = L.p1;
  L.raw[2] = L.p2;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float q = fminf(fmaxf(L.raw[c], 1e-7f), 1.f - 1e-7f);
    L.logit[c] = logf(q / (1.f - q));
  }
  return L;
}

__global__ void fusion_loss_kernel(const float *__restrict__ z1, const float *__restrict__ z2, const float *__restrict__ selector,
                                   const int *__restrict__ cls_gt, float *__restrict__ logits, float *__restrict__ mask, float *__restrict__ loss,
                                   int B, long long P) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * P) return;
  const int b = (int)(e / P);
  const long long px = e - (long long)b * P;
  const float s1 = selector[2 * b], s2 = selector[2 * b + 1];
  const LossPix L = loss_forward_pixel(z1[e], z2[e], s1, s2);
  const float m = fmaxf(L.logit[0], fmaxf(L.logit[1], L.logit[2]));
  const float e0 = expf(L.logit[0] - m), e1 = expf(L.logit[1] - m), e2 = expf(L.logit[2] - m);
  const float sum3 = (e0 + e1) + e2;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    logits[((long long)b * 3 + c) * P + px] = L.logit[c];
    mask[((long long)b * 3 + c) * P + px] = (c == 0 ? e0 : (c == 1 ? e1 : e2)) / sum3;
  }
  const int gt = cls_gt[e];
  const bool three = s2 > 0.5f;
  // F.cross_entropy = -log_softmax(logits)[gt] over the classes in use
  const float m2 = three ? m : fmaxf(L.logit[0], L.logit[1]);
  const float lse = m2 + logf(three ? ((expf(L.logit[0] - m2) + expf(L.logit[1] - m2)) + expf(L.logit[2] - m2))
                                    : (expf(L.logit[0] - m2) + expf(L.logit[1] - m2)));
  loss[e] = lse - L.logit[gt < 3 ? gt : 0];
}

// Exact k-th largest of loss[b][0..P) (one workgroup of 1024 threads per sample): MSB-first radix select, 8 bits per pass, on the
// order-preserving integer image of the floats.  out[b] = {tau, #(loss > tau), sum(loss > tau), #(loss == tau)}.
__device__ __forceinline__ uint32_t f2ord_u(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(1024) void fusion_kth_loss_kernel(const float *__restrict__ loss, const int *__restrict__ kk, float *__restrict__ out, long long P) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need;
  __shared__ float s_sum[1024];
  __shared__ unsigned s_cnt[1024], s_eq[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *src = loss + (long long)b * P;
  if (tid == 0) { s_prefix = 0u; s_need = (unsigned)kk[b]; }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix, himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (long long i = tid; i < P; i += 1024) {
      const uint32_t o = f2ord_u(src[i]);
      if ((o & himask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, d = 255u;
      for (;; --d) {                                   // largest digit first
        if (hist[d] >= need) break;
        need -= hist[d];
        if (d == 0u) break;
      }
      s_prefix = prefix | (d << shift);
      s_need = need;                                   // rank of the k-th largest inside the chosen bucket
    }
    __syncthreads();
  }
  const uint32_t tau_o = s_prefix;
  float sum = 0.f;
  unsigned cnt = 0u, eq = 0u;
  for (long long i = tid; i < P; i += 1024) {
    const float v = src[i];
    const uint32_t o = f2ord_u(v);
    if (o > tau_o) { sum += v; ++cnt; }
    eq += o == tau_o ? 1u : 0u;
  }
  s_sum[tid] = sum; s_cnt[tid] = cnt; s_eq[tid] = eq;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {                  // fixed-shape tree: deterministic
    if (tid < s) { s_sum[tid] += s_sum[tid + s]; s_cnt[tid] += s_cnt[tid + s]; s_eq[tid] += s_eq[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t u = (tau_o & 0x80000000u) ? (tau_o & 0x7fffffffu) : ~tau_o;
    out[4 * b] = __uint_as_float(u);
    out[4 * b + 1] = (float)s_cnt[0];
    out[4 * b + 2] = s_sum[0];
    out[4 * b + 3] = (float)s_eq[0];
  }
}

// d total_loss / d z1, d z2.  wsel[b] = {tau, weight of a pixel with loss > tau, weight of a pixel with loss == tau}
// (plain mean: tau = -inf, weight 1 / (P * B)).  A pixel's loss gradient w.r.t. the logits in use is w * (softmax - onehot).
__global__ void fusion_loss_grad_kernel(const float *__restrict__ z1, const float *__restrict__ z2, const float *__restrict__ selector,
                                        const int *__restrict__ cls_gt, const float *__restrict__ loss, const float *__restrict__ wsel,
                                        float *__restrict__ dz1, float *__restrict__ dz2, int B, long long P) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * P) return;
  const int b = (int)(e / P);
  const float tau = wsel[3 * b], lv = loss[e];
  const float w = lv > tau ? wsel[3 * b + 1] : (lv == tau ? wsel[3 * b + 2] : 0.f);
  if (w == 0.f) { dz1[e] = 0.f; dz2[e] = 0.f; return; }
  const float s1 = selector[2 * b], s2 = selector[2 * b + 1];
  const LossPix L = loss_forward_pixel(z1[e], z2[e], s1, s2);
  const bool three = s2 > 0.5f;
  const int gt = cls_gt[e];
  const float m = three ? fmaxf(L.logit[0], fmaxf(L.logit[1], L.logit[2])) : fmaxf(L.logit[0], L.logit[1]);
  const float e0 = expf(L.logit[0] - m), e1 = expf(L.logit[1] - m), e2 = three ? expf(L.logit[2] - m) : 0.f;
  const float sum = (e0 + e1) + e2;
  float gl[3] = {w * (e0 / sum - (gt == 0 ? 1.f : 0.f)), w * (e1 / sum - (gt == 1 ? 1.f : 0.f)), three ? w * (e2 / sum - (gt == 2 ? 1.f : 0.f)) : 0.f};
  // logits = log(q / (1 - q)), q = clamp(raw): d logit / d raw = 1 / (q (1 - q)) inside the clamp range, 0 outside
  float G[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r = L.raw[c];
    const bool in = r >= 1e-7f && r <= 1.f - 1e-7f;
    const float q = fminf(fmaxf(r, 1e-7f), 1.f - 1e-7f);
    G[c] = in ? gl[c] / (q * (1.f - q)) : 0.f;
  }
  const float dp1 = G[1] - G[0] * (1.f - L.p2), dp2 = G[2] - G[0] * (1.f - L.p1);     // raw0 = (1 - p1)(1 - p2)
  const float sg1 = 1.f / (1.f + expf(-z1[e])), sg2 = 1.f / (1.f + expf(-z2[e]));
  dz1[e] = dp1 * s1 * sg1 * (1.f - sg1);
  dz2[e] = dp2 * s2 * sg2 * (1.f - sg2);
}

__global__ void mul_positive_kernel(float *__restrict__ g, const float *__restrict__ y, long long n4) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<f32x4 *>(g)[e];
    const f32x4 t = reinterpret_cast<const f32x4 *>(y)[e];
    v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
    reinterpret_cast<f32x4 *>(g)[e] = v;
  }
}

// torch.optim.Adam (no amsgrad): g += wd * p; m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, long long n, float step_size,
                            float b1, float b2, float omb1, float omb2, float eps, float wd, float bc2_sqrt) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float pe = p[e];
  const float g = grad[e] + wd * pe;
  const float me = b1 * m[e] + omb1 * g;            // omb = 1 - beta, rounded from double like torch's python scalars
  const float ve = b2 * v[e] + omb2 * g * g;
  m[e] = me; v[e] = ve;
  const float denom = sqrtf(ve) / bc2_sqrt + eps;
  p[e] = pe - step_size * (me / denom);
}

}  // namespace mivos

using namespace mivos;

extern "C" int64_t mivos_fusion_wgrad_scratch_floats(void) { return 256ll * WG_PART; }

extern "C" int mivos_fusion_wgrad3x3(const float *x, int cx, const float *g, int cg, float *dw_ohwi, float *db, float *scratch, int64_t scratch_floats,
                                     int N, int H, int W, void *stream) {
  if (!x || !g || !dw_ohwi || !db || !scratch || N < 1 || H < 1 || W < 1 || !((cx == 16 || cx == 32)) || !(cg == 32 || cg == 1) || (cx == 16 && cg != 32))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_wgrad3x3: null pointer or unsupported channel counts (x: 16 / 32, g: 32 / 1)");
  long long units = (long long)N * H;
  int grid = (int)((units + 3) / 4 < 256 ? (units + 3) / 4 : 256);
  if (scratch_floats < (int64_t)grid * WG_PART) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_wgrad3x3: scratch too small (mivos_fusion_wgrad_scratch_floats)");
  hipStream_t st = (hipStream_t)stream;
  if (cx == 16) hipLaunchKernelGGL((fusion_wgrad3x3_kernel<16, 32>), dim3(grid), dim3(256), 0, st, x, g, scratch, N, H, W);
  else if (cg == 32) hipLaunchKernelGGL((fusion_wgrad3x3_kernel<32, 32>), dim3(grid), dim3(256), 0, st, x, g, scratch, N, H, W);
  else hipLaunchKernelGGL((fusion_wgrad3x3_kernel<32, 1>), dim3(grid), dim3(256), 0, st, x, g, scratch, N, H, W);
  if (int rc = check_launch("fusion_wgrad3x3")) return rc;
  hipLaunchKernelGGL(fusion_wgrad_reduce_kernel, dim3(cdiv(WG_PART, 256)), dim3(256), 0, st, (const float *)scratch, grid, dw_ohwi, db, cx, cg);
  return check_launch("fusion_wgrad_reduce");
}

extern "C" int mivos_fusion_loss(const float *z1, const float *z2, const float *selector, const int32_t *cls_gt, float *logits, float *mask, float *loss,
                                 int B, int64_t P, void *stream) {
  if (!z1 || !z2 || !selector || !cls_gt || !logits || !mask || !loss || B < 1 || P < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_loss: bad arguments");
  hipLaunchKernelGGL(fusion_loss_kernel, dim3(cdiv((long long)B * P, 256)), dim3(256), 0, (hipStream_t)stream, z1, z2, selector, cls_gt, logits, mask, loss, B,
                     (long long)P);
  return check_launch("fusion_loss");
}

extern "C" int mivos_fusion_kth_loss(const float *loss, const int32_t *k, float *out4, int B, int64_t P, void *stream) {
  if (!loss || !k || !out4 || B < 1 || P < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_kth_loss: bad arguments");
  hipLaunchKernelGGL(fusion_kth_loss_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, loss, k, out4, (long long)P);
  return check_launch("fusion_kth_loss");
}

extern "C" int mivos_fusion_loss_grad(const float *z1, const float *z2, const float *selector, const int32_t *cls_gt, const float *loss, const float *wsel,
                                      float *dz1, float *dz2, int B, int64_t P, void *stream) {
  if (!z1 || !z2 || !selector || !cls_gt || !loss || !wsel || !dz1 || !dz2 || B < 1 || P < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_loss_grad: bad arguments");
  hipLaunchKernelGGL(fusion_loss_grad_kernel, dim3(cdiv((long long)B * P, 256)), dim3(256), 0, (hipStream_t)stream, z1, z2, selector, cls_gt, loss, wsel, dz1,
                     dz2, B, (long long)P);
  return check_launch("fusion_loss_grad");
}

extern "C" int mivos_mul_positive(float *g, const float *y, int64_t n, void *stream) {
  if (!g || !y || n < 4 || (n & 3) || ((uintptr_t)g & 15) || ((uintptr_t)y & 15)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "mul_positive: bad arguments");
  const long long n4 = n / 4;
  const int blocks = (int)(n4 / 256 + 1 < 4096 ? n4 / 256 + 1 : 4096);
  hipLaunchKernelGGL(mul_positive_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, y, n4);
  return check_launch("mul_positive");
}

extern "C" int mivos_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                               double weight_decay, int step, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "adam_step: bad arguments");
  // bias corrections in double on the host, like torch.optim.Adam's python scalars
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long long)n, (float)(lr / bc1),
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2));
  return check_launch("adam_step");
}
