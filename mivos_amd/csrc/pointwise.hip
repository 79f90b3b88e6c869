// HBM-bound element-wise / small-stencil kernels of the propagation path (gfx950).
// All are memory-bound: 16-byte vector accesses where the layout allows, grid-stride loops capped
// at ~8 workgroups per CU (cdna guide, Guideline 11 / Appendix B "Element-wise").
#include <stdarg.h>
#include <string.h>

#include "conv_common.h"

namespace mivos {

char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

static inline int grid_for(int64_t work_items, int block = 256) {
  int64_t g = (work_items + block - 1) / block;
  const int64_t cap = 256 * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------- maxpool 3x3 / 2 / pad 1, NHWC
__global__ void maxpool3x3s2_kernel(const float *__restrict__ x, float *__restrict__ y, int N, int H, int W, int C4,
                                    int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int ih = oh * 2 - 1 + dy;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int iw = ow * 2 - 1 + dx;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[(((int64_t)n * H + ih) * W + iw) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    reinterpret_cast<f32x4 *>(y)[i] = m;
  }
}

// ---------------------------------------------------------------- bilinear helpers (align_corners=False)
// PyTorch area_pixel_compute_source_index: src = (dst + 0.5) * scale - 0.5, clamped at 0.
__device__ __forceinline__ void bilin_coord(int dst, float scale, int in_size, int &i0, int &i1, float &l1) {
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ void upsample2x_add_kernel(const float *__restrict__ skip, int64_t skip_ns, const float *__restrict__ up,
                                      float *__restrict__ out, int N, int h, int w, int C4) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(y, 0.5f, h, y0, y1, ly);
    bilin_coord(x, 0.5f, w, x0, x1, lx);
    const f32x4 *u = reinterpret_cast<const f32x4 *>(up) + (int64_t)n * h * w * C4 + c;
    const f32x4 v00 = u[((int64_t)y0 * w + x0) * C4], v01 = u[((int64_t)y0 * w + x1) * C4];
    const f32x4 v10 = u[((int64_t)y1 * w + x0) * C4], v11 = u[((int64_t)y1 * w + x1) * C4];
    const float hy = 1.f - ly, hx = 1.f - lx;
    const f32x4 s = reinterpret_cast<const f32x4 *>(skip + (int64_t)n * skip_ns)[((int64_t)y * W + x) * C4 + c];
    f32x4 o;
    o.x = s.x + (hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x));
    o.y = s.y + (hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y));
    o.z = s.z + (hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z));
    o.w = s.w + (hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w));
    reinterpret_cast<f32x4 *>(out)[i] = o;
  }
}

// The same with up to three outputs that feed the next layers without a conversion pass: dense fp32, SH32 (the pre-split
// activation format of the LDS-DMA convolutions, written into the interior of a zero-bordered buffer) and SH32 of relu(x)
// (pre-activation ResBlocks: the DMA-staged operand cannot be modified on load, so its producer applies the ReLU).
__global__ void upsample2x_add_multi_kernel(const float *__restrict__ skip, int64_t skip_ns, const float *__restrict__ up,
                                            float *__restrict__ out, float *__restrict__ raw, float *__restrict__ rel, int64_t a_ns,
                                            int64_t a_rs, int64_t a_ps, int N, int h, int w, int C4, int contig) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)N * H * W * C4;
  // Work assignment: every workgroup takes ONE contiguous run of output elements, and the runs of the workgroups that share an XCD
  // (hardware block b runs on XCD b % 8) are contiguous too, so each XCD owns one band of image rows.  An input pixel feeds the 2 x 2 ... 3 x 3
  // output pixels around it; with a grid-stride loop those land on all eight XCDs, whose L2s are not shared, and every XCD pulls its own
  // copy of every input line through the fabric: rocprofv3 FETCH_SIZE 304 MB per launch against 45 MB of input (6.7 x), at which point the
  // kernel runs at the fabric's rate, not at the rate of its 199 - 265 MB of writes (profiles/r04d_config3_pmc_traffic.json).
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = contig ? (int64_t)xcd_remap((int)blockIdx.x, (int)gridDim.x) * per : (int64_t)blockIdx.x * blockDim.x;
  const int64_t i1 = !contig ? total : (i0 + per < total ? i0 + per : total);
  const int64_t istep = contig ? (int64_t)blockDim.x : (int64_t)gridDim.x * blockDim.x;      // (contig = 0: grid-stride loop of rounds 1-3, A/B only)
  for (int64_t i = i0 + threadIdx.x; i < i1; i += istep) {
    // (element order (n, y, x, c).  Round 4 also measured (y, x, n, c) - the objects of a pixel adjacent, so that the decoder's broadcast skip
    // tensor is read once per pixel: fabric reads 182 -> 116 MB per launch, kernel time 77.8 -> 82.9 us: the kernel is bound by its 200 - 265 MB
    // of writes, which that order scatters over the N images; not kept.)
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(y, 0.5f, h, y0, y1, ly);
    bilin_coord(x, 0.5f, w, x0, x1, lx);
    const f32x4 *u = reinterpret_cast<const f32x4 *>(up) + (int64_t)n * h * w * C4 + c;
    const f32x4 v00 = u[((int64_t)y0 * w + x0) * C4], v01 = u[((int64_t)y0 * w + x1) * C4];
    const f32x4 v10 = u[((int64_t)y1 * w + x0) * C4], v11 = u[((int64_t)y1 * w + x1) * C4];
    const float hy = 1.f - ly, hx = 1.f - lx;
    const f32x4 s = reinterpret_cast<const f32x4 *>(skip + (int64_t)n * skip_ns)[((int64_t)y * W + x) * C4 + c];
    f32x4 o;
    o.x = s.x + (hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x));
    o.y = s.y + (hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y));
    o.z = s.z + (hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z));
    o.w = s.w + (hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w));
    if (out) reinterpret_cast<f32x4 *>(out)[i] = o;
    const long long pix = (long long)n * a_ns + (long long)y * a_rs + (long long)x * a_ps;
    if (raw) store_sh32x4(raw, pix, 4 * c, o);
    if (rel) store_sh32x4(rel, pix, 4 * c, f32x4{fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)});
  }
}

// MaxPool 3x3 / 2 / pad 1 writing SH32 into the interior of a zero-bordered buffer (the ResNet stem's output feeds stage 1)
__global__ void maxpool3x3s2_sh32_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t a_ns, int64_t a_rs, int64_t a_ps,
                                         int N, int H, int W, int C4, int Ho, int Wo, int contig) {
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  // contiguous run per workgroup, XCD-contiguous (see upsample2x_add_multi_kernel): the 3 x 3 / 2 windows of neighbouring outputs overlap
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = contig ? (int64_t)xcd_remap((int)blockIdx.x, (int)gridDim.x) * per : (int64_t)blockIdx.x * blockDim.x;
  const int64_t i1 = !contig ? total : (i0 + per < total ? i0 + per : total);
  const int64_t istep = contig ? (int64_t)blockDim.x : (int64_t)gridDim.x * blockDim.x;      // (contig = 0: grid-stride loop of rounds 1-3, A/B only)
  for (int64_t i = i0 + threadIdx.x; i < i1; i += istep) {
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int ih = oh * 2 - 1 + dy;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int iw = ow * 2 - 1 + dx;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[(((int64_t)n * H + ih) * W + iw) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    store_sh32x4(y, (long long)n * a_ns + (long long)oh * a_rs + (long long)ow * a_ps, 4 * c, m);
  }
}

// 3x3 / pad 1 convolution with ONE output channel, second half: t holds, per input pixel, the nine tap products
// t[p][k] = w[k] . x[p] (a 1x1 projection to 16 channels computed by the GEMM kernels, which read x exactly once);
// out[y][x] = bias + sum_k t[y + k/3 - 1][x + k%3 - 1][k], zero outside the image.
__global__ void tap_sum9_kernel(const float *__restrict__ t, const float *__restrict__ bias, float *__restrict__ out, int N, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  const float b = bias ? bias[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    int64_t r = i / W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const float *tn = t + (int64_t)n * H * W * 16;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += tn[((int64_t)yy * W + xx) * 16 + k];
    }
    out[i] = acc + b;
  }
}

// F.interpolate(mode='bilinear', align_corners=False) on an NHWC map (DeepLabV3+ head: ASPP output x4, _deeplab.py:48),
// written into a channel slice of a wider buffer (the concatenation with the low-level features is a stride, not a copy)
__global__ void resize_bilinear_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t y_nstride, int64_t y_pstride, int N,
                                            int h, int w, int H, int W, int C4) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ox = (int)(t % W); t /= W;
    const int oy = (int)(t % H);
    const int n = (int)(t / H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(oy, sy, h, y0, y1, ly);
    bilin_coord(ox, sx, w, x0, x1, lx);
    const f32x4 *u = reinterpret_cast<const f32x4 *>(x) + (int64_t)n * h * w * C4 + c;
    const f32x4 v00 = u[((int64_t)y0 * w + x0) * C4], v01 = u[((int64_t)y0 * w + x1) * C4];
    const f32x4 v10 = u[((int64_t)y1 * w + x0) * C4], v11 = u[((int64_t)y1 * w + x1) * C4];
    const float hy = 1.f - ly, hx = 1.f - lx;
    f32x4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *reinterpret_cast<f32x4 *>(y + (int64_t)n * y_nstride + ((int64_t)oy * W + ox) * y_pstride + 4 * c) = o;
  }
}

// nn.AdaptiveAvgPool2d(1) on NHWC (ASPPPooling, _deeplab.py:120-131): y[n][c] = mean over the P pixels; one workgroup per
// (image, group of 64 channels), pixels strided over the 4 waves, fixed summation order
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t P, int C) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, n = blockIdx.y;
  float s = 0.f;
  if (c < C)
    for (int64_t p = wave; p < P; p += 4) s += x[((int64_t)n * P + p) * C + c];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C) y[(int64_t)n * C + c] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)P;
}

// cv2.dilate(mask, 3x3 ones) on a binary plane (davis_processor.py:55-60): out = max over the 3x3 neighbourhood
__global__ void dilate3x3_kernel(const float *__restrict__ x, float *__restrict__ y, int planes, int H, int W) {
  const int64_t total = (int64_t)planes * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    int64_t r = i / W;
    const int yy = (int)(r % H);
    const float *pl = x + (r / H) * (int64_t)H * W;
    float m = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int y2 = yy + dy, x2 = xx + dx;
        if ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) m = fmaxf(m, pl[(int64_t)y2 * W + x2]);
      }
    y[i] = m;
  }
}

__global__ void resize_bilinear_kernel(const float *__restrict__ x, float *__restrict__ y, int planes, int h, int w,
                                       int H, int W, int act) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int64_t total = (int64_t)planes * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % W);
    int64_t t = i / W;
    const int oy = (int)(t % H);
    const int pl = (int)(t / H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(oy, sy, h, y0, y1, ly);
    bilin_coord(ox, sx, w, x0, x1, lx);
    const float *s = x + (int64_t)pl * h * w;
    const float hy = 1.f - ly, hx = 1.f - lx;
    float v = hy * (hx * s[y0 * w + x0] + lx * s[y0 * w + x1]) + ly * (hx * s[y1 * w + x0] + lx * s[y1 * w + x1]);
    if (act == 1) v = 1.f / (1.f + expf(-v));
    y[i] = v;
  }
}

// ---------------------------------------------------------------- area pool 16x16 (one wave per output cell row-chunk)
__global__ void area_pool16_kernel(const float *__restrict__ x, float *__restrict__ y, int planes, int H, int W) {
  const int h = H / 16, w = W / 16;
  const int64_t cells = (int64_t)planes * h * w;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t cell = wave; cell < cells; cell += nwaves) {
    const int cx = (int)(cell % w);
    int64_t t = cell / w;
    const int cy = (int)(t % h);
    const int pl = (int)(t / h);
    // lane -> (row = lane/4, 4 consecutive pixels): one float4 per lane covers the 16x16 block
    const int r = lane >> 2, q = lane & 3;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((int64_t)pl * H + cy * 16 + r) * W + cx * 16 + q * 4);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) y[cell] = s * (1.f / 256.f);
  }
}

// ---------------------------------------------------------------- aggregate (soft background / fixed background)
// prob [B][K][P] -> softmax over the K+1 logits log(p/(1-p)) (background first) -> out [B][K(+1)][P]; optionally the
// logits themselves [B][K+1][P] (aggregate_wbg_channel).  CACHE: logits of up to 32 objects stay in registers; the
// general variant recomputes them from prob in each of its three sweeps (same arithmetic, any K).
__device__ __forceinline__ float clamped_logit(float p, int hard) {
  const float c = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
  const float lg = logf(c / (1.f - c));
  return hard ? lg * 1000.f : lg;
}

template <bool WBG, bool CACHE>
__global__ void aggregate_kernel(const float *__restrict__ prob, float *__restrict__ out, float *__restrict__ logits, int B,
                                 int K, int64_t P, int keep_bg, int hard) {
  constexpr int KMAX = 32;
  const int64_t total = (int64_t)B * P;
  const int kout = keep_bg ? K + 1 : K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / P, px = i - b * P;
    const float *pb = prob + b * K * P + px;
    float *ob = out ? out + b * kout * P + px : nullptr;
    float *lb = logits ? logits + b * (K + 1) * P + px : nullptr;
    float l[CACHE ? KMAX + 1 : 1];
    float bg = WBG ? 1.f : 0.5f;
    float mx = -INFINITY;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const float pk = pb[(int64_t)k * P];
      if (WBG) bg *= (1.f - pk);
      const float lg = clamped_logit(pk, hard);
      if (CACHE) l[k + 1] = lg;
      if (lb) lb[(int64_t)(k + 1) * P] = lg;
      mx = fmaxf(mx, lg);
    }
    const float lbg = clamped_logit(bg, hard);
    if (lb) lb[0] = lbg;
    mx = fmaxf(mx, lbg);
    if (!ob) continue;
    float sum = 0.f;
    const float ebg = expf(lbg - mx);
    sum += ebg;
#pragma unroll 1
    for (int k = 1; k <= K; ++k) {
      const float e = expf((CACHE ? l[k] : clamped_logit(pb[(int64_t)(k - 1) * P], hard)) - mx);
      if (CACHE) l[k] = e;
      sum += e;
    }
    if (keep_bg) ob[0] = ebg / sum;
    float *o1 = keep_bg ? ob + P : ob;
#pragma unroll 1
    for (int k = 1; k <= K; ++k) {
      const float e = CACHE ? l[k] : expf(clamped_logit(pb[(int64_t)(k - 1) * P], hard) - mx);
      o1[(int64_t)(k - 1) * P] = e / sum;
    }
  }
}

template <bool WBG>
static int launch_aggregate(const float *prob, float *out, float *logits, int B, int K, int64_t P, int keep_bg, int hard,
                            hipStream_t st, const char *what) {
  if (!prob || (!out && !logits) || K < 1 || B < 1 || P < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "%s: bad arguments", what);
  if (K <= 32)
    hipLaunchKernelGGL((aggregate_kernel<WBG, true>), dim3(grid_for((int64_t)B * P)), dim3(256), 0, st, prob, out, logits, B, K, P, keep_bg, hard);
  else
    hipLaunchKernelGGL((aggregate_kernel<WBG, false>), dim3(grid_for((int64_t)B * P)), dim3(256), 0, st, prob, out, logits, B, K, P, keep_bg, hard);
  return check_launch(what);
}

__global__ void argmax_u8_kernel(const float *__restrict__ prob, int64_t plane_stride, uint8_t *__restrict__ out,
                                 int planes, int64_t P) {
  // 4 pixels per thread: float4 loads, one packed 32-bit store
  const int64_t P4 = P >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 best = reinterpret_cast<const f32x4 *>(prob)[i];
    uint32_t idx = 0;  // 4 x u8 lanes
    for (int c = 1; c < planes; ++c) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(prob + (int64_t)c * plane_stride)[i];
      if (v.x > best.x) { best.x = v.x; idx = (idx & 0xffffff00u) | (uint32_t)c; }
      if (v.y > best.y) { best.y = v.y; idx = (idx & 0xffff00ffu) | ((uint32_t)c << 8); }
      if (v.z > best.z) { best.z = v.z; idx = (idx & 0xff00ffffu) | ((uint32_t)c << 16); }
      if (v.w > best.w) { best.w = v.w; idx = (idx & 0x00ffffffu) | ((uint32_t)c << 24); }
    }
    reinterpret_cast<uint32_t *>(out)[i] = idx;
  }
}

__global__ void mask_diff_kernel(const float *__restrict__ mask, const float *__restrict__ prob, float *__restrict__ pos,
                                 float *__restrict__ neg, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = mask[i] - prob[i];
    pos[i] = fminf(fmaxf(d, 0.f), 1.f);
    neg[i] = fminf(fmaxf(-d, 0.f), 1.f);
  }
}

__global__ void sigmoid_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = 1.f / (1.f + expf(-x[i]));
}

__global__ void mask_others_kernel(const float *__restrict__ masks, float *__restrict__ others, int K, int64_t P) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    for (int a = 0; a < K; ++a) {
      float s = 0.f;  // torch.sum over the other objects, ascending object order (prop_net.py:151-155)
      for (int b = 0; b < K; ++b)
        if (b != a) s += masks[(int64_t)b * P + i];
      others[(int64_t)a * P + i] = s;
    }
  }
}

__global__ void interleave_kernel(mivos_interleave_desc d, float *__restrict__ out, int N, int64_t P) {
  const int C = d.C;
  const int64_t total = (int64_t)N * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / P, px = i - n * P;
    float *o = out + i * C;
    for (int c0 = 0; c0 < C; c0 += 4) {
      f32x4 v;
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;
        t[j] = d.plane[c] ? d.plane[c][n * d.nstride[c] + px] : d.cval[c];
      }
      v.x = t[0]; v.y = t[1]; v.z = t[2]; v.w = t[3];
      *reinterpret_cast<f32x4 *>(o + c0) = v;
    }
  }
}

}  // namespace mivos

using namespace mivos;
#define ST ((hipStream_t)stream)

extern "C" int mivos_version(void) { return MIVOS_ABI_VERSION; }
extern "C" const char *mivos_last_error(void) { return err_buf(); }

extern "C" int mivos_device_check(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(MIVOS_ERR_DEVICE, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(MIVOS_ERR_DEVICE, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
  return MIVOS_OK;
}

extern "C" int mivos_maxpool3x3s2(const float *x, float *y, int N, int H, int W, int C, void *stream) {
  if (!x || !y || C % 4 || N < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "maxpool: bad arguments (C %% 4 != 0?)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, ST, x, y, N, H, W, C / 4, Ho, Wo);
  return check_launch("maxpool3x3s2");
}

extern "C" int mivos_upsample2x_add(const float *skip, int64_t skip_nstride, const float *up, float *out, int N, int h,
                                    int w, int C, void *stream) {
  if (!skip || !up || !out || C % 4 || (skip_nstride & 3)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "upsample2x_add: bad arguments");
  hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid_for((int64_t)N * 4 * h * w * (C / 4))), dim3(256), 0, ST, skip, skip_nstride, up, out, N, h, w, C / 4);
  return check_launch("upsample2x_add");
}

extern "C" int mivos_upsample2x_add_multi(const float *skip, int64_t skip_nstride, const float *up, float *out, void *raw_sh32,
                                          void *relu_sh32, int64_t a_nstride, int64_t a_rstride, int64_t a_pstride, int N, int h, int w,
                                          int C, void *stream) {
  if (!skip || !up || (!out && !raw_sh32 && !relu_sh32) || C % 4 || (skip_nstride & 3)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "upsample2x_add_multi: bad arguments");
  if ((raw_sh32 || relu_sh32) && ((C & 31) || ((a_nstride | a_rstride | a_pstride) & 31) || ((uintptr_t)raw_sh32 & 127) || ((uintptr_t)relu_sh32 & 127)))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "upsample2x_add_multi: SH32 outputs need C %% 32 == 0 and 128-byte aligned pixels");
  hipLaunchKernelGGL(upsample2x_add_multi_kernel, dim3(grid_for((int64_t)N * 4 * h * w * (C / 4))), dim3(256), 0, ST, skip, skip_nstride, up, out,
                     (float *)raw_sh32, (float *)relu_sh32, a_nstride, a_rstride, a_pstride, N, h, w, C / 4, xcd_contig());
  return check_launch("upsample2x_add_multi");
}

extern "C" int mivos_maxpool3x3s2_sh32(const float *x, void *y_sh32, int64_t y_nstride, int64_t y_rstride, int64_t y_pstride, int N, int H,
                                       int W, int C, void *stream) {
  if (!x || !y_sh32 || (C & 31) || N < 1 || ((y_nstride | y_rstride | y_pstride) & 31) || ((uintptr_t)y_sh32 & 127))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "maxpool3x3s2_sh32: bad arguments (C %% 32 != 0?)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_sh32_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, ST, x, (float *)y_sh32, y_nstride,
                     y_rstride, y_pstride, N, H, W, C / 4, Ho, Wo, xcd_contig());
  return check_launch("maxpool3x3s2_sh32");
}

extern "C" int mivos_tap_sum9(const float *t, const float *bias, float *out, int N, int H, int W, void *stream) {
  if (!t || !out || N < 1 || H < 1 || W < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "tap_sum9: bad arguments");
  hipLaunchKernelGGL(tap_sum9_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, ST, t, bias, out, N, H, W);
  return check_launch("tap_sum9");
}

extern "C" int mivos_resize_bilinear_nhwc(const float *x, float *y, int64_t y_nstride, int64_t y_pstride, int N, int h, int w, int H, int W,
                                          int C, void *stream) {
  if (!x || !y || N < 1 || (C & 3) || ((y_nstride | y_pstride) & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "resize_bilinear_nhwc: bad arguments (C %% 4, 16-byte alignment)");
  hipLaunchKernelGGL(resize_bilinear_nhwc_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, ST, x, y, y_nstride, y_pstride, N, h, w, H, W, C / 4);
  return check_launch("resize_bilinear_nhwc");
}

extern "C" int mivos_global_avgpool(const float *x, float *y, int N, int64_t P, int C, void *stream) {
  if (!x || !y || N < 1 || P < 1 || C < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "global_avgpool: bad arguments");
  hipLaunchKernelGGL(global_avgpool_kernel, dim3(cdiv(C, 64), N), dim3(256), 0, ST, x, y, P, C);
  return check_launch("global_avgpool");
}

extern "C" int mivos_dilate3x3(const float *x, float *y, int planes, int H, int W, void *stream) {
  if (!x || !y || planes < 1 || H < 1 || W < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "dilate3x3: bad arguments");
  hipLaunchKernelGGL(dilate3x3_kernel, dim3(grid_for((int64_t)planes * H * W)), dim3(256), 0, ST, x, y, planes, H, W);
  return check_launch("dilate3x3");
}

extern "C" int mivos_area_pool16(const float *x, float *y, int planes, int H, int W, void *stream) {
  if (!x || !y || H % 16 || W % 16) return fail(MIVOS_ERR_INVALID_ARGUMENT, "area_pool16: H, W must be multiples of 16");
  const int64_t cells = (int64_t)planes * (H / 16) * (W / 16);
  hipLaunchKernelGGL(area_pool16_kernel, dim3(grid_for(cells * 64)), dim3(256), 0, ST, x, y, planes, H, W);
  return check_launch("area_pool16");
}

extern "C" int mivos_resize_bilinear(const float *x, float *y, int planes, int h, int w, int H, int W, int act,
                                     void *stream) {
  if (!x || !y || planes < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "resize_bilinear: bad arguments");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((int64_t)planes * H * W)), dim3(256), 0, ST, x, y, planes, h, w, H, W, act);
  return check_launch("resize_bilinear");
}

extern "C" int mivos_aggregate_wbg(const float *prob, float *out, int K, int64_t P, int keep_bg, int hard, void *stream) {
  return launch_aggregate<true>(prob, out, nullptr, 1, K, P, keep_bg, hard, ST, "aggregate_wbg");
}

extern "C" int mivos_aggregate_sbg(const float *prob, float *out, int K, int64_t P, int keep_bg, int hard, void *stream) {
  return launch_aggregate<false>(prob, out, nullptr, 1, K, P, keep_bg, hard, ST, "aggregate_sbg");
}

extern "C" int mivos_aggregate_wbg_channel(const float *prob, float *logits, float *soft, int B, int K, int64_t P, int keep_bg,
                                           int hard, void *stream) {
  return launch_aggregate<true>(prob, soft, logits, B, K, P, keep_bg, hard, ST, "aggregate_wbg_channel");
}

extern "C" int mivos_argmax_u8(const float *prob, int64_t plane_stride, uint8_t *out, int planes, int64_t P, void *stream) {
  if (!prob || !out || planes < 1 || planes > 255 || (P & 3) || (plane_stride & 3)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "argmax_u8: P and plane stride must be multiples of 4, planes <= 255");
  hipLaunchKernelGGL(argmax_u8_kernel, dim3(grid_for(P / 4)), dim3(256), 0, ST, prob, plane_stride, out, planes, P);
  return check_launch("argmax_u8");
}

extern "C" int mivos_mask_diff(const float *mask, const float *prob, float *pos, float *neg, int64_t n, void *stream) {
  if (!mask || !prob || !pos || !neg) return fail(MIVOS_ERR_INVALID_ARGUMENT, "mask_diff: null pointer");
  hipLaunchKernelGGL(mask_diff_kernel, dim3(grid_for(n)), dim3(256), 0, ST, mask, prob, pos, neg, n);
  return check_launch("mask_diff");
}

extern "C" int mivos_sigmoid(const float *x, float *y, int64_t n, void *stream) {
  if (!x || !y) return fail(MIVOS_ERR_INVALID_ARGUMENT, "sigmoid: null pointer");
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(n)), dim3(256), 0, ST, x, y, n);
  return check_launch("sigmoid");
}

extern "C" int mivos_mask_others(const float *masks, float *others, int K, int64_t P, void *stream) {
  if (!masks || !others || K < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "mask_others: bad arguments");
  hipLaunchKernelGGL(mask_others_kernel, dim3(grid_for(P)), dim3(256), 0, ST, masks, others, K, P);
  return check_launch("mask_others");
}

extern "C" int mivos_interleave_planes(const mivos_interleave_desc *d, float *out, int N, int64_t P, void *stream) {
  if (!d || !out || d->C < 4 || d->C > 16 || d->C % 4) return fail(MIVOS_ERR_INVALID_ARGUMENT, "interleave_planes: C must be 4, 8, 12 or 16");
  hipLaunchKernelGGL(interleave_kernel, dim3(grid_for((int64_t)N * P)), dim3(256), 0, ST, *d, out, N, P);
  return check_launch("interleave_planes");
}
