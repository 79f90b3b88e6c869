// Kernel-side parameter block shared by the convolution kernels.
#pragma once
#include "common.h"

namespace mivos {

struct ConvP {
  const float *x, *w, *scale, *bias, *res;
  float *y, *y2;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, split, relu_in, relu_out;
  int log2Cin, M, Ktot, HoWo, tiles_n, vec_epi;
  int kt_split;        // split-K: K steps per slice (gridDim.y slices), 0 = off
  float *partial;      // split-K: [slice][M][Cout] fp32 partial sums
  void *ws; long long ws_bytes;
  long long x_ns, x_ps, y_ns, y_ps, y2_ns, y2_ps, r_ns, r_ps;
  long long x_rs, y_rs, y2_rs, r_rs;   // row strides (floats); dense tensors: W * pixel stride
  int x_border;                        // zero pixels guaranteed around every input image (precision 2 needs >= pad)
  int y_fmt, r_fmt;                    // 0: fp32, 1: SH32 (fp16 hi | lo lines per 32 channels, conv_f16x3_dma.hip)
  int dil;                             // tap spacing (atrous convolution), >= 1
  int share;                           // launch streams the caller keeps busy on this GPU (>= 1): a hint for launch geometry that must not change results
                                       // (round 6: the LDS-DMA kernels' split-K slice COUNT, i.e. the fp32 summation order, no longer looks at it; it only
                                       // chooses whether the slices run folded inside one workgroup - bit-identical)
  unsigned *status;                    // optional device word: bit 0 is set when an output of an f16x3 launch leaves the fp16 range (|y| > 65504)
};

// fp16-range guard of the f16x3 paths (precision 1 / 2): an output beyond 65504 becomes inf in the hi half of the NEXT layer's operand split, and the
// ReLUs / clamps downstream turn the resulting NaNs into finite numbers - silently.  The epilogues keep a running maximum of what they store and
// raise bit 0 of *p.status (mivos_conv_desc.status) for the host to find (ops.check_activation_range, once per interaction).
__device__ __forceinline__ float range_max(float amax, f32x4 v) {
  return fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void range_flag(const ConvP &p, float amax) {
  if (!p.status) return;
  const unsigned long long m = __ballot(amax > 65504.f);          // lanes of this wavefront that stored an out-of-range value
  if (m && (int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicOr(p.status, 1u);   // one atomic per offending wavefront (all blocks are 1-D)
}

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// SH32 accessors: 4 consecutive channels c..c+3 (c % 4 == 0) of the pixel whose fp32-equivalent float offset is `pix_off`
__device__ __forceinline__ f32x4 load_sh32x4(const float *base, long long pix_off, int c) {
  const unsigned char *q = reinterpret_cast<const unsigned char *>(base + pix_off) + (c >> 5) * 128 + (c & 31) * 2;
  const half4_t hi = *reinterpret_cast<const half4_t *>(q), lo = *reinterpret_cast<const half4_t *>(q + 64);
  return f32x4{(float)hi.x + (float)lo.x, (float)hi.y + (float)lo.y, (float)hi.z + (float)lo.z, (float)hi.w + (float)lo.w};
}
__device__ __forceinline__ void store_sh32x4(float *base, long long pix_off, int c, f32x4 v) {
  unsigned char *q = reinterpret_cast<unsigned char *>(base + pix_off) + (c >> 5) * 128 + (c & 31) * 2;
  half4_t hi, lo;
  hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
  lo.x = (_Float16)(v.x - (float)hi.x); lo.y = (_Float16)(v.y - (float)hi.y);
  lo.z = (_Float16)(v.z - (float)hi.z); lo.w = (_Float16)(v.w - (float)hi.w);
  *reinterpret_cast<half4_t *>(q) = hi;
  *reinterpret_cast<half4_t *>(q + 64) = lo;
}


// Tile selection shared by the fp32 and the fp16x3 implicit-GEMM kernels.
// 0: 128x128 (best MFMA:LDS ratio, needs >= ~1 workgroup per CU), 1: 64x64, 2: 128x32 (Cout <= 32),
// 3: 128x64 (Cout <= 64, many pixels), 4: Cout == 1 dot-product kernel.
inline int select_variant(int M, int Cout) {
  if (Cout == 1) return 4;
  if (Cout <= 32) return 2;
  if (Cout <= 64) return cdiv(M, 128) >= 200 ? 3 : 1;
  return (long long)cdiv(M, 128) * cdiv(Cout, 128) >= 200 ? 0 : 1;
}

// XCD-aware tile order (cdna guide T1, bijective form): hardware block b runs on XCD b % 8, each XCD has
// a private L2.  Give every XCD one contiguous run of logical tiles (n-tiles of one m-tile adjacent, then
// the next rows of the image) so the A rows shared by neighbouring tiles and by the 3x3 taps hit in L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  if (nwg < 16) return b;
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, pos = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// scale / bias of an epilogue, two ways.  FUSED: one fma (what the compiler made of `acc * scale + bias` in the in-kernel epilogues of rounds 1-5; now explicit).
// Not fused: multiply, round, add, round - what splitk_reduce_kernel has always done (its multiply and its add sat in different basic blocks) and what the
// reference's batch_norm does.  A layer's bits must not depend on HOW its K slices ran (split + reduce pass, or folded inside one workgroup: conv_f16x3_pp_kernel
// <..., FOLD>), so the folded kernel's epilogue uses the reduce pass's arithmetic; every other launch keeps the arithmetic it had (the golden vectors of the
// small networks, whose layers all split, are sensitive to it at the 5e-5 level: profiles/r06g_epilogue_rounding.txt).
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
  const float t = a * b;
  return t + c;
}
template <bool FUSED>
__device__ __forceinline__ float scale_bias(float a, float sc, float bi) {
  return FUSED ? __builtin_fmaf(a, sc, bi) : mul_then_add(a, sc, bi);
}

// ---- epilogues shared by the implicit-GEMM kernels ----------------------------------------------------
// acc tile layout (32x32 MFMA C/D): lane (n = lane&31, h = lane>>5), register r -> pixel row mfma32_row(r, lane).
template <int MT, int NT, bool FUSED = true>
__device__ __forceinline__ void epilogue_scalar(const f32x16 (&acc)[MT][NT], const ConvP &p, int m_base, int n_base, int lane) {
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n_base + j * 32 + (lane & 31);
    if (n >= p.Cout) continue;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float bi = p.bias ? p.bias[n] : 0.f;
    float *dst;
    long long d_ns, d_rs, d_ps;
    int dn;
    if (n < p.split) { dst = p.y; d_ns = p.y_ns; d_rs = p.y_rs; d_ps = p.y_ps; dn = n; }
    else { dst = p.y2; d_ns = p.y2_ns; d_rs = p.y2_rs; d_ps = p.y2_ps; dn = n - p.split; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + i * 32 + mfma32_row(r, lane);
        if (m >= p.M) continue;
        const int img = m / p.HoWo, pix = m - img * p.HoWo;
        const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
        float v = scale_bias<FUSED>(acc[i][j][r], sc, bi);
        if (p.res) v += p.res[(long long)img * p.r_ns + (long long)oh * p.r_rs + (long long)ow * p.r_ps + n];
        if (p.relu_out) v = fmaxf(v, 0.f);
        amax = fmaxf(amax, fabsf(v));
        dst[(long long)img * d_ns + (long long)oh * d_rs + (long long)ow * d_ps + dn] = v;
      }
    }
  }
  range_flag(p, amax);
}

// Vectorised epilogue: each 32x32 tile is transposed through a per-wave LDS scratch (32 x 36 floats) so
// that a lane owns 4 consecutive CHANNELS of one pixel: residual loads and output stores become 16-byte
// accesses, 8 lanes per 128-B line (the memory-bound 1x1 expansion convs were store-issue bound with one
// dword per lane: 1.5 TB/s).  Needs Cout, split and all strides % 4 == 0 (p.vec_epi, checked on the host).
constexpr int EPI_PITCH = 36;
template <int MT, int NT, int IB = (MT > 2 ? 1 : MT), bool FUSED = true>   // IB: row tiles per residual-prefetch block (bounds the live registers)
__device__ __forceinline__ void epilogue_vec(const f32x16 (&acc)[MT][NT], float *scratch, const ConvP &p, int m_base,
                                             int n_base, int lane) {
  constexpr bool PREFETCH = MT <= 2;       // 256x256 tiles (128 accumulator VGPRs) have no registers to spare for it
  const int prow0 = lane >> 3, c4 = (lane & 7) * 4;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n_base + j * 32 + c4;
    const bool nok = n < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + n);
    if (nok && p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
    float *dst;
    long long d_ns, d_rs, d_ps;
    int dn, d_fmt;
    if (n < p.split) { dst = p.y; d_ns = p.y_ns; d_rs = p.y_rs; d_ps = p.y_ps; dn = n; d_fmt = p.y_fmt; }
    else { dst = p.y2; d_ns = p.y2_ns; d_rs = p.y2_rs; d_ps = p.y2_ps; dn = n - p.split; d_fmt = 0; }
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += IB) {
      // output offsets + ALL residual loads of IB row tiles in flight before the first one is needed: one memory latency
      // per block instead of one per 8 pixel rows (with one workgroup per CU nothing else hides them)
      long long yo[IB][4];
      f32x4 rr[IB][4];
      bool ok[IB][4];
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = m_base + (i0 + ii) * 32 + ps * 8 + prow0;
          ok[ii][ps] = m < p.M && nok;
          const int mm = m < p.M ? m : 0;
          const int img = mm / p.HoWo, pix = mm - img * p.HoWo;
          const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
          yo[ii][ps] = (long long)img * d_ns + (long long)oh * d_rs + (long long)ow * d_ps;
          rr[ii][ps] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (PREFETCH && p.res && ok[ii][ps]) {
            const long long ro = (long long)img * p.r_ns + (long long)oh * p.r_rs + (long long)ow * p.r_ps;
            rr[ii][ps] = p.r_fmt ? load_sh32x4(p.res, ro, n) : *reinterpret_cast<const f32x4 *>(p.res + ro + n);
          }
        }
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[mfma32_row(r, lane) * EPI_PITCH + (lane & 31)] = acc[i0 + ii][j][r];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          f32x4 v = *reinterpret_cast<const f32x4 *>(scratch + (ps * 8 + prow0) * EPI_PITCH + c4);
          if (ok[ii][ps]) {
            v.x = scale_bias<FUSED>(v.x, sc.x, bi.x); v.y = scale_bias<FUSED>(v.y, sc.y, bi.y); v.z = scale_bias<FUSED>(v.z, sc.z, bi.z); v.w = scale_bias<FUSED>(v.w, sc.w, bi.w);
            if (!PREFETCH && p.res) {
              const int m = m_base + (i0 + ii) * 32 + ps * 8 + prow0;
              const int img = m / p.HoWo, pix = m - img * p.HoWo;
              const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
              const long long ro = (long long)img * p.r_ns + (long long)oh * p.r_rs + (long long)ow * p.r_ps;
              rr[ii][ps] = p.r_fmt ? load_sh32x4(p.res, ro, n) : *reinterpret_cast<const f32x4 *>(p.res + ro + n);
            }
            v.x += rr[ii][ps].x; v.y += rr[ii][ps].y; v.z += rr[ii][ps].z; v.w += rr[ii][ps].w;
            if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            amax = range_max(amax, v);
            if (d_fmt) store_sh32x4(dst, yo[ii][ps], dn, v);
            else *reinterpret_cast<f32x4 *>(dst + yo[ii][ps] + dn) = v;
          }
        }
      }
    }
  }
  range_flag(p, amax);
}

// SH32-output epilogue (y_fmt == 1, single destination): a lane owns 8 consecutive channels of one pixel, so the fp16 hi
// and lo parts go out as one 16-byte store each (4 lanes cover the 64 B hi + 64 B lo halves of a 128-byte line) and an
// SH32 residual comes in as two 16-byte loads.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <int MT, int NT, int IB = (MT > 2 ? 1 : MT), bool FUSED = true>
__device__ __forceinline__ void epilogue_sh32(const f32x16 (&acc)[MT][NT], float *scratch, const ConvP &p, int m_base, int n_base, int lane) {
  constexpr bool PREFETCH = MT <= 2;
  const int prow0 = lane >> 2, c8 = (lane & 3) * 8;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n_base + j * 32 + c8;
    const bool nok = n < p.Cout;
    float sc[8], bi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { sc[q] = 1.f; bi[q] = 0.f; }
    if (nok && p.scale) { const f32x4 a = *reinterpret_cast<const f32x4 *>(p.scale + n), b = *reinterpret_cast<const f32x4 *>(p.scale + n + 4);
      sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w; }
    if (nok && p.bias) { const f32x4 a = *reinterpret_cast<const f32x4 *>(p.bias + n), b = *reinterpret_cast<const f32x4 *>(p.bias + n + 4);
      bi[0] = a.x; bi[1] = a.y; bi[2] = a.z; bi[3] = a.w; bi[4] = b.x; bi[5] = b.y; bi[6] = b.z; bi[7] = b.w; }
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += IB) {
      // output offsets + all residual loads of IB row tiles in flight at once (see epilogue_vec)
      long long yo[IB][2];
      float rr[IB][2][8];
      bool ok[IB][2];
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int m = m_base + (i0 + ii) * 32 + ps * 16 + prow0;
          ok[ii][ps] = m < p.M && nok;
          const int mm = m < p.M ? m : 0;
          const int img = mm / p.HoWo, pix = mm - img * p.HoWo;
          const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
          yo[ii][ps] = (long long)img * p.y_ns + (long long)oh * p.y_rs + (long long)ow * p.y_ps;
#pragma unroll
          for (int q = 0; q < 8; ++q) rr[ii][ps][q] = 0.f;
          if (p.res && ok[ii][ps] && (PREFETCH || IB * 2 == 2)) {     // MT > 2: one pass ahead only (register budget)
            const long long ro = (long long)img * p.r_ns + (long long)oh * p.r_rs + (long long)ow * p.r_ps;
            if (p.r_fmt) {
              const unsigned char *rq = reinterpret_cast<const unsigned char *>(p.res + ro) + (n >> 5) * 128 + (n & 31) * 2;
              const half8_t rh = *reinterpret_cast<const half8_t *>(rq), rl = *reinterpret_cast<const half8_t *>(rq + 64);
#pragma unroll
              for (int q = 0; q < 8; ++q) rr[ii][ps][q] = (float)rh[q] + (float)rl[q];
            } else {
              const f32x4 a = *reinterpret_cast<const f32x4 *>(p.res + ro + n), b = *reinterpret_cast<const f32x4 *>(p.res + ro + n + 4);
              rr[ii][ps][0] = a.x; rr[ii][ps][1] = a.y; rr[ii][ps][2] = a.z; rr[ii][ps][3] = a.w;
              rr[ii][ps][4] = b.x; rr[ii][ps][5] = b.y; rr[ii][ps][6] = b.z; rr[ii][ps][7] = b.w;
            }
          }
        }
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[mfma32_row(r, lane) * EPI_PITCH + (lane & 31)] = acc[i0 + ii][j][r];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int prow = ps * 16 + prow0;
          const f32x4 v0 = *reinterpret_cast<const f32x4 *>(scratch + prow * EPI_PITCH + c8);
          const f32x4 v1 = *reinterpret_cast<const f32x4 *>(scratch + prow * EPI_PITCH + c8 + 4);
          if (ok[ii][ps]) {
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            half8_t hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              float t = scale_bias<FUSED>(v[q], sc[q], bi[q]);
              t += rr[ii][ps][q];
              t = p.relu_out ? fmaxf(t, 0.f) : t;
              amax = fmaxf(amax, fabsf(t));
              hi[q] = (_Float16)t;
              lo[q] = (_Float16)(t - (float)hi[q]);
            }
            unsigned char *yq = reinterpret_cast<unsigned char *>(p.y + yo[ii][ps]) + (n >> 5) * 128 + (n & 31) * 2;
            *reinterpret_cast<half8_t *>(yq) = hi;
            *reinterpret_cast<half8_t *>(yq + 64) = lo;
          }
        }
      }
    }
  }
  range_flag(p, amax);
}

// split-K: output element group (pixel m, channels n..n+3) = act(sum_s partial[s] * scale + bias + res), slices summed in
// ascending order (deterministic whichever workgroup does it)
__device__ __forceinline__ void splitk_finish4(const ConvP &p, int n_slices, int m, int n) {
  const int c4n = p.Cout >> 2;
  const f32x4 *src = reinterpret_cast<const f32x4 *>(p.partial + (long long)m * p.Cout + n);
  f32x4 v = src[0];
  for (int s = 1; s < n_slices; ++s) {
    const f32x4 t = src[(long long)s * p.M * c4n];
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const int img = m / p.HoWo, pix = m - img * p.HoWo;
  const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
  // the SAME arithmetic, operation for operation, as the epilogues of the folded kernel (epilogue_* <..., FUSED = false>): sum * (scale | 1), rounded, + (bias | 0),
  // rounded, + (residual | 0), so that a layer gives the same bits whether its K slices ran as separate workgroups + this pass or one after the other inside
  // one workgroup (conv_f16x3_pp_kernel<..., FOLD>: what a launch does when other streams share the chip)
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f}, rr = {0.f, 0.f, 0.f, 0.f};
  if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + n);
  if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
  if (p.res) {
    const long long ro = (long long)img * p.r_ns + (long long)oh * p.r_rs + (long long)ow * p.r_ps;
    rr = p.r_fmt ? load_sh32x4(p.res, ro, n) : *reinterpret_cast<const f32x4 *>(p.res + ro + n);
  }
  v.x = mul_then_add(v.x, sc.x, bi.x); v.y = mul_then_add(v.y, sc.y, bi.y); v.z = mul_then_add(v.z, sc.z, bi.z); v.w = mul_then_add(v.w, sc.w, bi.w);
  v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
  if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  range_flag(p, range_max(0.f, v));
  float *dst;
  long long d_ns, d_rs, d_ps;
  int dn, d_fmt;
  if (n < p.split) { dst = p.y; d_ns = p.y_ns; d_rs = p.y_rs; d_ps = p.y_ps; dn = n; d_fmt = p.y_fmt; }
  else { dst = p.y2; d_ns = p.y2_ns; d_rs = p.y2_rs; d_ps = p.y2_ps; dn = n - p.split; d_fmt = 0; }
  const long long yo = (long long)img * d_ns + (long long)oh * d_rs + (long long)ow * d_ps;
  if (d_fmt) store_sh32x4(dst, yo, dn, v);
  else *reinterpret_cast<f32x4 *>(dst + yo + dn) = v;
}

// fills ConvP from the public descriptor after validating it; returns a status code
int conv_params_from_desc(const mivos_conv_desc *d, ConvP &p);
int launch_conv_f16x3(ConvP &p, hipStream_t st);
int select_variant_f16x3(int M, int Cout);
int launch_conv_f16x3_dma(ConvP &p, hipStream_t st);
int launch_splitk_reduce(ConvP &p, int slices, hipStream_t st);   // p.partial [slices][M][Cout] -> y   // precision 2: SH32 input, LDS-DMA staging

}  // namespace mivos
