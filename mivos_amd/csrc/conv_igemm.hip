// Fused NHWC convolution as an implicit GEMM on the CDNA4 fp32 matrix cores.
//
//   C[M = N*Ho*Wo pixels][Cout] = A[M][K = KH*KW*Cin] x B[K][Cout],   B = OHWI weights (K contiguous)
//
// Why fp32 MFMA (v_mfma_f32_32x32x2_f32): the parity budget of this path is |dlogit| <= 1e-3 and
// bf16/fp16 operands miss it by 16-100x (SURVEY.md §7 hard part 1); the f32 MFMA is bit-for-bit an
// fmaf chain at 157 TF/s peak, so it is the exact path every other variant is checked against.
//
// Tiling (wave64, 4 waves / workgroup): block tile BM x BN x 32, wave tile (BM/WGM) x (BN/WGN) made
// of 32x32 MFMA tiles.  Both operands are staged through LDS with K contiguous and a 36-float row
// pitch: a lane (i = lane&31, h = lane>>5) fetches its fragment with ONE ds_read_b128 at
// row i, k = 8*kk + 4*h (conflict-free for the 16-lane b128 groups, MI355X_MICROARCH §LDS) and
// feeds 4 MFMAs from it (the k-permutation is the same for A and B, so the product is unchanged).
// Global->LDS goes through registers (float4 per lane, one 128-B line per 8 lanes) because conv
// zero-padding needs per-element predication; tile k+1 is fetched while tile k is multiplied.
// Epilogue (fused): * scale[c] + bias[c] (+ residual) (ReLU) and a two-destination channel split.
#include "conv_common.h"

namespace mivos {

constexpr int BK = 32;   // k-chunk (floats) per pipeline stage
constexpr int LDK = 36;  // LDS row pitch in floats (32 + 4 pad)

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int A_LD = BM / 32, B_LD = BN / 32;  // float4 loads per thread and stage
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int STAGE = (BM + BN) * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;

  // ---- loader state: thread -> (row lrow + 32 j, float4 k4 of the 32-wide chunk)
  const int k4 = tid & 7, lrow = tid >> 3;
  const float *a_base[A_LD];
  int a_ih0[A_LD], a_iw0[A_LD];
  bool a_ok[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    int m = m0 + lrow + 32 * j;
    a_ok[j] = m < p.M;
    int mm = a_ok[j] ? m : 0;
    int n = mm / p.HoWo, rem = mm - n * p.HoWo;
    int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    a_ih0[j] = oh * p.stride - p.pad;
    a_iw0[j] = ow * p.stride - p.pad;
    a_base[j] = p.x + (long long)n * p.x_ns;
  }
  const float *b_ptr[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    int n = n0 + lrow + 32 * j;
    b_ok[j] = n < p.Cout;
    b_ptr[j] = p.w + (long long)(b_ok[j] ? n : 0) * p.Ktot;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 ra[A_LD], rb[B_LD];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto gload = [&](int k0) {
    const int k = k0 + 4 * k4;
    const bool kok = k < p.Ktot;
    const int tap = k >> p.log2Cin, c = k & (p.Cin - 1);
    int kh, kw;
    if (p.KW == 1) { kh = tap; kw = 0; }            // (KH == KW on this path)
    else if (p.KW == 3) { kh = tap / 3; kw = tap - 3 * kh; }
    else { kh = tap / p.KW; kw = tap - p.KW * kh; }
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int ih = a_ih0[j] + kh * p.dil, iw = a_iw0[j] + kw * p.dil;
      const bool ok = a_ok[j] && kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      f32x4 v = zero4;
      if (ok) v = *reinterpret_cast<const f32x4 *>(a_base[j] + ((long long)ih * p.W + iw) * p.x_ps + c);
      if (p.relu_in) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      f32x4 v = zero4;
      if (b_ok[j] && kok) v = *reinterpret_cast<const f32x4 *>(b_ptr[j] + k);
      rb[j] = v;
    }
  };
  auto lwrite = [&](int buf) {
    float *A = lds + buf * STAGE, *B = A + BM * LDK;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) *reinterpret_cast<f32x4 *>(A + (lrow + 32 * j) * LDK + 4 * k4) = ra[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) *reinterpret_cast<f32x4 *>(B + (lrow + 32 * j) * LDK + 4 * k4) = rb[j];
  };

  const int nk = (p.Ktot + BK - 1) / BK;
  gload(0);
  lwrite(0);
  __syncthreads();
  const int frag_off = (lane & 31) * LDK + 4 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const float *A = lds + (kt & 1) * STAGE + wm * TM * LDK + frag_off;
    const float *B = lds + (kt & 1) * STAGE + (BM + wn * TN) * LDK + frag_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4 *>(A + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int i = 0; i < NT; ++i) b[i] = *reinterpret_cast<const f32x4 *>(B + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lwrite((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue (LDS stages are dead: reuse them as per-wave transpose scratch)
  if (p.vec_epi) {
    __syncthreads();
    epilogue_vec<MT, NT>(acc, lds + wave * 32 * EPI_PITCH, p, m0 + wm * TM, n0 + wn * TN, lane);
  } else {
    epilogue_scalar<MT, NT>(acc, p, m0 + wm * TM, n0 + wn * TN, lane);
  }
}

// ---- Cout == 1 (decoder.pred 256->1, FusionNet final 32->1): a GEMM would waste 31/32 of every MFMA.
// LPP lanes share one output pixel; per tap they read the pixel's Cin floats as LPP consecutive float4
// (one full line per 8 lanes), 8 rounds, then a butterfly over the LPP lanes.  Weights (<= 9*256
// floats) live in LDS.
template <int LPP, int QPL>   // LPP lanes per pixel, QPL float4 per lane and tap: Cin = 4 * LPP * QPL
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  for (int i = threadIdx.x; i < p.Ktot; i += 256) wl[i] = p.w[i];
  __syncthreads();
  const int sub = threadIdx.x % LPP;
  // XCD-aware block order: consecutive pixel chunks (image rows) stay on one XCD so that the KH*KW tap
  // re-reads hit its L2 (PMC before: 1010 MB fetched per launch for 133 MB of input)
  const long long m = ((long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x) / LPP;
  const bool ok = m < p.M;
  const int mm = ok ? (int)m : 0;
  const int img = mm / p.HoWo, pix = mm - img * p.HoWo;
  const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
  const float *xb = p.x + (long long)img * p.x_ns;
  float acc = 0.f;
  for (int kh = 0; kh < p.KH; ++kh) {
    const int ih = oh * p.stride - p.pad + kh * p.dil;
    if ((unsigned)ih >= (unsigned)p.H) continue;
    for (int kw = 0; kw < p.KW; ++kw) {
      const int iw = ow * p.stride - p.pad + kw * p.dil;
      if ((unsigned)iw >= (unsigned)p.W) continue;
      const f32x4 *px = reinterpret_cast<const f32x4 *>(xb + ((long long)ih * p.W + iw) * p.x_ps) + sub;
      const f32x4 *pw = reinterpret_cast<const f32x4 *>(wl + (kh * p.KW + kw) * p.Cin) + sub;
#pragma unroll
      for (int q = 0; q < QPL; ++q) {
        f32x4 v = px[q * LPP], w4 = pw[q * LPP];
        if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        acc = fmaf(v.x, w4.x, acc); acc = fmaf(v.y, w4.y, acc);
        acc = fmaf(v.z, w4.z, acc); acc = fmaf(v.w, w4.w, acc);
      }
    }
  }
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (ok && sub == 0) {
    float v = acc * (p.scale ? p.scale[0] : 1.f) + (p.bias ? p.bias[0] : 0.f);
    if (p.res) v += p.res[(long long)img * p.r_ns + (long long)pix * p.r_ps];
    if (p.relu_out) v = fmaxf(v, 0.f);
    p.y[(long long)img * p.y_ns + (long long)pix * p.y_ps] = v;
  }
}

template <int BM, int BN, int WGM, int WGN>
static int launch_igemm(ConvP &p, hipStream_t st) {
  const int tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  const size_t lds = 2ull * (BM + BN) * LDK * sizeof(float);
  auto kern = conv_igemm_kernel<BM, BN, WGM, WGN>;
  static std::atomic<uint64_t> attr_mask{0};  // per instantiation, one bit per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_mask, "conv_igemm")) return rc;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(256), lds, st, p);
  return check_launch("conv_igemm");
}

}  // namespace mivos

using namespace mivos;

namespace mivos {
int conv_params_from_desc(const mivos_conv_desc *d, ConvP &p) {
  if (!d || !d->x || !d->w || !d->y) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: null pointer");
  // precision 0 / 1 decode the K index with shifts (Cin a power of two); precision 2 walks 32-channel slabs (Cin % 32 == 0)
  if (d->precision == 2 ? (d->Cin < 32 || (d->Cin & 31)) : (d->Cin < 4 || (d->Cin & (d->Cin - 1))))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: Cin=%d must be a power of two >= 4 (a multiple of 32 for precision 2)", d->Cin);
  if (d->KH != d->KW || d->KH < 1 || d->stride < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: unsupported kernel %dx%d stride %d", d->KH, d->KW, d->stride);
  const int dil = d->dilation > 1 ? d->dilation : 1;
  if (d->Ho != (d->H + 2 * d->pad - dil * (d->KH - 1) - 1) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - dil * (d->KW - 1) - 1) / d->stride + 1)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: Ho/Wo inconsistent with H/W/pad/stride/dilation");
  if (dil > 1 && d->precision == 2) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: dilated convolutions run on the precision 0 / 1 kernels");
  if ((d->x_pstride & 3) || (d->x_nstride & 3) || ((uintptr_t)d->x & 15) || ((uintptr_t)d->w & 15))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: x / w must be 16-byte aligned with strides %% 4 == 0");
  if (d->N < 1 || d->Cout < 1 || (long long)d->N * d->Ho * d->Wo > 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: bad N/Cout");
  const bool dual = d->y2 != nullptr && d->split < d->Cout;
  p.x = d->x; p.w = d->w; p.scale = d->scale; p.bias = d->bias; p.res = d->res; p.y = d->y; p.y2 = d->y2;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo; p.dil = dil;
  p.split = dual ? d->split : d->Cout;
  p.relu_in = d->relu_in; p.relu_out = d->relu_out;
  p.log2Cin = __builtin_ctz(d->Cin);
  p.HoWo = d->Ho * d->Wo; p.M = d->N * p.HoWo; p.Ktot = d->KH * d->KW * d->Cin; p.tiles_n = 1;
  p.x_ns = d->x_nstride; p.x_ps = d->x_pstride; p.y_ns = d->y_nstride; p.y_ps = d->y_pstride;
  p.y2_ns = d->y2_nstride; p.y2_ps = d->y2_pstride; p.r_ns = d->res_nstride; p.r_ps = d->res_pstride;
  p.kt_split = 0; p.partial = nullptr; p.ws = d->workspace; p.ws_bytes = d->workspace_bytes;
  p.x_rs = d->x_rstride ? d->x_rstride : (long long)d->W * d->x_pstride;
  p.y_rs = d->y_rstride ? d->y_rstride : (long long)d->Wo * d->y_pstride;
  p.y2_rs = (long long)d->Wo * d->y2_pstride;
  p.r_rs = d->res_rstride ? d->res_rstride : (long long)d->Wo * d->res_pstride;
  p.x_border = d->x_border; p.y_fmt = d->y_format; p.r_fmt = d->res_format;
  p.share = d->chip_share > 1 ? (d->chip_share > 8 ? 8 : d->chip_share) : 1;
  p.status = d->precision != 0 ? d->status : nullptr;          // exact fp32 has no fp16 range to leave
  if ((d->x_format != 0) != (d->precision == 2)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: x_format 1 (SH32) goes with precision 2 and only with it");
  if (d->precision != 2 && p.x_rs != (long long)d->W * d->x_pstride) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: precision 0/1 read dense input rows");
  if (d->precision != 2 && (p.y_rs != (long long)d->Wo * d->y_pstride || (d->res && p.r_rs != (long long)d->Wo * d->res_pstride)))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: precision 0/1 write / add dense rows");
  if (p.x_rs & 3) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: x row stride %% 4 != 0");
  auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
  p.vec_epi = !(p.Cout & 3) && !(p.split & 3) && !((p.y_ns | p.y_ps | p.y_rs) & 3) && al16(p.y) && al16(p.scale) && al16(p.bias) &&
              (!dual || (!((p.y2_ns | p.y2_ps) & 3) && al16(p.y2))) && (!p.res || (!((p.r_ns | p.r_ps | p.r_rs) & 3) && al16(p.res)));
  if ((p.y_fmt || p.r_fmt) && (!p.vec_epi || (p.split & 31) || ((p.y_ps | p.y_rs | p.y_ns) & 31 && p.y_fmt) || (p.r_fmt && ((p.r_ps | p.r_rs | p.r_ns) & 31))))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: SH32 output / residual needs 32-channel groups, strides %% 32 == 0 and 16-byte aligned operands");
  if ((p.y_fmt || p.r_fmt) && d->precision != 2) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: SH32 output / residual is implemented by the precision-2 kernels only");
  return MIVOS_OK;
}
}  // namespace mivos

extern "C" int mivos_conv2d_fused(const mivos_conv_desc *d, void *stream) {
  ConvP p;
  int rc = conv_params_from_desc(d, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == 2) return launch_conv_f16x3_dma(p, st);
  if (d->precision == 1 && p.Cout > 1) return launch_conv_f16x3(p, st);

  if (p.Cout == 1) {
    if (p.Cin % 32) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: Cout=1 path needs Cin %% 32 == 0");
    const size_t lds = (size_t)p.Ktot * sizeof(float);
    const int blocks = cdiv((long long)p.M * 8, 256);         // 8 lanes per pixel: one 128-B line per tap
    switch (p.Cin) {
      case 32: hipLaunchKernelGGL((conv_cout1_kernel<8, 1>), dim3(blocks), dim3(256), lds, st, p); break;
      case 64: hipLaunchKernelGGL((conv_cout1_kernel<8, 2>), dim3(blocks), dim3(256), lds, st, p); break;
      case 128: hipLaunchKernelGGL((conv_cout1_kernel<8, 4>), dim3(blocks), dim3(256), lds, st, p); break;
      case 256: hipLaunchKernelGGL((conv_cout1_kernel<8, 8>), dim3(blocks), dim3(256), lds, st, p); break;
      default: return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d: Cout=1 path supports Cin in {32,64,128,256}");
    }
    return check_launch("conv_cout1");
  }
  switch (select_variant(p.M, p.Cout)) {
    case 0: return launch_igemm<128, 128, 2, 2>(p, st);
    case 1: return launch_igemm<64, 64, 2, 2>(p, st);
    case 2: return launch_igemm<128, 32, 4, 1>(p, st);
    default: return launch_igemm<128, 64, 2, 2>(p, st);
  }
}

extern "C" int mivos_conv2d_variant(int M, int Cout) { return select_variant(M, Cout); }
