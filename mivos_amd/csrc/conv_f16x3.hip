// Error-compensated fp16 implicit-GEMM convolution for gfx950 ("f16x3").
//
// The exact-fp32 MFMA (conv_igemm.hip) tops out at 157 TFLOP/s; the 16-bit matrix cores are 16x faster but
// plain fp16/bf16 operands miss the 1e-3 logit budget by 16-100x.  Here every fp32 operand
// is split into two fp16 numbers, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits),
// and the product is accumulated in fp32 as
//        x*w  ~=  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi            (dropped: x_lo*w_lo ~ 2^-22 |x*w|)
// i.e. 3 x v_mfma_f32_32x32x16_f16 per 32x32x16 block = 16/3 = 5.3x the fp32-MFMA rate at fp32-class
// accuracy (measured: same error vs an fp64 run as the fp32 path).  Weights are split offline
// (mivos_pack_weights_f16x3) after an exact power-of-two scaling that lifts their lo parts out of the fp16
// subnormal range; activations stay fp32 in HBM and are split while they are staged into LDS, so no other
// kernel or tensor layout changes.
//
// Tiling: block tile BM x BN x 64, 4 waves, wave tile of 32x32 MFMA tiles.  LDS holds four fp16 images per
// stage (A_hi, A_lo, B_hi, B_lo; K contiguous, 72-half pitch => conflict-free ds_read_b128 fragment reads,
// one read = the 8 halves a lane feeds to one MFMA).  One LDS stage + register prefetch of the next stage
// (two barriers per 64-deep step); two workgroups per CU overlap each other's barrier/convert phases.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

namespace mivos {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// K order of the packed f16x3 weights ("taps inner"): for Cin % 32 == 0 the GEMM K axis is re-ordered to
//      k' = (c / 32) * (ntaps * 32) + tap * 32 + c % 32
// so that the KH*KW taps of one 32-channel slab are consecutive K steps: a workgroup then re-reads the same
// 128 B of every input pixel (shifted by one pixel per tap) within 9 consecutive steps and hits in L1/L2,
// instead of streaming the whole input once per tap (PMC before: 50 % L2 hit rate, 1.4 GB/launch from
// beyond L2 on the 129600x256x2304 decoder GEMM).
__device__ __forceinline__ void decode_k(int k, int cin, int log2cin, int ntaps, bool taps_inner, int &tap, int &c) {
  if (taps_inner) {
    const int s = k >> 5;                                   // 32-channel step index
    const int chunk = ntaps == 9 ? s / 9 : (ntaps == 1 ? s : s / ntaps);
    tap = s - chunk * ntaps;
    c = chunk * 32 + (k & 31);
  } else {
    tap = k >> log2cin;
    c = k & (cin - 1);
  }
}

constexpr int BKH = 64;     // k per stage
constexpr int PITCH = 72;   // halves per LDS row (64 + 8 pad = 144 B)

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(ConvP p, int kpad4) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int RP = 16;                       // rows staged per pass (256 threads / 16 float4 per row)
  constexpr int A_LD = BM / RP, B_LD = BN / RP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Ah = reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *Al = Ah + BM * PITCH;
  _Float16 *Bh = Al + BM * PITCH;
  _Float16 *Bl = Bh + BN * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;

  const int k4 = tid & 15, lrow = tid >> 4;
  const float *a_base[A_LD];
  int a_ih0[A_LD], a_iw0[A_LD];
  bool a_ok[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    int m = m0 + lrow + RP * j;
    a_ok[j] = m < p.M;
    int mm = a_ok[j] ? m : 0;
    int n = mm / p.HoWo, rem = mm - n * p.HoWo;
    int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    a_ih0[j] = oh * p.stride - p.pad;
    a_iw0[j] = ow * p.stride - p.pad;
    a_base[j] = p.x + (long long)n * p.x_ns;
  }
  const f32x4 *b_ptr[B_LD];   // 16-byte units: [hi0..3 | lo0..3] per 4 k
  bool b_ok[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    int n = n0 + lrow + RP * j;
    b_ok[j] = n < p.Cout;
    b_ptr[j] = reinterpret_cast<const f32x4 *>(p.w) + (long long)(b_ok[j] ? n : 0) * kpad4 + k4;
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

  const int ntaps = p.KH * p.KW;
  const bool taps_inner = (p.Cin & 31) == 0;
  auto gload = [&](int k0) {
    const int k = k0 + 4 * k4;
    const bool kok = k < p.Ktot;
    int tap, c;
    decode_k(k, p.Cin, p.log2Cin, ntaps, taps_inner, tap, c);
    int kh, kw;
    if (p.KW == 1) { kh = tap; kw = 0; }
    else if (p.KW == 3) { kh = tap / 3; kw = tap - 3 * kh; }
    else { kh = tap / p.KW; kw = tap - p.KW * kh; }
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int ih = a_ih0[j] + kh * p.dil, iw = a_iw0[j] + kw * p.dil;
      const bool ok = a_ok[j] && kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      f32x4 v = zero4;
      if (ok) v = *reinterpret_cast<const f32x4 *>(a_base[j] + ((long long)ih * p.W + iw) * p.x_ps + c);
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      f32x4 v = zero4;
      if (b_ok[j]) v = b_ptr[j][k0 >> 2];        // weights are zero padded up to kpad
      rb[j] = v;
    }
  };
  auto lwrite = [&]() {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      f32x4 v = ra[j];
      if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      h4 hi, lo;
      hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
      lo.x = (_Float16)(v.x - (float)hi.x); lo.y = (_Float16)(v.y - (float)hi.y);
      lo.z = (_Float16)(v.z - (float)hi.z); lo.w = (_Float16)(v.w - (float)hi.w);
      const int off = (lrow + RP * j) * PITCH + 4 * k4;
      *reinterpret_cast<h4 *>(Ah + off) = hi;
      *reinterpret_cast<h4 *>(Al + off) = lo;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int off = (lrow + RP * j) * PITCH + 4 * k4;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 h, l;
      h.x = rb[j].x; h.y = rb[j].y; l.x = rb[j].z; l.y = rb[j].w;
      *reinterpret_cast<f32x2 *>(Bh + off) = h;
      *reinterpret_cast<f32x2 *>(Bl + off) = l;
    }
  };

  // split-K (small-M layers): gridDim.y slices of kt_split K steps each; partial tiles go to p.partial
  const int nk_all = (p.Ktot + BKH - 1) / BKH;
  const int kt0 = p.kt_split ? blockIdx.y * p.kt_split : 0;
  const int nk = p.kt_split ? (kt0 + p.kt_split < nk_all ? kt0 + p.kt_split : nk_all) : nk_all;
  gload(kt0 * BKH);
  const int frag = (lane & 31) * PITCH + 8 * (lane >> 5);
  for (int kt = kt0; kt < nk; ++kt) {
    __syncthreads();                       // everybody is done reading the previous stage
    lwrite();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BKH);
    const _Float16 *pAh = Ah + wm * TM * PITCH + frag, *pAl = Al + wm * TM * PITCH + frag;
    const _Float16 *pBh = Bh + wn * TN * PITCH + frag, *pBl = Bl + wn * TN * PITCH + frag;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      h8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ah[i] = *reinterpret_cast<const h8 *>(pAh + i * 32 * PITCH + kk * 16);
        al[i] = *reinterpret_cast<const h8 *>(pAl + i * 32 * PITCH + kk * 16);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        bh[i] = *reinterpret_cast<const h8 *>(pBh + i * 32 * PITCH + kk * 16);
        bl[i] = *reinterpret_cast<const h8 *>(pBl + i * 32 * PITCH + kk * 16);
      }
      // small cross terms first, then the main product; tiles interleaved so that consecutive MFMAs
      // never wait on the same accumulator
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  }

  if (p.kt_split) {     // raw fp32 partial tile of this K slice: [slice][M][Cout], dense
    ConvP q = p;
    q.scale = q.bias = q.res = nullptr;
    q.status = nullptr;                    // partial sums are not outputs: the reduce kernel checks the finished values
    q.relu_out = 0;
    q.split = p.Cout;
    q.y = p.partial + (long long)blockIdx.y * p.M * p.Cout;
    q.y_ps = p.Cout;
    q.y_rs = (long long)p.Wo * p.Cout;
    q.y_ns = (long long)p.HoWo * p.Cout;
    q.y_fmt = 0;
    __syncthreads();
    epilogue_vec<MT, NT>(acc, reinterpret_cast<float *>(smem_raw) + wave * 32 * EPI_PITCH, q, m0 + wm * TM, n0 + wn * TN, lane);
    return;
  }
  if (p.vec_epi) {
    __syncthreads();   // LDS stages are dead: reuse them as per-wave transpose scratch
    epilogue_vec<MT, NT>(acc, reinterpret_cast<float *>(smem_raw) + wave * 32 * EPI_PITCH, p, m0 + wm * TM, n0 + wn * TN, lane);
  } else {
    epilogue_scalar<MT, NT>(acc, p, m0 + wm * TM, n0 + wn * TN, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined 8-wave variant for the big GEMMs (decoder, mask encoder at batch K).
// PMC on the 4-wave kernel above (profiles/): MFMA pipe 30 % busy, VALU 35 % busy, and the two never
// overlap inside a wave because "convert + ds_write" sits between the barriers and the MFMAs after them.
// Here: block tile up to 256x256, 8 waves (2 per SIMD), K step 32, TWO LDS stages; while the MFMAs of
// stage t run, the same wave converts and writes stage t+1 (segments interleaved between MFMA groups) and
// the global loads of stage t+2 are in flight -> one barrier per step.  BN = 256 covers all output
// channels of the decoder convs, so every activation element is split once per tap instead of twice, and
// the 128x64 wave tile halves the LDS bytes read per MFMA.
constexpr int BK2 = 32;     // k per stage
constexpr int PITCH2 = 40;  // halves per LDS row (32 + 8 pad = 80 B, conflict-free b128 fragment reads)

template <int BM, int BN, int WGM, int WGN, int ABL = 0>   // ABL: ablation switch for profiling builds only
__global__ __launch_bounds__(512) void conv_f16x3_pipe_kernel(ConvP p, int kpad4) {
  static_assert(WGM * WGN == 8, "8 waves per workgroup");
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int RP = 64;                       // rows staged per pass (512 threads / 8 float4 per row)
  constexpr int A_LD = BM / RP, B_LD = BN / RP;
  constexpr int NSEG = A_LD > B_LD ? A_LD : B_LD;
  constexpr int STAGE = 2 * (BM + BN) * PITCH2;   // halves per stage: Ah | Al | Bh | Bl
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *lds = reinterpret_cast<_Float16 *>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;

  const int k4 = tid & 7, lrow = tid >> 3;
  const float *a_base[A_LD];
  int a_ih0[A_LD], a_iw0[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    int m = m0 + lrow + RP * j;
    const bool ok = m < p.M;
    int mm = ok ? m : 0;
    int n = mm / p.HoWo, rem = mm - n * p.HoWo;
    int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    a_ih0[j] = ok ? oh * p.stride - p.pad : -0x40000000;     // out-of-range rows fail the bounds test
    a_iw0[j] = ow * p.stride - p.pad;
    a_base[j] = p.x + (long long)n * p.x_ns;
  }
  const f32x4 *b_ptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    int n = n0 + lrow + RP * j;
    b_ptr[j] = reinterpret_cast<const f32x4 *>(p.w) + (long long)(n < p.Cout ? n : p.Cout - 1) * kpad4 + k4;   // columns >= Cout are never stored
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 ra[A_LD], rb[B_LD];
  const bool uniform_tap = (p.Cin & 31) == 0;     // taps-inner K order: a 32-deep step is one tap of one slab
  const int ntaps = p.KH * p.KW;
  const float relu_floor = p.relu_in ? 0.f : -INFINITY;
  const int nk = (p.Ktot + BK2 - 1) / BK2;
  const int k_last = (nk - 1) * BK2;

  // Branch-free loaders (the main loop must stay ONE basic block so that the scheduler can put the
  // staging VALU / LDS writes / global loads in the shadow of the MFMAs): invalid taps / rows load from
  // the tensor base and are zeroed by a select, steps past the end re-load the last step.
  // Loads are issued PER STAGING PASS, right after the pass that frees their registers (see step()):
  // every load then has a whole K step (~3000 cycles) to land.  (Issuing them at the end of the step
  // left ~300 cycles before the first use: ablation = 2x slower than with the loads removed.)
  int ld_kh = 0, ld_kw = 0, ld_c = 0, ld_k0 = 0;
  bool ld_kok = true;
  auto gload_setup = [&](int k0) {
    k0 = k0 < k_last ? k0 : k_last;
    int tap;
    if (uniform_tap) { decode_k(k0, p.Cin, p.log2Cin, ntaps, true, tap, ld_c); ld_c += 4 * k4; }   // wave-uniform tap
    else decode_k(k0 + 4 * k4, p.Cin, p.log2Cin, ntaps, false, tap, ld_c);
    ld_kok = k0 + 4 * k4 < p.Ktot;
    if (p.KW == 1) { ld_kh = tap; ld_kw = 0; }
    else if (p.KW == 3) { ld_kh = tap / 3; ld_kw = tap - 3 * ld_kh; }
    else { ld_kh = tap / p.KW; ld_kw = tap - p.KW * ld_kh; }
    ld_k0 = k0;
  };
  auto gload_pass = [&](int s) {
    if (ABL == 1) {   // no global loads
      if (s < A_LD) ra[s < A_LD ? s : 0] = f32x4{1.f, 2.f, 3.f, 4.f};
      if (s < B_LD) rb[s < B_LD ? s : 0] = f32x4{1.f, 2.f, 3.f, 4.f};
      return;
    }
    if (s < A_LD) {
      const int j = s < A_LD ? s : 0;
      const int ih = a_ih0[j] + ld_kh * p.dil, iw = a_iw0[j] + ld_kw * p.dil;
      const bool ok = ld_kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      long long off = ok ? ((long long)ih * p.W + iw) * p.x_ps + ld_c : 0ll;
      if (ABL == 4) off = (off & 0xfff);          // all loads from one hot 16 KB window (cache-hit ablation)
      if (ABL == 5) off = ((long long)(m0 + lrow + RP * j) * p.Ktot + ld_k0) % ((long long)p.M * p.x_ps - 64) & ~3ll;   // streaming, no re-reads
      f32x4 v = *reinterpret_cast<const f32x4 *>(a_base[j] + off);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      ra[j] = v;
    }
    if (s < B_LD) {
      const int j = s < B_LD ? s : 0;
      rb[j] = b_ptr[j][ABL == 4 ? 0 : (ld_k0 >> 2)];
    }
  };
  auto gload = [&](int k0) {
    gload_setup(k0);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) gload_pass(s);
  };
  // convert + write one staging pass (A pass s and B pass s) of the prefetched registers into stage `st`
  auto lwrite_seg = [&](_Float16 *st, int s) {
    if (ABL == 2 || ABL == 6) return;   // no conversion / LDS writes
    if (s < A_LD) {
      f32x4 v = ra[s];
      v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
      h4 hi, lo;
      hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
      lo.x = (_Float16)__builtin_fmaf((float)hi.x, -1.f, v.x); lo.y = (_Float16)__builtin_fmaf((float)hi.y, -1.f, v.y);
      lo.z = (_Float16)__builtin_fmaf((float)hi.z, -1.f, v.z); lo.w = (_Float16)__builtin_fmaf((float)hi.w, -1.f, v.w);
      const int off = (lrow + RP * s) * PITCH2 + 4 * k4;
      *reinterpret_cast<h4 *>(st + off) = hi;
      *reinterpret_cast<h4 *>(st + BM * PITCH2 + off) = lo;
    }
    if (s < B_LD) {
      const int off = 2 * BM * PITCH2 + (lrow + RP * s) * PITCH2 + 4 * k4;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 h, l;
      h.x = rb[s].x; h.y = rb[s].y; l.x = rb[s].z; l.y = rb[s].w;
      *reinterpret_cast<f32x2 *>(st + off) = h;
      *reinterpret_cast<f32x2 *>(st + BN * PITCH2 + off) = l;
    }
  };
  const int frag = (lane & 31) * PITCH2 + 8 * (lane >> 5);
  // one 32-deep step on stage `cur`; when STAGE_NEXT, the prefetched registers are converted and written
  // into `nxt` in segments placed behind the MFMA groups
  auto step = [&](const _Float16 *cur, _Float16 *nxt, auto stage_next) {
    constexpr bool STAGE_NEXT = decltype(stage_next)::value;
    const _Float16 *pA = cur + wm * TM * PITCH2 + frag;
    const _Float16 *pB = cur + 2 * BM * PITCH2 + wn * TN * PITCH2 + frag;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h8 bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (ABL == 6) { bh[j] = h8{1, 2, 3, 4, 5, 6, 7, 8}; bl[j] = h8{8, 7, 6, 5, 4, 3, 2, 1}; continue; }   // no LDS reads
        bh[j] = *reinterpret_cast<const h8 *>(pB + j * 32 * PITCH2 + kk * 16);
        bl[j] = *reinterpret_cast<const h8 *>(pB + BN * PITCH2 + j * 32 * PITCH2 + kk * 16);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const h8 ah = ABL == 6 ? h8{1, 1, 2, 2, 3, 3, 4, 4} : *reinterpret_cast<const h8 *>(pA + i * 32 * PITCH2 + kk * 16);
        const h8 al = ABL == 6 ? h8{4, 4, 3, 3, 2, 2, 1, 1} : *reinterpret_cast<const h8 *>(pA + BM * PITCH2 + i * 32 * PITCH2 + kk * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
        if (STAGE_NEXT && kk * MT + i < NSEG) { lwrite_seg(nxt, kk * MT + i); gload_pass(kk * MT + i); }
      }
    }
    if (STAGE_NEXT) {
#pragma unroll
      for (int s = 2 * MT; s < NSEG; ++s) { lwrite_seg(nxt, s); gload_pass(s); }
    }
  };

  gload(0);
#pragma unroll
  for (int s = 0; s < NSEG; ++s) lwrite_seg(lds, s);
  gload(BK2);
  __syncthreads();
  for (int kt = 0; kt + 1 < nk; ++kt) {
    gload_setup((kt + 2) * BK2);          // addresses of the stage whose loads are issued inside step()
    step(lds + (kt & 1) * STAGE, lds + ((kt + 1) & 1) * STAGE, std::true_type{});
    __syncthreads();
  }
  step(lds + ((nk - 1) & 1) * STAGE, nullptr, std::false_type{});
  if (ABL == 3) {   // no epilogue (keep the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) t += acc[a][b][0];
    if (t == 12345.678f) p.y[0] = t;
    return;
  }

  if (p.vec_epi) {
    __syncthreads();   // LDS stages are dead: reuse them as per-wave transpose scratch
    epilogue_vec<MT, NT>(acc, reinterpret_cast<float *>(smem_raw) + wave * 32 * EPI_PITCH, p, m0 + wm * TM, n0 + wn * TN, lane);
  } else {
    epilogue_scalar<MT, NT>(acc, p, m0 + wm * TM, n0 + wn * TN, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// Direct 3x3 / stride 1 / pad 1 convolution for wide layers (decoder, KeyValue, encoder 3x3; Cin % 32 == 0).
// Ablations of the implicit-GEMM kernel above show it is bound by the vector-memory instruction path: the
// same input element is loaded, split and written to LDS once per TAP.  Here the block tile is an 8-row x
// 32-column patch of ONE image: per 32-channel slab the 10 x 34 input patch is loaded + split ONCE into LDS
// and the 9 taps read their A fragments from it at shifted addresses (80-byte pixel pitch, conflict-free);
// only the weights stream per tap (double-buffered LDS stages, loads issued per staging pass two steps
// ahead).  Per K step: 4 B loads + 0.67 A loads per thread instead of 8, 1/6 of the fp32->fp16 splitting.
// 8 waves = 2 (rows 0-3 / 4-7) x 4 (64 output channels each); wave tile = 4 image rows x 32 px x 64 ch.
template <int BN>
__global__ __launch_bounds__(512) void conv3x3_direct_kernel(ConvP p, int kpad4, int tiles_x, int tiles_y) {
  constexpr int TR = 8, TC = 32, PR = TR + 2, PC = TC + 2;
  constexpr int PPX = 40;                                  // halves per patch pixel (32 + 8 pad)
  constexpr int WGN = 4, TN = BN / WGN, NT = TN / 32, MT = 4;
  constexpr int B_LD = BN / 64;                            // weight staging passes (64 rows per pass)
  constexpr int NPL = (PR * PC * 8 + 511) / 512;           // float4 patch loads per thread and slab
  constexpr int PATCH = 2 * PR * PC * PPX;                 // halves: hi image | lo image
  constexpr int BSTAGE = 2 * BN * PITCH2;                  // halves: Bh | Bl
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Ph = reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *Pl = Ph + PR * PC * PPX;
  _Float16 *Bs = Ph + PATCH;                               // two weight stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = t % p.tiles_n; t /= p.tiles_n;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int img = t / tiles_y;
  const int y0 = ty * TR, x0 = tx * TC, n0 = nt * BN;
  const float *xb = p.x + (long long)img * p.x_ns;
  const float relu_floor = p.relu_in ? 0.f : -INFINITY;

  const int k4 = tid & 7, lrow = tid >> 3;
  const f32x4 *b_ptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int n = n0 + lrow + 64 * j;
    b_ptr[j] = reinterpret_cast<const f32x4 *>(p.w) + (long long)(n < p.Cout ? n : p.Cout - 1) * kpad4 + k4;
  }
  // patch element e -> (pixel, float4 channel group); offsets inside the image are slab independent
  long long p_off[NPL];
  bool p_ok[NPL];
#pragma unroll
  for (int l = 0; l < NPL; ++l) {
    const int e = tid + 512 * l;
    const int px = e >> 3, c4 = e & 7;
    const int pr = px / PC, pc = px - pr * PC;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    p_ok[l] = e < PR * PC * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    p_off[l] = p_ok[l] ? ((long long)iy * p.W + ix) * p.x_ps + 4 * c4 : 0ll;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 pa[NPL], rb[B_LD];
  const int n_slab = p.Cin >> 5, n_step = n_slab * 9;

  auto load_patch = [&](int slab) {
#pragma unroll
    for (int l = 0; l < NPL; ++l) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(xb + p_off[l] + (p_ok[l] ? 32 * slab : 0));
      v.x = p_ok[l] ? v.x : 0.f; v.y = p_ok[l] ? v.y : 0.f; v.z = p_ok[l] ? v.z : 0.f; v.w = p_ok[l] ? v.w : 0.f;
      pa[l] = v;
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int l = 0; l < NPL; ++l) {
      const int e = tid + 512 * l;
      if (e < PR * PC * 8) {
        f32x4 v = pa[l];
        v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
        h4 hi, lo;
        hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
        lo.x = (_Float16)(v.x - (float)hi.x); lo.y = (_Float16)(v.y - (float)hi.y);
        lo.z = (_Float16)(v.z - (float)hi.z); lo.w = (_Float16)(v.w - (float)hi.w);
        const int off = (e >> 3) * PPX + 4 * (e & 7);
        *reinterpret_cast<h4 *>(Ph + off) = hi;
        *reinterpret_cast<h4 *>(Pl + off) = lo;
      }
    }
  };
  auto load_b = [&](int step, int j) {                     // weights of K step `step` (= slab * 9 + tap), pass j
    const int st = step < n_step ? step : n_step - 1;
    rb[j] = b_ptr[j][st * 8];
  };
  auto write_b = [&](_Float16 *stg, int j) {
    const int off = (lrow + 64 * j) * PITCH2 + 4 * k4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 hh, ll;
    hh.x = rb[j].x; hh.y = rb[j].y; ll.x = rb[j].z; ll.y = rb[j].w;
    *reinterpret_cast<f32x2 *>(stg + off) = hh;
    *reinterpret_cast<f32x2 *>(stg + BN * PITCH2 + off) = ll;
  };

  // prologue: patch of slab 0, weights of step 0 in stage 0, weights of step 1 in registers
  load_patch(0);
#pragma unroll
  for (int j = 0; j < B_LD; ++j) load_b(0, j);
  write_patch();
#pragma unroll
  for (int j = 0; j < B_LD; ++j) write_b(Bs, j);
#pragma unroll
  for (int j = 0; j < B_LD; ++j) load_b(1, j);
  if (n_slab > 1) load_patch(1);
  __syncthreads();

  const int i = lane & 31, h = lane >> 5;
  int step = 0;
  for (int slab = 0; slab < n_slab; ++slab) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++step) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const _Float16 *cur = Bs + (step & 1) * BSTAGE;
      _Float16 *nxt = Bs + ((step + 1) & 1) * BSTAGE;
      const _Float16 *pB = cur + (wn * TN + i) * PITCH2 + 8 * h;
      const int aoff = ((wm * 4 + kh) * PC + i + kw) * PPX + 8 * h;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          bh[j] = *reinterpret_cast<const h8 *>(pB + j * 32 * PITCH2 + kk * 16);
          bl[j] = *reinterpret_cast<const h8 *>(pB + BN * PITCH2 + j * 32 * PITCH2 + kk * 16);
        }
#pragma unroll
        for (int r = 0; r < MT; ++r) {
          const h8 ah = *reinterpret_cast<const h8 *>(Ph + aoff + r * PC * PPX + kk * 16);
          const h8 al = *reinterpret_cast<const h8 *>(Pl + aoff + r * PC * PPX + kk * 16);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[r][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[r][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[r][j], 0, 0, 0);
          // weight staging of the next step + loads of the one after, one pass behind each MFMA group
          const int sg = kk * MT + r;
          if (sg < B_LD) { write_b(nxt, sg); load_b(step + 2, sg); }
        }
      }
      if (tap == 8 && slab + 1 < n_slab) {
        __syncthreads();                                   // everybody is done reading this slab's patch
        write_patch();
        if (slab + 2 < n_slab) load_patch(slab + 2);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: vectorised through per-wave LDS scratch (stages are dead)
  float *scratch = reinterpret_cast<float *>(smem_raw) + wave * 32 * EPI_PITCH;
  const int prow0 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * TN + j * 32 + c4;
    const bool nok = n < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + n);
    if (nok && p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
    float *dst;
    long long d_ns, d_ps;
    int dn;
    if (n < p.split) { dst = p.y; d_ns = p.y_ns; d_ps = p.y_ps; dn = n; }
    else { dst = p.y2; d_ns = p.y2_ns; d_ps = p.y2_ps; dn = n - p.split; }
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      const int y = y0 + wm * 4 + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) scratch[mfma32_row(q, lane) * EPI_PITCH + (lane & 31)] = acc[r][j][q];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int col = ps * 8 + prow0, x = x0 + col;
        f32x4 v = *reinterpret_cast<const f32x4 *>(scratch + col * EPI_PITCH + c4);
        if (y < p.H && x < p.W && nok) {
          const long long pix = (long long)y * p.W + x;
          v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
          if (p.res) {
            const f32x4 rr = *reinterpret_cast<const f32x4 *>(p.res + (long long)img * p.r_ns + pix * p.r_ps + n);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          *reinterpret_cast<f32x4 *>(dst + (long long)img * d_ns + pix * d_ps + dn) = v;
        }
      }
    }
  }
}

template <int BN>
static int launch_direct3x3(ConvP &p, hipStream_t st) {
  const int tiles_x = cdiv(p.W, 32), tiles_y = cdiv(p.H, 8);
  p.tiles_n = cdiv(p.Cout, BN);
  const size_t lds = (2ull * 10 * 34 * 40 + 2ull * 2 * BN * PITCH2) * sizeof(_Float16);
  auto kern = conv3x3_direct_kernel<BN>;
  static std::atomic<uint64_t> attr_mask{0};  // per instantiation, one bit per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_mask, "conv3x3_direct")) return rc;
  const int kpad4 = cdiv(p.Ktot, BKH) * BKH / 4;
  hipLaunchKernelGGL(kern, dim3(tiles_x * tiles_y * p.N * p.tiles_n), dim3(512), lds, st, p, kpad4, tiles_x, tiles_y);
  return check_launch("conv3x3_direct");
}

// ------------------------------------------------------------------------------------------------
// Direct 3x3 / stride 1 / pad 1 convolution with 32 output channels (FusionNet body, fusion_net.py:12-27).
// As an implicit GEMM these layers (N = 32, K = 144/288, 2 M pixels at 480p x 5 objects) are staging-bound:
// every input element is converted and written to LDS nine times for 32*3 MFMA MACs each.  Here a
// workgroup (8 waves) owns an 8-row x 32-column output tile: the 10 x 34 input patch is split to fp16
// hi/lo ONCE into LDS, all 9 taps read their A fragments from it at shifted addresses (80-byte pixel pitch,
// conflict-free), and the packed weights of all taps are LDS-resident.  Traffic per layer = one read of the
// input + one write of the output (+ residual): HBM-bound.
template <int CIN>
__global__ __launch_bounds__(512) void conv3x3_n32_direct_kernel(ConvP p, int kpad4, int tiles_x, int tiles_y, int n_tiles) {
  constexpr int TR = 8, TC = 32, PR = TR + 2, PC = TC + 2;
  constexpr int PPX = CIN + 8;                 // halves per patch pixel (pitch 80 B / 48 B)
  constexpr int PW = CIN + 8;                  // halves per weight row
  constexpr int KH16 = CIN / 16;               // MFMA k-blocks per tap
  constexpr int NLD = (PR * PC * (CIN / 4) + 511) / 512;     // float4 patch loads per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int PATCH = 2 * PR * PC * PPX;                      // halves per patch buffer: hi image | lo image
  _Float16 *P0 = reinterpret_cast<_Float16 *>(smem_raw);        // two patch buffers (double buffered)
  _Float16 *Wh = P0 + 2 * PATCH;                                // weights hi [9][32][PW]
  _Float16 *Wl = Wh + 9 * 32 * PW;                              // weights lo

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- weights once per (persistent) workgroup: packed [32][kpad4][hi0..3 | lo0..3] -> [tap][n][c] hi / lo
  for (int e = tid; e < 32 * 9 * CIN / 4; e += 512) {
    const int n = e / (9 * CIN / 4), q = e - n * (9 * CIN / 4);
    const f32x4 v = reinterpret_cast<const f32x4 *>(p.w)[(long long)n * kpad4 + q];
    const int k = 4 * q, tap = k / CIN, c = k - tap * CIN;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 hh, ll;
    hh.x = v.x; hh.y = v.y; ll.x = v.z; ll.y = v.w;
    *reinterpret_cast<f32x2 *>(Wh + (tap * 32 + n) * PW + c) = hh;
    *reinterpret_cast<f32x2 *>(Wl + (tap * 32 + n) * PW + c) = ll;
  }
  const float relu_floor = p.relu_in ? 0.f : -INFINITY;
  const int i = lane & 31, h = lane >> 5;
  f32x4 pre[NLD];
  // all patch loads of a tile are issued back to back (registers), converted + written later
  auto load_patch = [&](int tile) {
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int img = t / tiles_y;
    const float *xb = p.x + (long long)img * p.x_ns;
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
      const int e = tid + 512 * l;
      const int px = e / (CIN / 4), c4 = e - px * (CIN / 4);
      const int pr = px / PC, pc = px - pr * PC;
      const int iy = ty * TR + pr - 1, ix = tx * TC + pc - 1;
      const bool ok = e < PR * PC * (CIN / 4) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long long off = ok ? ((long long)iy * p.W + ix) * p.x_ps + 4 * c4 : 0ll;
      f32x4 v = *reinterpret_cast<const f32x4 *>(xb + off);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      pre[l] = v;
    }
  };
  auto write_patch = [&](_Float16 *Ph) {
    _Float16 *Pl = Ph + PR * PC * PPX;
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
      const int e = tid + 512 * l;
      if (e < PR * PC * (CIN / 4)) {
        const int px = e / (CIN / 4), c4 = e - px * (CIN / 4);
        f32x4 v = pre[l];
        v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
        h4 hi, lo;
        hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
        lo.x = (_Float16)(v.x - (float)hi.x); lo.y = (_Float16)(v.y - (float)hi.y);
        lo.z = (_Float16)(v.z - (float)hi.z); lo.w = (_Float16)(v.w - (float)hi.w);
        *reinterpret_cast<h4 *>(Ph + px * PPX + 4 * c4) = hi;
        *reinterpret_cast<h4 *>(Pl + px * PPX + 4 * c4) = lo;
      }
    }
  };

  // software pipeline over the tiles of this workgroup: while tile t is multiplied out of patch buffer
  // t&1, tile t+1 (loaded into registers during tile t-1) is converted into the other buffer and the loads of
  // tile t+2 are issued: one barrier per tile, every global load has a full tile of time to land.
  int tile = blockIdx.x;
  if (tile < n_tiles) { load_patch(tile); write_patch(P0); }
  if (tile + (int)gridDim.x < n_tiles) load_patch(tile + gridDim.x);
  __syncthreads();
  for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
    const _Float16 *Ph = P0 + (it & 1) * PATCH, *Pl = Ph + PR * PC * PPX;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = kh * 3 + kw;
        const int aoff = ((wave + kh) * PC + i + kw) * PPX + 8 * h;
        const int boff = (tap * 32 + i) * PW + 8 * h;
#pragma unroll
        for (int kb = 0; kb < KH16; ++kb) {
          const h8 ah = *reinterpret_cast<const h8 *>(Ph + aoff + 16 * kb);
          const h8 al = *reinterpret_cast<const h8 *>(Pl + aoff + 16 * kb);
          const h8 bh = *reinterpret_cast<const h8 *>(Wh + boff + 16 * kb);
          const h8 bl = *reinterpret_cast<const h8 *>(Wl + boff + 16 * kb);
          // weights are the MFMA "A" operand here: D[channel][pixel], so a lane ends up with 4 consecutive
          // CHANNELS of one pixel per register quad -> 16-byte residual loads / stores
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc, 0, 0, 0);
        }
      }
    }
    // epilogue: lane = pixel (column x0 + lane&31), registers 4g..4g+3 = channels 8g + 4h .. +3
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int img = t / tiles_y;
    const int y = ty * TR + wave;
    const int x = tx * TC + i;
    if (y < p.H && x < p.W) {
      const long long pix = (long long)y * p.W + x;
      const float *rp = p.res ? p.res + (long long)img * p.r_ns + pix * p.r_ps : nullptr;
      float *yp = p.y + (long long)img * p.y_ns + pix * p.y_ps;
      f32x4 out[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * h;
        const f32x4 s4 = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v;
        v.x = acc[4 * g] * s4.x + b4.x; v.y = acc[4 * g + 1] * s4.y + b4.y;
        v.z = acc[4 * g + 2] * s4.z + b4.z; v.w = acc[4 * g + 3] * s4.w + b4.w;
        if (rp) { const f32x4 r4 = *reinterpret_cast<const f32x4 *>(rp + c); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
        if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        out[g] = v;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(yp + 8 * g + 4 * h) = out[g];
    }
    if (tile + (int)gridDim.x < n_tiles) write_patch(P0 + ((it + 1) & 1) * PATCH);
    if (tile + 2 * (int)gridDim.x < n_tiles) load_patch(tile + 2 * gridDim.x);
    __syncthreads();
  }
}

template <int CIN>
static int launch_n32_direct(ConvP &p, hipStream_t st) {
  const int tiles_x = cdiv(p.W, 32), tiles_y = cdiv(p.H, 8);
  const size_t lds = (2ull * 2 * 10 * 34 * (CIN + 8) + 2ull * 9 * 32 * (CIN + 8)) * sizeof(_Float16);
  auto kern = conv3x3_n32_direct_kernel<CIN>;
  static std::atomic<uint64_t> attr_mask{0};  // per instantiation, one bit per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_mask, "conv3x3_n32_direct")) return rc;
  const int kpad4 = cdiv(p.Ktot, BKH) * BKH / 4;
  const int n_tiles = tiles_x * tiles_y * p.N;
  const int grid = n_tiles < 256 ? n_tiles : 256;            // persistent: one workgroup per CU walks the tiles
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, kpad4, tiles_x, tiles_y, n_tiles);
  return check_launch("conv3x3_n32_direct");
}

__global__ void pack_weights_f16x3_kernel(const float *__restrict__ w, _Float16 *__restrict__ out, int Cout, int Ktot,
                                          int Kpad, float mult, int cin, int ntaps) {
  const bool taps_inner = (cin & 31) == 0;
  const long long total = (long long)Cout * (Kpad / 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / (Kpad / 4)), q = (int)(i - (long long)n * (Kpad / 4));
    _Float16 *o = out + i * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 4 * q + t;                      // position in the kernel's K order
      int src = k;                                  // position in OHWI order: tap * cin + c
      if (taps_inner && k < Ktot) {
        const int s = k >> 5, chunk = s / ntaps, tap = s - chunk * ntaps;
        src = tap * cin + chunk * 32 + (k & 31);
      }
      const float v = k < Ktot ? w[(long long)n * Ktot + src] * mult : 0.f;
      const _Float16 hi = (_Float16)v;
      o[t] = hi;
      o[4 + t] = (_Float16)(v - (float)hi);
    }
  }
}

// split-K second pass: y = act(sum_s partial[s] * scale + bias + res), slices summed in ascending order
__global__ void splitk_reduce_kernel(ConvP p, int n_slices) {
  const int c4n = p.Cout >> 2;
  const long long total = (long long)p.M * c4n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(e / c4n), n = (int)(e - (long long)m * c4n) * 4;
    splitk_finish4(p, n_slices, m, n);
  }
}

int launch_splitk_reduce(ConvP &p, int slices, hipStream_t st) {
  const long long total = (long long)p.M * (p.Cout / 4);
  const int blocks = (int)(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, slices);
  return check_launch("splitk_reduce");
}

template <int BM, int BN, int WGM, int WGN>
static int launch_f16x3(ConvP &p, hipStream_t st) {
  const int tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  // split-K when the tile grid cannot fill the chip and K is long (batch-1 / 30x54 layers)
  const int nk = cdiv(p.Ktot, BKH), wgs = tiles_m * p.tiles_n;
  int slices = 1;
  if (p.vec_epi && p.ws && wgs < 400 && nk >= 8) {
    slices = (640 + wgs / 2) / wgs;                       // aim at ~2.5 workgroups per CU
    if (slices > 8) slices = 8;
    if (slices > nk / 4) slices = nk / 4;
    if ((long long)slices * p.M * p.Cout * 4 > p.ws_bytes) slices = 1;
  }
  p.kt_split = 0;
  if (slices > 1) { p.kt_split = cdiv(nk, slices); slices = cdiv(nk, p.kt_split); p.partial = (float *)p.ws; }
  const size_t lds = 2ull * (BM + BN) * PITCH * sizeof(_Float16);
  auto kern = conv_f16x3_kernel<BM, BN, WGM, WGN>;
  static std::atomic<uint64_t> attr_mask{0};  // per instantiation, one bit per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_mask, "conv_f16x3")) return rc;
  const int kpad4 = cdiv(p.Ktot, BKH) * BKH / 4;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n, slices), dim3(256), lds, st, p, kpad4);
  if (slices > 1) return launch_splitk_reduce(p, slices, st);
  return check_launch("conv_f16x3");
}

template <int BM, int BN, int WGM, int WGN, int ABL = 0>
static int launch_f16x3_pipe(ConvP &p, hipStream_t st) {
  const int tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  const size_t lds = 2ull * 2 * (BM + BN) * PITCH2 * sizeof(_Float16);
  auto kern = conv_f16x3_pipe_kernel<BM, BN, WGM, WGN, ABL>;
  static std::atomic<uint64_t> attr_mask{0};  // per instantiation, one bit per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_mask, "conv_f16x3_pipe")) return rc;
  const int kpad4 = cdiv(p.Ktot, BKH) * BKH / 4;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(512), lds, st, p, kpad4);
  return check_launch("conv_f16x3_pipe");
}

// f16x3 tile selection: 5: 256x256 / 8 waves, 6: 128x256 / 8 waves, 7: 128x128 / 8 waves, 8: 64x256 / 8 waves
// (all pipelined),
// else the 4-wave variants 0..3
int select_variant_f16x3(int M, int Cout) {
  if (Cout >= 224) {
    const long long t256 = (long long)cdiv(M, 256) * cdiv(Cout, 256), t128 = (long long)cdiv(M, 128) * cdiv(Cout, 256);
    if (t256 >= 384) return 5;
    if (t128 >= 200) return 6;
    if (Cout % 256 == 0 && (long long)cdiv(M, 64) * cdiv(Cout, 256) >= 200) return 8;
  }
  if (Cout >= 96 && (long long)cdiv(M, 128) * cdiv(Cout, 128) >= 200) return 7;
  return select_variant(M, Cout);
}

int launch_conv_f16x3(ConvP &p, hipStream_t st) {
  if (p.vec_epi && p.Cout == 32 && p.split == 32 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.dil == 1 && (p.Cin == 16 || p.Cin == 32))
    return p.Cin == 16 ? launch_n32_direct<16>(p, st) : launch_n32_direct<32>(p, st);
  // direct 3x3 (LDS patch reuse) for wide stride-1 layers with enough 8x32 tiles to fill the chip
  // Measured equal to the pipelined implicit GEMM on the decoder shapes (302 vs 307 TFLOP/s) although it
  // issues 45 % fewer vector-memory instructions and 40 % fewer LDS writes, i.e. neither of those bounds
  // the GEMM today; kept selectable (MIVOS_DIRECT3X3_MIN_TILES=<n>, 0 forces it) for the next tuning round.
  const char *mt = getenv("MIVOS_DIRECT3X3_MIN_TILES");
  const long long min_tiles = mt ? atoll(mt) : (1ll << 60);
  if (p.vec_epi && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.dil == 1 && (p.Cin & 31) == 0 && p.Cout >= 224 &&
      (long long)p.N * cdiv(p.H, 8) * cdiv(p.W, 32) * cdiv(p.Cout, 256) >= min_tiles)
    return launch_direct3x3<256>(p, st);
  switch (select_variant_f16x3(p.M, p.Cout)) {
    case 5: {
      static const int abl = getenv("MIVOS_ABL") ? atoi(getenv("MIVOS_ABL")) : 0;   // profiling only
      if (abl == 1) return launch_f16x3_pipe<256, 256, 2, 4, 1>(p, st);
      if (abl == 2) return launch_f16x3_pipe<256, 256, 2, 4, 2>(p, st);
      if (abl == 3) return launch_f16x3_pipe<256, 256, 2, 4, 3>(p, st);
      if (abl == 4) return launch_f16x3_pipe<256, 256, 2, 4, 4>(p, st);
      if (abl == 6) return launch_f16x3_pipe<256, 256, 2, 4, 6>(p, st);
      return launch_f16x3_pipe<256, 256, 2, 4>(p, st);
    }
    case 6: return launch_f16x3_pipe<128, 256, 2, 4>(p, st);
    case 7: return launch_f16x3_pipe<128, 128, 4, 2>(p, st);
    case 8: return launch_f16x3_pipe<64, 256, 2, 4>(p, st);
    case 0: return launch_f16x3<128, 128, 2, 2>(p, st);
    case 1: return launch_f16x3<64, 64, 2, 2>(p, st);
    case 2: return launch_f16x3<128, 32, 4, 1>(p, st);
    default: return launch_f16x3<128, 64, 2, 2>(p, st);
  }
}

}  // namespace mivos

using namespace mivos;

extern "C" int mivos_conv2d_variant_f16x3(int M, int Cout) { return select_variant_f16x3(M, Cout); }

extern "C" int mivos_pack_weights_f16x3(const float *w, void *out, int Cout, int KH, int KW, int Cin, float mult, void *stream) {
  const int Ktot = KH * KW * Cin;
  if (!w || !out || Cout < 1 || Cin < 4 || (Cin & (Cin - 1)) || KH < 1 || KW < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "pack_weights_f16x3: bad arguments");
  const int Kpad = cdiv(Ktot, BKH) * BKH;
  hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3(cdiv((long long)Cout * (Kpad / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     w, (_Float16 *)out, Cout, Ktot, Kpad, mult, Cin, KH * KW);
  return check_launch("pack_weights_f16x3");
}
