// f16x3 implicit-GEMM convolution fed by LDS-DMA from PRE-SPLIT, ZERO-BORDERED operands (precision 2).
//
// What bounds the register-staged kernel (conv_f16x3.hip) - measured, scripts/ubench/lds_vs_mfma.hip:
// while one wave of a SIMD streams MFMAs back to back, its partner wave gets ONE vector-ALU instruction
// issued per MFMA (~40 cycles each), although ds_read_b128, scalar ALU and LDS-DMA issue at full speed.
// Splitting fp32 activations into fp16 (hi, lo) during staging, im2col address arithmetic and padding
// masks are a few hundred VALU instructions per wave and K step: they, not the matrix pipe, LDS or L2
// (33 TB/s measured for LDS-DMA), set the pace (MFMA pipe 35 % busy).  This kernel has NO vector-ALU
// instruction in its main loop:
//   * activations arrive already split ("SH32": per pixel and 32 channels one 128-byte line = 32 fp16 hi |
//     32 fp16 lo, same 4 bytes per element as fp32, written by the producer's epilogue);
//   * their storage has a border of zero pixels >= the conv padding, so im2col needs no masks: the source
//     of every LDS-DMA piece is  constant per-lane byte offset (VGPR)  +  per-step tap/slab offset (SGPR,
//     scalar ALU)  in one buffer_load_dwordx4 ... offen lds;  tile rows past M / columns past Cout carry
//     an out-of-range offset and the buffer unit returns zeros for them;
//   * weights are packed as [K step][Cout][128 B] hi|lo lines, pre-swizzled for the LDS image;
//   * fragment reads use per-lane base addresses computed once + immediate offsets (K loop unrolled x6 so
//     that the 3-deep activation ring and the 2-deep weight ring are addressed by constants).
//
// LDS image per stage: rows x 128 B, chunk c (16 B) of row r at chunk position c ^ ((r >> 1) & 7).  LDS-DMA
// writes lane-linear (8 rows x 128 B per wave instruction), so the swizzle is applied to the SOURCE offset
// (cdna guide rule 21); every ds_read_b128 lane group {0-3,12-15,20-27}, ... of a fragment read then covers all
// sixteen 16-byte slots of the 256-byte bank row: conflict-free.
//
// Ping-pong schedule (cdna guide "256^2 8-phase template" adapted to the 3-product f16x3 step): waves 0-3
// (group G0) and 4-7 (G1) sit one per SIMD each.  A wave alternates LOAD phases (all fragments of one
// 16-deep half step by ds_read_b128 + a few LDS-DMA pieces, ~600 cycles) and MFMA phases (MT*NT*3 back-to-back
// MFMAs on register-resident fragments), separated by workgroup barriers; G1 runs one barrier behind G0,
// so each SIMD always has one wave inside an MFMA phase:
//   G0:  L(t,0) | M(t,0) | L(t,1) | M(t,1) | L(t+1,0) ...
//   G1:         | L(t,0) | M(t,0) | L(t,1) | M(t,1)   ...
//   L(t,0) issues the weight pieces of step t+1 (ring of 2 stages; L2 hits),
//   L(t,1) issues the activation pieces of step t+2 (ring of 3: first touches come from HBM) and ends with
//   vmcnt(#those pieces): everything older - all operands of step t+1 - has landed before the barrier that
//   every reader of step t+1 passes first.  A stage is overwritten no earlier than one step after its last
//   reader retired its ds_reads (lgkmcnt(0) before each barrier).
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

namespace mivos {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int ROWB = 128;   // bytes per tile row and K step: 32 k x (hi, lo) fp16
constexpr unsigned OOB_OFFSET = 0x80000000u;   // + any step offset stays >= num_records (< 2 GB) without wrapping

// FOLD: the K slices of a split-K layer run ONE AFTER THE OTHER inside this workgroup instead of as gridDim.y workgroups + splitk_reduce_kernel: every
// p.kt_split steps the accumulator chain is closed (tot += acc, in ascending slice order; acc restarts from zero) and the epilogue works on the total.
// Same chains, same order of the same fp32 additions as the split launch + reduce pass => the same bits (tests/test_gpu_ops.py); what a launch does when the
// caller says other streams share the chip (mivos_conv_desc.chip_share > 1): no partial-sum round trip, no reduce launch, no workgroups in slots a
// neighbour stream would fill.  The second accumulator set costs MT*NT*16 registers, which the 128x128 / 128x64 / 64x128 tiles absorb (114 VGPRs, still four waves per SIMD);
// the 128x256 tile (242 VGPRs) cannot, and is not instantiated folded - its launches split physically whatever the hint (same bits).
template <int BM, int BN, int WGM, int WGN, int ABL = 0, bool FOLD = false>   // ABL: profiling ablations (1: no DMA, 2: no DMA wait, 3: no fragment reads); 9: experimental merged-half-step schedule
__global__ __launch_bounds__(512) void conv_f16x3_pp_kernel(ConvP p, unsigned x_bytes, unsigned w_bytes) {
  static_assert(WGM * WGN == 8, "8 waves per workgroup");
  static_assert(!FOLD || ABL == 0, "the folded variant exists for the product schedule only");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tile rows are fetched 64 at a time");
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int A_LD = BM / 64, B_LD = BN / 64;        // DMA pieces per wave and step
  constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * ROWB, B_BASE = 3 * A_STAGE;
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist in the device pass only; the host pass just needs the stub
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;
  const int ntaps = p.KH * p.KW;

  // buffer resources: activations from the first border pixel of image 0, weights from their zero line
  const unsigned char *x_lo = reinterpret_cast<const unsigned char *>(p.x) - ((long long)p.pad * p.x_rs + (long long)p.pad * p.x_ps) * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)x_lo, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, w_bytes, 0x00020000);

  const int lrow = lane >> 3, pch = lane & 7;
  // rows this lane fetches: r = (j * 8 + wave) * 8 + lrow  =>  (r >> 1) & 7 = 4 * (wave & 1) + (lrow >> 1)
  const int src_chunk = (pch ^ (4 * (wave & 1) + (lrow >> 1))) * 16;
  unsigned voff_a[A_LD], voff_b[B_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int m = m0 + (j * 8 + wave) * 8 + lrow;
    const int mm = m < p.M ? m : 0;
    const int n = mm / p.HoWo, rem = mm - n * p.HoWo;
    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    const long long off = ((long long)n * p.x_ns + (long long)oh * p.stride * p.x_rs + (long long)ow * p.stride * p.x_ps) * 4 + src_chunk;
    voff_a[j] = m < p.M ? (unsigned)off : OOB_OFFSET;
  }
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int n = n0 + (j * 8 + wave) * 8 + lrow;
    voff_b[j] = n < p.Cout ? (unsigned)(ROWB + n * ROWB + pch * 16) : OOB_OFFSET;
  }

  f32x16 acc[MT][NT];
  f32x16 tot[MT][NT];          // FOLD only (dead code otherwise): sum of the closed K slices
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; if (FOLD) tot[a][b][r] = 0.f; }
  int fold_left = p.kt_split;  // FOLD: K steps until the running slice ends
  auto fold = [&]() {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { tot[a][b][r] += acc[a][b][r]; acc[a][b][r] = 0.f; }
  };

  // ---- LDS-DMA issue: scalar state only --------------------------------------------------------------
  const int lds0 = __builtin_amdgcn_readfirstlane((int)(size_t)smem) + wave * 1024;
  const int pix_step = (int)(p.x_ps * 4);
  const int row_step = (int)(p.x_rs * 4) - p.KW * pix_step;             // from behind the last tap of a row to the next row
  // split-K (small-M layers): gridDim.y slices of kt_split K steps each; raw partial tiles go to p.partial
  const int nk_all = (p.Cin >> 5) * ntaps;
  const int kt0 = p.kt_split ? (int)blockIdx.y * p.kt_split : 0;
  const int nk = FOLD ? nk_all : (p.kt_split && kt0 + p.kt_split < nk_all ? kt0 + p.kt_split : nk_all) - kt0;      // steps of this slice (FOLD: all slices, in turn)
  const int b_step = p.Cout * ROWB;
  int a_tap = kt0 % ntaps, a_kw = a_tap % p.KW;                         // activation stream: tap / slab of its next step
  int a_tap_off = ((a_tap / p.KW) * (int)(p.x_rs * 4)) + a_kw * pix_step, a_slab_off = (kt0 / ntaps) * ROWB;
  int b_off = kt0 * b_step;                                             // weight stream: byte offset of its next step
  auto issue_a = [&](int stage) {
    const int soff = a_tap_off + a_slab_off;
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(size_t)(lds0 + stage * A_STAGE + j * 8192), 16, voff_a[j], soff, 0, 0);
    if (ABL == 7) return;                  // ablation: constant source offsets (no scalar tap / slab bookkeeping)
    a_tap_off += pix_step;
    if (++a_kw == p.KW) { a_kw = 0; a_tap_off += row_step; }
    if (++a_tap == ntaps) { a_tap = 0; a_tap_off = 0; a_slab_off += ROWB; }
  };
  auto issue_b = [&](int stage) {
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(size_t)(lds0 + B_BASE + stage * B_STAGE + j * 8192), 16, voff_b[j], b_off, 0, 0);
    if (ABL != 7) b_off += b_step;
  };

  // ---- fragment addressing: per-lane bases once, stages / tiles by immediates ---------------------------
  // row = tile row of the wave + (lane & 31); logical chunk = part * 4 + kk * 2 + (lane >> 5), XOR ((row >> 1) & 7)
  const int swz = (lane >> 1) & 7;
  const unsigned char *fa[2][2], *fb[2][2];
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = ((part * 4 + kk * 2 + (lane >> 5)) ^ swz) * 16;
      fa[part][kk] = smem + (wm * TM + (lane & 31)) * ROWB + fo;
      fb[part][kk] = smem + B_BASE + (wn * TN + (lane & 31)) * ROWB + fo;
    }

  h8 ah[MT], al[MT], bh[NT], bl[NT];
  h8 ah1[MT], al1[MT], bh1[NT], bl1[NT];          // second fragment set: the merged-half-step schedule only (ABL == 9; dead code otherwise)
  auto load_frags_into = [&](h8 (&AH)[MT], h8 (&AL)[MT], h8 (&BH)[NT], h8 (&BL)[NT], int sa, int sb, int kk) {
    if (ABL == 3) return;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      BH[j] = *reinterpret_cast<const h8 *>(fb[0][kk] + sb * B_STAGE + j * 32 * ROWB);
      BL[j] = *reinterpret_cast<const h8 *>(fb[1][kk] + sb * B_STAGE + j * 32 * ROWB);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      AH[i] = *reinterpret_cast<const h8 *>(fa[0][kk] + sa * A_STAGE + i * 32 * ROWB);
      AL[i] = *reinterpret_cast<const h8 *>(fa[1][kk] + sa * A_STAGE + i * 32 * ROWB);
    }
  };
  auto load_frags = [&](int sa, int sb, int kk) { load_frags_into(ah, al, bh, bl, sa, sb, kk); };
  // per accumulator the three products keep the order of conv_f16x3.hip (lo*hi, hi*lo, hi*hi): bit-identical sums
  auto mfma_phase_on = [&](const h8 (&AH)[MT], const h8 (&AL)[MT], const h8 (&BH)[NT], const h8 (&BL)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], acc[i][j], 0, 0, 0);
  };
  auto mfma_phase = [&]() { mfma_phase_on(ah, al, bh, bl); };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // one K step; U = kt % 6 is a compile-time constant => ring positions kt % 3 / kt % 2 are immediates.
  // TAIL = false: steady state (both prefetches exist), no scalar branches inside the step.
  auto step = [&](int kt, auto u_tag, auto tail_tag) {
    constexpr int U = decltype(u_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    constexpr int sa = U % 3, sb = U % 2;
    if constexpr (ABL == 9) {
      // EXPERIMENTAL (MIVOS_PP_MERGE=1, off by default; design note in docs/NOTEBOOK.md): ONE load phase and ONE MFMA phase per K step - two
      // barriers instead of four.  The DMA burst k = {weights of step k + 1, activations of step k + 2} is issued by BOTH wave groups in the
      // same wall-clock phase - group 0 from LOAD(k), group 1 (one phase behind) from the start of MFMA(k - 1) - and both wait for it at the
      // end of their next phase (vmcnt(A_LD): everything but the newest activation pieces), so that every wave's rows of step k + 1 have
      // landed one barrier before anybody reads them.  The stages a burst overwrites were last read two phases earlier by either group.
      // Same MFMA order per accumulator as the two half steps: bit-identical results.
      const bool g1 = wave >= 4;
      load_frags_into(ah, al, bh, bl, sa, sb, 0);
      load_frags_into(ah1, al1, bh1, bl1, sa, sb, 1);
      if (!g1) {
        if (!TAIL || kt + 1 < nk) issue_b(sb ^ 1);
        if (!TAIL || kt + 2 < nk) issue_a((sa + 2) % 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (TAIL) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_LD) : "memory");
      }
      phase_barrier();
      if (g1) {                                                // burst kt + 1: weights of step kt + 2 over THIS step's weight stage, activations
        if (!TAIL || kt + 2 < nk) issue_b(sb);                 // of step kt + 3 over this step's activation stage (both just read)
        if (!TAIL || kt + 3 < nk) issue_a(sa);
      }
      mfma_phase_on(ah, al, bh, bl);
      mfma_phase_on(ah1, al1, bh1, bl1);
      if (!g1) {
        if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD) : "memory");
      }
      phase_barrier();
      return;
    }
    load_frags(sa, sb, 0);                                     // L(kt, 0)
    if (ABL != 1 && (!TAIL || kt + 1 < nk)) issue_b(sb ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();
    mfma_phase();                                              // M(kt, 0)
    phase_barrier();
    load_frags(sa, sb, 1);                                     // L(kt, 1)
    if (ABL == 1 || ABL == 2) {
      if (ABL == 2 && (!TAIL || kt + 2 < nk)) issue_a((sa + 2) % 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (!TAIL || kt + 2 < nk) {
      issue_a((sa + 2) % 3);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_LD) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    phase_barrier();
    mfma_phase();                                              // M(kt, 1)
    phase_barrier();
    if constexpr (FOLD) {                                      // wave-uniform: every wave counts its own steps
      if (--fold_left == 0) { fold_left = p.kt_split; fold(); }
    }
  };
  using std::integral_constant;
  typedef std::false_type steady;
  typedef std::true_type tail;

  long long t_start = 0;
  if (ABL == 8) t_start = __builtin_readcyclecounter();      // profiling: shader cycles of the main loop / epilogue of workgroup 0
  issue_a(0);
  if (ABL != 1) issue_b(0);
  if (nk > 1) issue_a(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  phase_barrier();                       // step 0 (and the activations of step 1) landed for everybody
  if (ABL == 9 && wave >= 4) {           // merged schedule: group 1 issues burst 0 in the phase in which group 0 runs LOAD(0)
    if (1 < nk) issue_b(1);
    if (2 < nk) issue_a(2);
  }
  if (wave >= 4) phase_barrier();        // G1 runs one phase behind
  int kt = 0;
  for (; kt + (ABL == 9 ? 9 : 8) <= nk; kt += 6) {        // every step of the group still has its prefetches (merged: group 1 looks 3 steps ahead)
    step(kt, integral_constant<int, 0>{}, steady{});
    step(kt + 1, integral_constant<int, 1>{}, steady{});
    step(kt + 2, integral_constant<int, 2>{}, steady{});
    step(kt + 3, integral_constant<int, 3>{}, steady{});
    step(kt + 4, integral_constant<int, 4>{}, steady{});
    step(kt + 5, integral_constant<int, 5>{}, steady{});
  }
  for (; kt < nk; kt += 6) {
    step(kt, integral_constant<int, 0>{}, tail{});
    if (kt + 1 < nk) step(kt + 1, integral_constant<int, 1>{}, tail{});
    if (kt + 2 < nk) step(kt + 2, integral_constant<int, 2>{}, tail{});
    if (kt + 3 < nk) step(kt + 3, integral_constant<int, 3>{}, tail{});
    if (kt + 4 < nk) step(kt + 4, integral_constant<int, 4>{}, tail{});
    if (kt + 5 < nk) step(kt + 5, integral_constant<int, 5>{}, tail{});
  }
  if (wave < 4) phase_barrier();         // G0 waits for G1's last phase; the LDS stages are dead after this
  if (ABL == 8 && blockIdx.x == 0 && tid == 0) {
    reinterpret_cast<long long *>(p.ws)[0] = __builtin_readcyclecounter() - t_start;
    reinterpret_cast<long long *>(p.ws)[1] = nk;
  }
  if (!FOLD && p.kt_split) {             // raw fp32 partial tile of this K slice: [slice][M][Cout], dense
    ConvP q = p;
    q.scale = q.bias = q.res = nullptr;
    q.status = nullptr;                    // partial sums are not outputs: the reduce kernel checks the finished values
    q.relu_out = 0;
    q.split = p.Cout;
    q.y = p.partial + (long long)blockIdx.y * p.M * p.Cout;
    q.y_ps = p.Cout; q.y_rs = (long long)p.Wo * p.Cout; q.y_ns = (long long)p.HoWo * p.Cout; q.y_fmt = 0;
    epilogue_vec<MT, NT>(acc, reinterpret_cast<float *>(smem) + wave * 32 * EPI_PITCH, q, m0 + wm * TM, n0 + wn * TN, lane);
    // (the partial tiles are summed by splitk_reduce_kernel.  Summing them here - the last slice of a tile to arrive, found
    // with a per-tile arrival counter - was built and measured in round 3: the release fence every workgroup needs before it
    // counts itself writes back its XCD's whole L2, and config 3 fell from 192.6 to 132.8 frames/s.)
    return;
  }
  // tiles that fit two workgroups per CU (<= 80 KB LDS) must also stay within 128 VGPRs: prefetch one row tile at a time
  constexpr int EIB = (3 * BM + 2 * BN) * ROWB <= 80 * 1024 ? 1 : (MT > 2 ? 1 : MT);
  if constexpr (FOLD) fold();            // the last (possibly shorter) slice; a no-op sum of zeros when the step count divides evenly
  const f32x16 (&fin)[MT][NT] = FOLD ? tot : acc;
  // (the folded kernel finishes with the reduce pass's arithmetic - multiply, round, add - every other launch with the fused one it always had: conv_common.h)
  if (p.y_fmt && p.split == p.Cout) epilogue_sh32<MT, NT, EIB, !FOLD>(fin, reinterpret_cast<float *>(smem) + wave * 32 * EPI_PITCH, p, m0 + wm * TM, n0 + wn * TN, lane);
  else if (p.vec_epi) epilogue_vec<MT, NT, EIB, !FOLD>(fin, reinterpret_cast<float *>(smem) + wave * 32 * EPI_PITCH, p, m0 + wm * TM, n0 + wn * TN, lane);
  else epilogue_scalar<MT, NT, !FOLD>(fin, p, m0 + wm * TM, n0 + wn * TN, lane);
  if (ABL == 8 && blockIdx.x == 0 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    reinterpret_cast<long long *>(p.ws)[2] = __builtin_readcyclecounter() - t_start;
  }
#endif
}

// fp32 NHWC (strided) -> SH32 (strided, e.g. the interior of a zero-bordered buffer), optionally through ReLU
__global__ void pack_activation_sh32_kernel(const float *__restrict__ x, long long x_ns, long long x_rs, long long x_ps,
                                            unsigned char *__restrict__ y, long long y_ns, long long y_rs, long long y_ps, int N, int H, int W,
                                            int C, int relu) {
  const int c8n = C >> 3;
  const long long total = (long long)N * H * W * c8n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    long long pix = e / c8n;
    const int c = (int)(e - pix * c8n) * 8;
    const int n = (int)(pix / ((long long)H * W));
    pix -= (long long)n * H * W;
    const int h = (int)(pix / W), w = (int)(pix - (long long)h * W);
    const f32x4 *src = reinterpret_cast<const f32x4 *>(x + n * x_ns + h * x_rs + w * x_ps + c);
    const f32x4 v0 = src[0], v1 = src[1];
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    h8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = relu ? fmaxf(v[i], 0.f) : v[i];
      hi[i] = (_Float16)t;
      lo[i] = (_Float16)__builtin_fmaf((float)hi[i], -1.f, t);
    }
    unsigned char *dst = y + (n * y_ns + h * y_rs + w * y_ps) * 4 + (c >> 5) * ROWB + (c & 31) * 2;
    *reinterpret_cast<h8 *>(dst) = hi;
    *reinterpret_cast<h8 *>(dst + 64) = lo;
  }
}

// SH32 (strided) -> fp32 NHWC (strided): x = hi + lo
__global__ void unpack_activation_sh32_kernel(const unsigned char *__restrict__ x, long long x_ns, long long x_rs, long long x_ps,
                                              float *__restrict__ y, long long y_ns, long long y_rs, long long y_ps, int N, int H, int W, int C) {
  const int c8n = C >> 3;
  const long long total = (long long)N * H * W * c8n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    long long pix = e / c8n;
    const int c = (int)(e - pix * c8n) * 8;
    const int n = (int)(pix / ((long long)H * W));
    pix -= (long long)n * H * W;
    const int h = (int)(pix / W), w = (int)(pix - (long long)h * W);
    const unsigned char *src = x + (n * x_ns + h * x_rs + w * x_ps) * 4 + (c >> 5) * ROWB + (c & 31) * 2;
    const h8 hi = *reinterpret_cast<const h8 *>(src), lo = *reinterpret_cast<const h8 *>(src + 64);
    f32x4 v0, v1;
    v0.x = (float)hi[0] + (float)lo[0]; v0.y = (float)hi[1] + (float)lo[1]; v0.z = (float)hi[2] + (float)lo[2]; v0.w = (float)hi[3] + (float)lo[3];
    v1.x = (float)hi[4] + (float)lo[4]; v1.y = (float)hi[5] + (float)lo[5]; v1.z = (float)hi[6] + (float)lo[6]; v1.w = (float)hi[7] + (float)lo[7];
    f32x4 *dst = reinterpret_cast<f32x4 *>(y + n * y_ns + h * y_rs + w * y_ps + c);
    dst[0] = v0;
    dst[1] = v1;
  }
}

__global__ void pack_weights_dma_kernel(const float *__restrict__ w, unsigned char *__restrict__ out, int Cout, int cin,
                                        int ntaps, float mult) {
  const int nsteps = (cin >> 5) * ntaps;
  const long long total = (long long)nsteps * Cout * 8;
  if (blockIdx.x == 0 && threadIdx.x < 32) reinterpret_cast<float *>(out)[threadIdx.x] = 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(e & 7);                          // physical chunk
    const long long row = e >> 3;                        // s * Cout + n
    const int s = (int)(row / Cout), n = (int)(row - (long long)s * Cout);
    const int lc = q ^ ((n >> 1) & 7), part = lc >> 2, k0 = (lc & 3) * 8;
    const int slab = s / ntaps, tap = s - slab * ntaps;
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = w[((long long)n * ntaps + tap) * cin + slab * 32 + k0 + i] * mult;
      const _Float16 hi = (_Float16)v;
      o[i] = part ? (_Float16)(v - (float)hi) : hi;
    }
    *reinterpret_cast<h8 *>(out + ROWB + row * ROWB + q * 16) = o;
  }
}

static std::atomic<int> g_fold_mode{getenv("MIVOS_PP_FOLD") ? atoi(getenv("MIVOS_PP_FOLD")) : 1};

static FILE *shape_log_file() {
  static FILE *const f = getenv("MIVOS_CONV_LOG") ? fopen(getenv("MIVOS_CONV_LOG"), "a") : nullptr;
  return f;
}

template <int BM, int BN, int WGM, int WGN, int ABL = 0>
static int launch_pp(ConvP &p, hipStream_t st) {
  const int tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  const size_t lds = (3ull * BM + 2ull * BN) * ROWB;
  auto kern = conv_f16x3_pp_kernel<BM, BN, WGM, WGN, ABL>;
  static std::atomic<uint64_t> attr_mask{0};  // per instantiation, one bit per device
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_mask, "conv_f16x3_pp")) return rc;
  const long long x_bytes = ((long long)(p.N - 1) * p.x_ns + (long long)(p.H + 2 * p.pad - 1) * p.x_rs + (long long)(p.W + 2 * p.pad) * p.x_ps) * 4;
  const long long w_bytes = ROWB + (long long)(p.Cin >> 5) * p.KH * p.KW * p.Cout * ROWB;
  if (x_bytes >= 0x7ff00000ll || w_bytes >= 0x7ff00000ll) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d (SH32 input): tensor larger than 2 GB");
  // Split-K: ONLY for grids that leave three quarters of the workgroup slots empty and have enough K steps to amortise the partial-sum round trip.
  // The rule looks at the layer shape alone - never at mivos_conv_desc.chip_share - because the number of K slices is the fp32 summation order: with a
  // share-dependent rule (round 5) the masks of a clip changed with the number of clips in flight.  Measured per shape IN SITU (rocprofv3 kernel trace of a
  // config-3 session joined with the launch log, profiles/r06b_insitu_*): at M = 8100 the 3x3 256->256 layers (128 workgroups, 72 K steps) take 42 us split by 4
  // (reduce included) against 61 us unsplit; the 1x1 1024->256 layers (32 steps) 34.8 vs 34.6, the decoder's 512->512 layers (256 workgroups) 222 vs 220,
  // KeyValue (320) 275 vs 283 - those no longer split (less partial-sum traffic, 9 fewer reduce launches per frame, and nothing a second clip's launches
  // would have to queue behind).
  static const int split_on = getenv("MIVOS_PP_SPLIT_THR") ? atoi(getenv("MIVOS_PP_SPLIT_THR")) : 1;   // tuning only: 0 = never split
  static const int min_nk = getenv("MIVOS_PP_SPLIT_MIN_NK") ? atoi(getenv("MIVOS_PP_SPLIT_MIN_NK")) : 64;   // tuning only
  const int nk = (p.Cin >> 5) * p.KH * p.KW, wgs = tiles_m * p.tiles_n, cap = lds <= 80 * 1024 ? 512 : 256;
  int slices = 1;
  if (split_on && p.vec_epi && p.ws && wgs * 4 <= cap && nk >= (wgs >= 64 ? min_nk : 16)) {
    slices = cap / wgs;
    if (slices > 8) slices = 8;
    if (slices > nk / 8) slices = nk / 8;
    if ((long long)slices * p.M * p.Cout * 4 > p.ws_bytes) slices = 1;
  }
  static const int force_slices = getenv("MIVOS_PP_SPLIT") ? atoi(getenv("MIVOS_PP_SPLIT")) : 0;   // tuning only
  if (force_slices && p.vec_epi && p.ws && (long long)force_slices * p.M * p.Cout * 4 <= p.ws_bytes && nk >= 2 * force_slices) slices = force_slices;
  p.kt_split = 0;
  if (slices > 1) { p.kt_split = cdiv(nk, slices); slices = cdiv(nk, p.kt_split); p.partial = (float *)p.ws; }
  // The caller keeps other launch streams busy (chip_share > 1) and the split grid would be large (>= 64 tiles x slices): the same K slices, folded inside
  // one workgroup each (see FOLD above) - identical bits, no partial sums, no reduce launch, the slots stay free for the neighbour streams.  Small grids
  // (one-object clips: 26 tiles) split physically whatever the hint: measured same box, three one-object clips in flight 568 frames/s split vs 540 folded; two
  // five-object clips 202.7 vs 200.7 (profiles/r06c_fold_ab.txt).  Tiles up to 128x128 only (the accumulators of the total must fit the register file).
  // MIVOS_PP_FOLD=0 never folds, =2 folds every split layer whatever the hint or the grid (A/B, tests: mivos_conv2d_set_fold_mode).
  const int fold_mode = g_fold_mode.load(std::memory_order_relaxed);
  if constexpr (ABL == 0 && BM * BN <= 128 * 128) {
    if (slices > 1 && (fold_mode == 2 || (fold_mode == 1 && p.share > 1 && wgs >= 64))) {
      auto kfold = conv_f16x3_pp_kernel<BM, BN, WGM, WGN, 0, true>;
      static std::atomic<uint64_t> fold_attr_mask{0};
      if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kfold), lds, fold_attr_mask, "conv_f16x3_pp (folded)")) return rc;
      if (shape_log_file()) { fprintf(shape_log_file(), "pp %d %d %d %d %d %d %d %d %d %d %d %d\n", BM, BN, p.M, p.Cin, p.Cout, p.KH, p.stride, p.res ? 1 : 0, -slices, tiles_m * p.tiles_n, p.share, p.y_fmt); fflush(shape_log_file()); }
      hipLaunchKernelGGL(kfold, dim3(tiles_m * p.tiles_n, 1), dim3(512), lds, st, p, (unsigned)x_bytes, (unsigned)w_bytes);
      return check_launch("conv_f16x3_pp (folded)");
    }
  }
  // profiling only (scripts/insitu_shape_table.py): one line per launch, in host order, to join with a rocprofv3 kernel trace by dispatch order
  FILE *const shape_log = shape_log_file();
  if (shape_log) {
    fprintf(shape_log, "pp %d %d %d %d %d %d %d %d %d %d %d %d\n", BM, BN, p.M, p.Cin, p.Cout, p.KH, p.stride, p.res ? 1 : 0, slices, tiles_m * p.tiles_n, p.share, p.y_fmt);
    fflush(shape_log);
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n, slices), dim3(512), lds, st, p, (unsigned)x_bytes, (unsigned)w_bytes);
  if (slices > 1) return launch_splitk_reduce(p, slices, st);
  return check_launch("conv_f16x3_pp");
}

// Tile selection for precision 2 (nk = K steps of 32 channels x 1 tap).  20: 128x128 (80 KB LDS: two workgroups per CU overlap each other's barrier
// bubbles; the robust default), 21: 128x256 (one workgroup per CU; wins when its tiles fill whole rounds of the
// 256 CUs), 22: 128x64 (Cout <= 64), 23: 256x256 (profiling only: equal to 20 on its best shapes).
int select_variant_pp(int M, int Cout, int nk) {
  if (Cout <= 64) return 22;
  // tuning only (A/B): under-filled 128x128 grids (30x54 layers: 64 row tiles) as 128x64 tiles - twice the workgroups, less split-K
  static const int small_wgs = getenv("MIVOS_PP_SMALL_WGS") ? atoi(getenv("MIVOS_PP_SMALL_WGS")) : 0;
  if (small_wgs && (long long)cdiv(M, 128) * cdiv(Cout, 128) <= small_wgs) return 22;
  if (Cout % 256 == 0 && nk >= 36) {      // short-K (1x1) layers are bound by loads/stores: two workgroups per CU overlap them
    const long long t = (long long)cdiv(M, 128) * (Cout / 256);
    const long long rounds = (t + 255) / 256;
    if (t >= 200 && (double)t / (double)(rounds * 256) >= 0.85) return 21;
    // Long-K layers whose 128x256 tiles fill half (a quarter) of the chip: split-K by 2 (4) makes it one full round of 256
    // workgroups with >= 18 K steps each.  The 128x128 tiles these layers used to get run at a third of the matrix-pipe
    // utilisation of the 128x256 ones (PMC MfmaUtil 24 % vs 65 %: a wave of the 2x4 grid computes 64x32 outputs per fragment
    // set instead of 64x64, so the LDS fragment traffic per MFMA doubles).
    static const int wide_nk = getenv("MIVOS_PP_WIDE_NK") ? atoi(getenv("MIVOS_PP_WIDE_NK")) : 0;   // tuning: 0 = off, else the minimal K steps
    if (wide_nk && nk >= wide_nk && (t == 128 || t == 64)) return 21;
  }
  return 20;
}

int launch_conv_f16x3_dma(ConvP &p, hipStream_t st) {
  if (p.Cin & 31) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d (SH32 input): Cin %% 32 != 0");
  if (p.relu_in) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d (SH32 input): relu_in must be applied by the producer");
  if ((p.x_ps & 31) || (p.x_ns & 31) || (p.x_rs & 31)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d (SH32 input): strides must be multiples of 32");
  if (p.x_border < p.pad) return fail(MIVOS_ERR_INVALID_ARGUMENT, "conv2d (SH32 input): needs a zero border >= pad around every image");
  static const int force = getenv("MIVOS_PP_TILE") ? atoi(getenv("MIVOS_PP_TILE")) : 0;   // tuning only
  switch (force ? force : select_variant_pp(p.M, p.Cout, (p.Cin >> 5) * p.KH * p.KW)) {
    case 23: {
      static const int abl = getenv("MIVOS_ABL") ? atoi(getenv("MIVOS_ABL")) : 0;   // profiling only
      if (abl == 1) return launch_pp<256, 256, 2, 4, 1>(p, st);
      if (abl == 2) return launch_pp<256, 256, 2, 4, 2>(p, st);
      if (abl == 3) return launch_pp<256, 256, 2, 4, 3>(p, st);
      if (abl == 7) return launch_pp<256, 256, 2, 4, 7>(p, st);
      if (abl == 8) return launch_pp<256, 256, 2, 4, 8>(p, st);
      return launch_pp<256, 256, 2, 4>(p, st);
    }
    case 21: {
      static const int abl = getenv("MIVOS_ABL") ? atoi(getenv("MIVOS_ABL")) : 0;   // profiling only
      if (abl == 8) return launch_pp<128, 256, 2, 4, 8>(p, st);
      return launch_pp<128, 256, 2, 4>(p, st);
    }
    case 22: return launch_pp<128, 64, 4, 2>(p, st);
    case 24: return launch_pp<64, 128, 2, 4>(p, st);       // 64-row tiles: M = 8100 layers as 127 row tiles (56 KB LDS: two workgroups per CU)
    default: {
      static const int merge = getenv("MIVOS_PP_MERGE") ? atoi(getenv("MIVOS_PP_MERGE")) : 0;   // experimental schedule (round-5 experiment), off
      if (merge) return launch_pp<128, 128, 2, 4, 9>(p, st);
      return launch_pp<128, 128, 2, 4>(p, st);
    }
  }
}

}  // namespace mivos

using namespace mivos;

extern "C" int mivos_conv2d_variant_pp(int M, int Cout, int ksteps) { return select_variant_pp(M, Cout, ksteps); }

extern "C" int mivos_conv2d_set_fold_mode(int mode) {
  const int prev = g_fold_mode.load(std::memory_order_relaxed);
  if (mode >= 0 && mode <= 2) g_fold_mode.store(mode, std::memory_order_relaxed);
  return prev;
}

extern "C" int mivos_pack_activation_sh32(const float *x, int64_t x_nstride, int64_t x_rstride, int64_t x_pstride, void *y, int64_t y_nstride,
                                           int64_t y_rstride, int64_t y_pstride, int N, int H, int W, int C, int relu, void *stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 32 || (C & 31) || ((x_pstride | x_rstride | x_nstride) & 3) || ((y_pstride | y_rstride | y_nstride) & 31) ||
      ((uintptr_t)x & 15) || ((uintptr_t)y & 127))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "pack_activation_sh32: bad arguments");
  const long long total = (long long)N * H * W * (C >> 3);
  const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  hipLaunchKernelGGL(pack_activation_sh32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)x_nstride, (long long)x_rstride,
                     (long long)x_pstride, (unsigned char *)y, (long long)y_nstride, (long long)y_rstride, (long long)y_pstride, N, H, W, C, relu);
  return check_launch("pack_activation_sh32");
}

extern "C" int mivos_unpack_activation_sh32(const void *x, int64_t x_nstride, int64_t x_rstride, int64_t x_pstride, float *y, int64_t y_nstride,
                                             int64_t y_rstride, int64_t y_pstride, int N, int H, int W, int C, void *stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 32 || (C & 31) || ((y_pstride | y_rstride | y_nstride) & 3) || ((x_pstride | x_rstride | x_nstride) & 31) ||
      ((uintptr_t)y & 15) || ((uintptr_t)x & 127))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "unpack_activation_sh32: bad arguments");
  const long long total = (long long)N * H * W * (C >> 3);
  const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  hipLaunchKernelGGL(unpack_activation_sh32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)x, (long long)x_nstride,
                     (long long)x_rstride, (long long)x_pstride, y, (long long)y_nstride, (long long)y_rstride, (long long)y_pstride, N, H, W, C);
  return check_launch("unpack_activation_sh32");
}

extern "C" int64_t mivos_pack_weights_f16x3_dma_bytes(int Cout, int KH, int KW, int Cin) {
  return 128 + (int64_t)(Cin >> 5) * KH * KW * Cout * 128;
}

extern "C" int mivos_pack_weights_f16x3_dma(const float *w, void *out, int Cout, int KH, int KW, int Cin, float mult, void *stream) {
  if (!w || !out || Cout < 1 || Cin < 32 || (Cin & 31) || KH < 1 || KW < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "pack_weights_f16x3_dma: bad arguments");
  const long long total = (long long)(Cin >> 5) * KH * KW * Cout * 8;
  const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  hipLaunchKernelGGL(pack_weights_dma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned char *)out, Cout, Cin, KH * KW, mult);
  return check_launch("pack_weights_f16x3_dma");
}
