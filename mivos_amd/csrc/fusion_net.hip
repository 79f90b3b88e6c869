// FusionNet.forward (model/fusion_net.py:32-50) for gfx950: three kinds of launches instead of the reference's six
// convolutions + two adds, behind ONE C-ABI call (mivos_fusion_net_forward):
//
//   conv1 (9 -> 32, ReLU)                      the library's direct 3x3 kernel (conv_f16x3.hip: conv3x3_n32_direct_kernel<16>)
//   2 x fusion_resblock_kernel                 x <- relu(x + conv_b(relu(conv_a(x))))   fusion_net.py:42-43 / :45-46, ONE launch each
//   fusion_head_kernel                         final_conv 32 -> 1                       fusion_net.py:49
//
// Why: at 480p x 5 objects a 32-channel fp32 plane set is 265 MB, and the layer-by-layer path moved 3.6 GB per fused frame
// (read x, read the residual, write y for each of four convolutions at ~3.5 TB/s = HBM bound, 190-240 us each) plus the head as a
// 1x1 projection to nine tap planes (132 MB written, read again by a 9-point sum).  Fused, a residual block reads x once
// (+ halo) and writes once, its intermediate r = relu(conv_a(x)) never leaves LDS, and the head reads x once and writes
// one plane.
//
// fusion_resblock_kernel - persistent, 8 waves, one workgroup per CU (157 KB of LDS):
//   * tile = 8 x 30 output pixels of one image; the intermediate is needed on 10 x 32 pixels (one 32-pixel MFMA column
//     group per row - the reason for the 30), the input on 12 x 34;
//   * the 12 x 34 x 32 input patch is split to fp16 hi / lo ONCE into LDS (80-byte pixel pitch: conflict-free fragment
//     reads), the packed hi / lo weights of BOTH convolutions stay LDS-resident for the whole launch (92 KB);
//   * phase A: waves take the 10 intermediate rows (waves 0, 1 two rows each, sharing the weight fragments): per tap and
//     16-channel block three v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi, fp32 accumulate - the arithmetic of every
//     convolution of the engine, same product order as conv3x3_n32_direct_kernel); weights are the MFMA A operand, so a lane
//     ends up with 4 consecutive channels of one pixel per register quad;
//   * r = relu(acc * scale + bias), forced to ZERO outside the image (conv_b's zero padding), is split to hi / lo in
//     registers and - after a barrier, every wave being done with the input patch - written OVER the patch;
//   * phase B: 8 output rows, one per wave, the same MFMA sequence on r; epilogue: + bias + x (16-byte loads from
//     global memory: the lines were just read for the patch and hit in L2), ReLU, 16-byte stores;
//   * the next tile's patch is in flight in registers during the whole tile (requested a tile ahead, converted and
//     written when phase B is done).
// fusion_head_kernel - 3x3 / pad 1 / Cout = 1 on 32 channels in exact fp32 FMA (0.6 GFMA per launch: nothing for the
//   vector ALU): a 10 x 34 patch in LDS, one output pixel per thread, weights through scalar loads.  HBM bound: reads x
//   once, writes one plane.
#include <string.h>

#include "conv_common.h"

namespace mivos {

typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
typedef _Float16 fh4 __attribute__((ext_vector_type(4)));

constexpr int FB_TR = 8, FB_TC = 30;                 // output tile
constexpr int FB_IR = FB_TR + 2, FB_IC = 32;         // intermediate region (rows, columns = one MFMA column group)
constexpr int FB_PR = FB_TR + 4, FB_PC = FB_TC + 4;  // input patch
constexpr int FB_PPX = 40;                           // halves per pixel of an LDS image (32 channels + 8 pad = 80 bytes)
constexpr int FB_PW = 40;                            // halves per weight row
constexpr int FB_IMG_P = FB_PR * FB_PC * FB_PPX;     // halves of one patch image (hi or lo)
constexpr int FB_RPX = FB_IR * FB_IC + 2;            // intermediate pixels + 2 of slack (lanes 30, 31 of an output row read past it)
constexpr int FB_IMG_R = FB_RPX * FB_PPX;
constexpr int FB_WIMG = 9 * 32 * FB_PW;              // halves of one weight image (hi or lo of one convolution)
constexpr int FB_LDS_HALVES = 2 * FB_IMG_P + 4 * FB_WIMG + 2 * 4 * 32;   // + scale / bias of both convolutions (4 x 32 floats)
constexpr int FB_NLD = (FB_PR * FB_PC * 8 + 511) / 512;   // float4 patch loads per thread
static_assert(2 * FB_IMG_R <= 2 * FB_IMG_P, "the intermediate overlays the input patch");
static_assert(FB_LDS_HALVES * 2 <= 160 * 1024, "LDS budget of one CU");

struct FuseBlockP {
  const float *x;                 // [B][H][W][32] fp32, dense
  float *y;                       // same shape
  const float *wa, *wb;           // mivos_pack_weights_f16x3 rows: [32][kpad4] float4 = hi0..3 | lo0..3 (fp16), K = tap * 32 + c
  const float *sa, *sb;           // per-channel scale (2^-s of the weight pre-scaling)
  const float *ba, *bb;           // per-channel bias or NULL
  int B, H, W, kpad4, tiles_x, tiles_y, n_tiles, contig;
};

// 3x3 taps x 32 channels on NR pixel rows at once (rows share the weight fragments).  X*: hi / lo images of the source
// pixels with `pitch_px` pixels per row; row[r]: first source row of output row r; lane (i, h): pixel column i, k half h.
template <int NR>
__device__ __forceinline__ void fb_conv_rows(const _Float16 *Xh, const _Float16 *Xl, int pitch_px, const _Float16 *Wh, const _Float16 *Wl,
                                             const int (&row)[NR], int i, int h, f32x16 (&acc)[NR]) {
  // 18 steps (tap, 16-channel block), software pipelined by hand: the fragments of step s + 1 are requested BEFORE the MFMAs of
  // step s are issued, so the LDS latency runs under the 96 / 192 matrix-pipe cycles of a step (left to itself the compiler
  // requests a step's fragments right before its MFMAs and waits: LDS time and MFMA time added up, 344 us per launch at
  // 480p x 5 objects against 0.14 ms of MFMA issue).  Same products in the same order: results unchanged.
  fh8 wh[2], wl[2], xh[2][NR], xl[2][NR];
  auto load = [&](int s, int set) {
    const int tap = s >> 1, kb = s & 1, kh = tap / 3, kw = tap - 3 * kh;
    const int boff = (tap * 32 + i) * FB_PW + 8 * h + 16 * kb;
    wh[set] = *reinterpret_cast<const fh8 *>(Wh + boff);
    wl[set] = *reinterpret_cast<const fh8 *>(Wl + boff);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int aoff = ((row[r] + kh) * pitch_px + i + kw) * FB_PPX + 8 * h + 16 * kb;
      xh[set][r] = *reinterpret_cast<const fh8 *>(Xh + aoff);
      xl[set][r] = *reinterpret_cast<const fh8 *>(Xl + aoff);
    }
  };
  load(0, 0);
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int cur = s & 1;
    if (s + 1 < 18) load(s + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      // weights are the MFMA "A" operand: D[channel][pixel]; small terms first (the order of conv3x3_n32_direct_kernel)
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur], xl[cur][r], acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cur], xh[cur][r], acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur], xh[cur][r], acc[r], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(512) void fusion_resblock_kernel(const FuseBlockP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
  _Float16 *Ph = reinterpret_cast<_Float16 *>(fb_smem), *Pl = Ph + FB_IMG_P;   // input patch, hi | lo images
  _Float16 *Rh = Ph, *Rl = Ph + FB_IMG_R;                                       // the intermediate overlays it
  _Float16 *Wah = Ph + 2 * FB_IMG_P, *Wal = Wah + FB_WIMG, *Wbh = Wal + FB_WIMG, *Wbl = Wbh + FB_WIMG;
  float *SB = reinterpret_cast<float *>(Wbl + FB_WIMG);                         // scale_a | bias_a | scale_b | bias_b, 32 floats each

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;

  // ---- packed weights of both convolutions, once per (persistent) workgroup: [32][kpad4][hi0..3 | lo0..3] -> [tap][n][c]
  for (int e = tid; e < 2 * 32 * 72; e += 512) {
    const int conv = e / (32 * 72), ee = e - conv * (32 * 72);
    const int n = ee / 72, q = ee - n * 72;
    const f32x4 v = reinterpret_cast<const f32x4 *>(conv ? p.wb : p.wa)[(long long)n * p.kpad4 + q];
    const int k = 4 * q, tap = k >> 5, c = k & 31;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 hh, ll;
    hh.x = v.x; hh.y = v.y; ll.x = v.z; ll.y = v.w;
    *reinterpret_cast<f32x2 *>((conv ? Wbh : Wah) + (tap * 32 + n) * FB_PW + c) = hh;
    *reinterpret_cast<f32x2 *>((conv ? Wbl : Wal) + (tap * 32 + n) * FB_PW + c) = ll;
  }

  if (tid < 128) {
    const int which = tid >> 5, c = tid & 31;
    const float *src = which == 0 ? p.sa : (which == 1 ? p.ba : (which == 2 ? p.sb : p.bb));
    SB[tid] = src ? src[c] : 0.f;                       // (a missing bias is a zero bias)
  }

  f32x4 pre[FB_NLD];
  auto tile_coords = [&](int tile, int &img, int &y0, int &x0) {
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    img = tile / p.tiles_y;
    y0 = ty * FB_TR;
    x0 = tx * FB_TC;
  };
  // all patch loads of a tile are issued back to back into registers; converted + written a tile later
  auto load_patch = [&](int tile) {
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    const float *xb = p.x + (long long)img * p.H * p.W * 32;
#pragma unroll
    for (int l = 0; l < FB_NLD; ++l) {
      const int e = tid + 512 * l;
      const int px = e >> 3, c4 = e & 7;
      const int pr = px / FB_PC, pc = px - pr * FB_PC;
      const int iy = y0 - 2 + pr, ix = x0 - 2 + pc;
      const bool ok = e < FB_PR * FB_PC * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long long off = ok ? ((long long)iy * p.W + ix) * 32 + 4 * c4 : 0ll;
      f32x4 v = *reinterpret_cast<const f32x4 *>(xb + off);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      pre[l] = v;
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int l = 0; l < FB_NLD; ++l) {
      const int e = tid + 512 * l;
      if (e < FB_PR * FB_PC * 8) {
        const int px = e >> 3, c4 = e & 7;
        const f32x4 v = pre[l];
        fh4 hi, lo;
        hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
        lo.x = (_Float16)(v.x - (float)hi.x); lo.y = (_Float16)(v.y - (float)hi.y);
        lo.z = (_Float16)(v.z - (float)hi.z); lo.w = (_Float16)(v.w - (float)hi.w);
        *reinterpret_cast<fh4 *>(Ph + px * FB_PPX + 4 * c4) = hi;
        *reinterpret_cast<fh4 *>(Pl + px * FB_PPX + 4 * c4) = lo;
      }
    }
  };

  // r = relu(acc * scale + bias) of intermediate pixel (ir, i), zero outside the image, split: registers 4g..4g+3 of the
  // accumulator = channels 8g + 4h .. +3  ->  four hi and four lo quads
  // per-channel scale / bias of this lane's 16 channels (8g + 4h .. +3) from their LDS copy
  auto load_sb = [&](int conv, f32x4 (&s4)[4], f32x4 (&b4)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s4[g] = *reinterpret_cast<const f32x4 *>(SB + 64 * conv + 8 * g + 4 * h);
      b4[g] = *reinterpret_cast<const f32x4 *>(SB + 64 * conv + 32 + 8 * g + 4 * h);
    }
  };
  auto finish_a = [&](const f32x16 &acc, const f32x4 (&sc)[4], const f32x4 (&bi)[4], int ir, int y0, int x0, fh4 (&rh)[4], fh4 (&rl)[4]) {
    const int y = y0 - 1 + ir, x = x0 - 1 + i;
    const bool inside = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 s4 = sc[g], b4 = bi[g];
      float v[4];
      v[0] = __builtin_fmaf(acc[4 * g], s4.x, b4.x); v[1] = __builtin_fmaf(acc[4 * g + 1], s4.y, b4.y);
      v[2] = __builtin_fmaf(acc[4 * g + 2], s4.z, b4.z); v[3] = __builtin_fmaf(acc[4 * g + 3], s4.w, b4.w);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = inside ? fmaxf(v[j], 0.f) : 0.f;
        const _Float16 hi = (_Float16)t;
        rh[g][j] = hi;
        rl[g][j] = (_Float16)(t - (float)hi);
      }
    }
  };
  auto write_r = [&](int ir, const fh4 (&rh)[4], const fh4 (&rl)[4]) {
    const int px = ir * FB_IC + i;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<fh4 *>(Rh + px * FB_PPX + 8 * g + 4 * h) = rh[g];
      *reinterpret_cast<fh4 *>(Rl + px * FB_PPX + 8 * g + 4 * h) = rl[g];
    }
  };

  // Persistent walk, XCD-contiguous: workgroup b takes ONE contiguous run of tiles and the runs of the workgroups sharing an XCD (hardware
  // block b runs on XCD b % 8) are contiguous, so the halo rows / columns neighbouring tiles share are re-read from that XCD's L2 (or by
  // the same CU) instead of through the fabric by another XCD (PMC FETCH_SIZE 2.5 - 5 x the input before, profiles/r04d_config3_pmc_traffic.json)
  const int per_wg = (p.n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tstep = p.contig ? 1 : (int)gridDim.x;               // (contig = 0: the round-robin walk of rounds 1-3, A/B only)
  int tile = p.contig ? xcd_remap((int)blockIdx.x, (int)gridDim.x) * per_wg : (int)blockIdx.x;
  const int tile_end = !p.contig ? p.n_tiles : (tile + per_wg < p.n_tiles ? tile + per_wg : p.n_tiles);
  if (tile < tile_end) { load_patch(tile); write_patch(); }
  if (tile + tstep < tile_end) load_patch(tile + tstep);
  // the two slack pixels behind the intermediate (read by lanes 30 / 31 of an output row, whose results are dropped) hold
  // whatever patch data lies there: finite fp16 numbers, and an MFMA column only ever mixes data of its own pixel
  __syncthreads();
  for (; tile < tile_end; tile += tstep) {
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);

    // ---- phase A: intermediate rows `wave` (all waves) and 8 + wave (waves 0, 1)
    fh4 rh0[4], rl0[4], rh1[4], rl1[4];
    f32x4 sc[4], bi[4];
    if (wave < 2) {
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      const int rows[2] = {wave, 8 + wave};
      fb_conv_rows<2>(Ph, Pl, FB_PC, Wah, Wal, rows, i, h, acc);
      load_sb(0, sc, bi);
      finish_a(acc[0], sc, bi, wave, y0, x0, rh0, rl0);
      finish_a(acc[1], sc, bi, 8 + wave, y0, x0, rh1, rl1);
    } else {
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
      const int rows[1] = {wave};
      fb_conv_rows<1>(Ph, Pl, FB_PC, Wah, Wal, rows, i, h, acc);
      load_sb(0, sc, bi);
      finish_a(acc[0], sc, bi, wave, y0, x0, rh0, rl0);
    }
    __syncthreads();                                    // every wave is done reading the input patch
    write_r(wave, rh0, rl0);
    if (wave < 2) write_r(8 + wave, rh1, rl1);
    __syncthreads();

    // ---- phase B: output row `wave`
    {
      // the residual x[y][x][:] of this lane's output pixel is requested BEFORE the MFMAs (the lines were read for the patch a
      // tile ago: L2 hits, but still ~1 us that the epilogue would otherwise wait for with nothing to overlap it)
      const int y = y0 + wave, x = x0 + i;
      const bool valid = i < FB_TC && y < p.H && x < p.W;
      const long long pix = valid ? ((long long)img * p.H + y) * p.W + x : 0ll;
      const float *xp = p.x + pix * 32;
      f32x4 res[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) res[g] = *reinterpret_cast<const f32x4 *>(xp + 8 * g + 4 * h);
      __builtin_amdgcn_sched_barrier(0);                // (the scheduler would sink the loads behind the MFMAs, to their use)
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
      const int rows[1] = {wave};
      fb_conv_rows<1>(Rh, Rl, FB_IC, Wbh, Wbl, rows, i, h, acc);
      load_sb(1, sc, bi);
      if (valid) {
        float *yp = p.y + pix * 32;
        f32x4 out[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 s4 = sc[g], b4 = bi[g];
          f32x4 v;
          v.x = fmaxf(__builtin_fmaf(acc[0][4 * g], s4.x, b4.x) + res[g].x, 0.f);
          v.y = fmaxf(__builtin_fmaf(acc[0][4 * g + 1], s4.y, b4.y) + res[g].y, 0.f);
          v.z = fmaxf(__builtin_fmaf(acc[0][4 * g + 2], s4.z, b4.z) + res[g].z, 0.f);
          v.w = fmaxf(__builtin_fmaf(acc[0][4 * g + 3], s4.w, b4.w) + res[g].w, 0.f);
          out[g] = v;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(yp + 8 * g + 4 * h) = out[g];
      }
    }
    __syncthreads();                                    // every wave is done reading the intermediate
    if (tile + tstep < tile_end) write_patch();
    if (tile + 2 * tstep < tile_end) load_patch(tile + 2 * tstep);
    __syncthreads();
  }
}

// ---- conv1 (9 -> 32, ReLU) straight from the planar inputs -------------------------------------------------------------
// The layer-by-layer path first interleaves the nine input planes of fusion_net.py:38 (image 3, seg1, seg2, attention 2, time 2)
// into a 16-channel NHWC tensor (132 MB written and read again at 480p x 5 objects) and then runs the direct 3x3 kernel on it.
// Here the 10 x 34 patch is gathered from the planes themselves (coalesced row segments per plane; the three image planes are
// shared by all objects and stay in L2), split to fp16 hi / lo into a [pixel][16 + 8] LDS image whose channels 9..15 stay
// zero, and multiplied exactly like conv3x3_n32_direct_kernel<16> (same products, same order: bit-identical output).
constexpr int C1_TR = 8, C1_TC = 32, C1_PR = 10, C1_PC = 34, C1_PPX = 24, C1_PW = 24, C1_NPX = C1_PR * C1_PC;
static_assert(C1_NPX <= 512, "one thread per patch pixel");

struct FuseConv1P {
  const float *plane[9];
  long long nstride[9];
  float cval[9];
  const float *w, *scale, *bias;   // mivos_pack_weights_f16x3 rows of conv1 (Cin padded to 16: K = tap * 16 + c, kpad4 = 48)
  float *y;                        // [B][H][W][32]
  int B, H, W, tiles_x, tiles_y, n_tiles, contig;
};

__global__ __launch_bounds__(512) void fusion_conv1_kernel(const FuseConv1P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c1_smem[];
  constexpr int IMG = C1_NPX * C1_PPX;                         // halves of one patch image
  _Float16 *P0 = reinterpret_cast<_Float16 *>(c1_smem);        // two patch buffers, each hi | lo image
  _Float16 *Wh = P0 + 4 * IMG, *Wl = Wh + 9 * 32 * C1_PW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;

  for (int e = tid; e < 32 * 36; e += 512) {                   // weights: [32][kpad4 = 48] float4, 36 real quads (K = 144)
    const int n = e / 36, q = e - n * 36;
    const f32x4 v = reinterpret_cast<const f32x4 *>(p.w)[(long long)n * 48 + q];
    const int k = 4 * q, tap = k >> 4, c = k & 15;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 hh, ll;
    hh.x = v.x; hh.y = v.y; ll.x = v.z; ll.y = v.w;
    *reinterpret_cast<f32x2 *>(Wh + (tap * 32 + n) * C1_PW + c) = hh;
    *reinterpret_cast<f32x2 *>(Wl + (tap * 32 + n) * C1_PW + c) = ll;
  }

  // thread t < 340 owns patch pixel t: nine plane reads (neighbouring threads read neighbouring pixels of the same plane; the
  // plane index is a compile-time constant - a run-time index into the by-value parameter block would put it into scratch
  // memory), then the pixel's 16 hi and 16 lo halves (channels 9..15 zero) go to LDS as four 16-byte stores
  float pre[9];
  auto tile_coords = [&](int tile, int &img, int &y0, int &x0) {
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    img = tile / p.tiles_y;
    y0 = ty * C1_TR;
    x0 = tx * C1_TC;
  };
  auto load_patch = [&](int tile) {
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    const int pr = tid / C1_PC, pc = tid - pr * C1_PC;
    const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
    const bool ok = tid < C1_NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const long long pix = ok ? (long long)iy * p.W + ix : 0ll;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const float *src = p.plane[c];
      float v = 0.f;
      if (ok) v = src ? src[(long long)img * p.nstride[c] + pix] : p.cval[c];
      pre[c] = v;                                              // (outside the image every channel is zero: the padding of the concatenation)
    }
  };
  auto write_patch = [&](_Float16 *Ph) {
    if (tid < C1_NPX) {
      _Float16 *Pl = Ph + IMG;
      fh8 h0, l0, h1 = {0, 0, 0, 0, 0, 0, 0, 0}, l1 = h1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        h0[c] = (_Float16)pre[c];
        l0[c] = (_Float16)(pre[c] - (float)h0[c]);
      }
      h1[0] = (_Float16)pre[8];
      l1[0] = (_Float16)(pre[8] - (float)h1[0]);
      *reinterpret_cast<fh8 *>(Ph + tid * C1_PPX) = h0;
      *reinterpret_cast<fh8 *>(Ph + tid * C1_PPX + 8) = h1;
      *reinterpret_cast<fh8 *>(Pl + tid * C1_PPX) = l0;
      *reinterpret_cast<fh8 *>(Pl + tid * C1_PPX + 8) = l1;
    }
  };

  // (see fusion_resblock_kernel) Persistent walk, XCD-contiguous: workgroup b takes ONE contiguous run of tiles and the runs of the workgroups sharing an XCD (hardware
  // block b runs on XCD b % 8) are contiguous, so the halo rows / columns neighbouring tiles share are re-read from that XCD's L2 (or by
  // the same CU) instead of through the fabric by another XCD (PMC FETCH_SIZE 2.5 - 5 x the input before, profiles/r04d_config3_pmc_traffic.json)
  const int per_wg = (p.n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tstep = p.contig ? 1 : (int)gridDim.x;               // (contig = 0: the round-robin walk of rounds 1-3, A/B only)
  int tile = p.contig ? xcd_remap((int)blockIdx.x, (int)gridDim.x) * per_wg : (int)blockIdx.x;
  const int tile_end = !p.contig ? p.n_tiles : (tile + per_wg < p.n_tiles ? tile + per_wg : p.n_tiles);
  if (tile < tile_end) { load_patch(tile); write_patch(P0); }
  if (tile + tstep < tile_end) load_patch(tile + tstep);
  __syncthreads();
  for (int it = 0; tile < tile_end; tile += tstep, ++it) {
    const _Float16 *Ph = P0 + (it & 1) * 2 * IMG, *Pl = Ph + IMG;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int aoff = ((wave + kh) * C1_PC + i + kw) * C1_PPX + 8 * h;
        const int boff = ((kh * 3 + kw) * 32 + i) * C1_PW + 8 * h;
        const fh8 ah = *reinterpret_cast<const fh8 *>(Ph + aoff), al = *reinterpret_cast<const fh8 *>(Pl + aoff);
        const fh8 bh = *reinterpret_cast<const fh8 *>(Wh + boff), bl = *reinterpret_cast<const fh8 *>(Wl + boff);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc, 0, 0, 0);
      }
    }
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    const int y = y0 + wave, x = x0 + i;
    if (y < p.H && x < p.W) {
      float *yp = p.y + (((long long)img * p.H + y) * p.W + x) * 32;
      f32x4 out[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * h;
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(p.scale + c);
        const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v;
        v.x = fmaxf(acc[4 * g] * s4.x + b4.x, 0.f); v.y = fmaxf(acc[4 * g + 1] * s4.y + b4.y, 0.f);
        v.z = fmaxf(acc[4 * g + 2] * s4.z + b4.z, 0.f); v.w = fmaxf(acc[4 * g + 3] * s4.w + b4.w, 0.f);
        out[g] = v;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(yp + 8 * g + 4 * h) = out[g];
    }
    if (tile + tstep < tile_end) write_patch(P0 + ((it + 1) & 1) * 2 * IMG);
    if (tile + 2 * tstep < tile_end) load_patch(tile + 2 * tstep);
    __syncthreads();
  }
}

// ---- head: 3x3 / pad 1 / 32 -> 1 in exact fp32 ------------------------------------------------------------------------
constexpr int HD_TR = 8, HD_TC = 32, HD_PR = HD_TR + 2, HD_PC = HD_TC + 2, HD_PITCH = 36;   // floats per patch pixel (32 + 4 pad)

__global__ __launch_bounds__(256, 3) void fusion_head_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ out, int H, int W, int tiles_x, int tiles_y, int contig) {
  __shared__ __attribute__((aligned(16))) float patch[HD_PR * HD_PC * HD_PITCH];
  const int tid = threadIdx.x;
  int t = contig ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;     // neighbouring tiles (shared halo) on one XCD
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int img = t / tiles_y;
  const float *xb = x + (long long)img * H * W * 32;
  for (int e = tid; e < HD_PR * HD_PC * 8; e += 256) {
    const int px = e >> 3, c4 = e & 7;
    const int pr = px / HD_PC, pc = px - pr * HD_PC;
    const int iy = ty * HD_TR + pr - 1, ix = tx * HD_TC + pc - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4 *>(xb + ((long long)iy * W + ix) * 32 + 4 * c4);
    *reinterpret_cast<f32x4 *>(patch + px * HD_PITCH + 4 * c4) = v;
  }
  __syncthreads();
  const int r = tid >> 5, c = tid & 31;
  const int y = ty * HD_TR + r, xx = tx * HD_TC + c;
  // four partial sums (taps in OHWI order, channels ascending within a tap) keep the dependent-FMA chains short
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 3                                        // one kernel row per iteration (fully unrolled the compiler hoists all 72 LDS reads: 296 VGPRs)
  for (int tap = 0; tap < 9; ++tap) {
    const float *pp = patch + ((r + tap / 3) * HD_PC + c + tap % 3) * HD_PITCH;
    const float *wt = w + tap * 32;                     // uniform addresses: scalar loads
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(pp + 4 * q);
      a0 = __builtin_fmaf(v.x, wt[4 * q], a0);
      a1 = __builtin_fmaf(v.y, wt[4 * q + 1], a1);
      a2 = __builtin_fmaf(v.z, wt[4 * q + 2], a2);
      a3 = __builtin_fmaf(v.w, wt[4 * q + 3], a3);
    }
  }
  if (y < H && xx < W) out[((long long)img * H + y) * W + xx] = ((a0 + a1) + (a2 + a3)) + (bias ? bias[0] : 0.f);
}

static int launch_resblock(const float *x, float *y, const mivos_fusion_layer &a, const mivos_fusion_layer &b, int batch, int H, int W, hipStream_t st) {
  FuseBlockP p;
  p.x = x; p.y = y;
  p.wa = (const float *)a.w16; p.wb = (const float *)b.w16; p.sa = a.scale16; p.sb = b.scale16; p.ba = a.bias; p.bb = b.bias;
  p.B = batch; p.H = H; p.W = W;
  p.kpad4 = 80;                                          // K = 9 * 32 = 288 padded to the packer's 64-deep stages: 320 / 4
  p.tiles_x = cdiv(W, FB_TC); p.tiles_y = cdiv(H, FB_TR);
  const long long n_tiles = (long long)p.tiles_x * p.tiles_y * batch;
  if (n_tiles > 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_resblock: too many tiles");
  p.n_tiles = (int)n_tiles;
  p.contig = xcd_contig() >= 2;
  const size_t lds = (size_t)FB_LDS_HALVES * 2;
  static std::atomic<uint64_t> attr_mask{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(fusion_resblock_kernel), lds, attr_mask, "fusion_resblock")) return rc;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
  const int grid = p.n_tiles < cus ? p.n_tiles : cus;   // persistent: one workgroup per CU walks the tiles
  hipLaunchKernelGGL(fusion_resblock_kernel, dim3(grid), dim3(512), lds, st, p);
  return check_launch("fusion_resblock");
}

static int launch_conv1_planes(const mivos_interleave_desc &pl, const mivos_fusion_layer &L, float *y, int batch, int H, int W, hipStream_t st) {
  FuseConv1P p;
  for (int c = 0; c < 9; ++c) { p.plane[c] = pl.plane[c]; p.nstride[c] = pl.nstride[c]; p.cval[c] = pl.cval[c]; }
  p.w = (const float *)L.w16; p.scale = L.scale16; p.bias = L.bias; p.y = y;
  p.B = batch; p.H = H; p.W = W;
  p.tiles_x = cdiv(W, C1_TC); p.tiles_y = cdiv(H, C1_TR);
  const long long n_tiles = (long long)p.tiles_x * p.tiles_y * batch;
  if (n_tiles > 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_conv1: too many tiles");
  p.n_tiles = (int)n_tiles;
  p.contig = xcd_contig() >= 2;
  const size_t lds = (size_t)(4 * C1_NPX * C1_PPX + 2 * 9 * 32 * C1_PW) * 2;
  static std::atomic<uint64_t> attr_mask{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(fusion_conv1_kernel), lds, attr_mask, "fusion_conv1")) return rc;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
  const int grid = p.n_tiles < cus ? p.n_tiles : cus;
  hipLaunchKernelGGL(fusion_conv1_kernel, dim3(grid), dim3(512), lds, st, p);
  return check_launch("fusion_conv1");
}

static int launch_head(const float *x, const float *w, const float *bias, float *out, int batch, int H, int W, hipStream_t st) {
  const int tiles_x = cdiv(W, HD_TC), tiles_y = cdiv(H, HD_TR);
  const long long n = (long long)tiles_x * tiles_y * batch;
  if (n > 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_head: too many tiles");
  hipLaunchKernelGGL(fusion_head_kernel, dim3((unsigned)n), dim3(256), 0, st, x, w, bias, out, H, W, tiles_x, tiles_y, xcd_contig() >= 1);
  return check_launch("fusion_head");
}

}  // namespace mivos

using namespace mivos;

namespace {

// one precision-1 convolution on dense NHWC tensors (conv1 of the network)
int conv(const mivos_fusion_net_desc &d, const mivos_fusion_layer &L, const float *x, int cin, float *y, int cout, int k,
         const float *res, int relu_out, void *stream) {
  mivos_conv_desc c = {};
  c.x = x; c.w = (const float *)L.w16; c.scale = L.scale16; c.bias = L.bias; c.res = res; c.y = y; c.y2 = nullptr;
  c.N = d.batch; c.H = d.height; c.W = d.width; c.Cin = cin; c.Cout = cout; c.KH = k; c.KW = k; c.stride = 1; c.pad = k / 2;
  c.Ho = d.height; c.Wo = d.width; c.split = cout; c.relu_in = 0; c.relu_out = relu_out; c.precision = 1;
  const int64_t px = (int64_t)d.height * d.width;
  c.x_nstride = px * cin; c.x_pstride = cin;
  c.y_nstride = px * cout; c.y_pstride = cout;
  c.res_nstride = px * cout; c.res_pstride = cout;
  c.workspace = d.workspace; c.workspace_bytes = d.workspace_bytes;
  c.dilation = 1;
  return mivos_conv2d_fused(&c, stream);
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int mivos_fusion_resblock(const float *x, float *y, const mivos_fusion_layer *conv_a, const mivos_fusion_layer *conv_b, int batch,
                                     int height, int width, void *stream) {
  if (!x || !y || !conv_a || !conv_b || x == y || batch < 1 || height < 1 || width < 1 || !conv_a->w16 || !conv_a->scale16 || !conv_b->w16 ||
      !conv_b->scale16 || !aligned16(x) || !aligned16(y) || !aligned16(conv_a->w16) || !aligned16(conv_b->w16) || !aligned16(conv_a->scale16) ||
      !aligned16(conv_b->scale16) || !aligned16(conv_a->bias) || !aligned16(conv_b->bias))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_resblock: null / misaligned pointer, in-place call or bad sizes");
  return launch_resblock(x, y, *conv_a, *conv_b, batch, height, width, (hipStream_t)stream);
}

extern "C" int mivos_fusion_conv1_planes(const mivos_interleave_desc *planes, const mivos_fusion_layer *conv1, float *y, int batch, int height, int width,
                                         void *stream) {
  if (!planes || !conv1 || !y || batch < 1 || height < 1 || width < 1 || !conv1->w16 || !conv1->scale16 || !aligned16(y) || !aligned16(conv1->w16) ||
      !aligned16(conv1->scale16) || !aligned16(conv1->bias) || planes->C < 9)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_conv1_planes: null / misaligned pointer, fewer than 9 planes or bad sizes");
  return launch_conv1_planes(*planes, *conv1, y, batch, height, width, (hipStream_t)stream);
}

extern "C" int mivos_fusion_head(const float *x, const float *w_ohwi, const float *bias, float *logits, int batch, int height, int width, void *stream) {
  if (!x || !w_ohwi || !logits || batch < 1 || height < 1 || width < 1 || !aligned16(x))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_head: null / misaligned pointer or bad sizes");
  return launch_head(x, w_ohwi, bias, logits, batch, height, width, (hipStream_t)stream);
}

extern "C" int64_t mivos_fusion_net_scratch_floats(int batch, int height, int width) {
  if (batch < 1 || height < 1 || width < 1) return 0;
  return 2ll * batch * height * width * 32;
}

extern "C" int mivos_fusion_net_forward(const mivos_fusion_net_desc *dp, void *stream) {
  if (!dp) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: null descriptor");
  const mivos_fusion_net_desc &d = *dp;
  if ((!d.x16 && !d.planes) || !d.logits || !d.scratch || !d.final_w || d.batch < 1 || d.height < 1 || d.width < 1)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: null pointer or bad sizes");
  if (d.scratch_floats < mivos_fusion_net_scratch_floats(d.batch, d.height, d.width))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: scratch too small (mivos_fusion_net_scratch_floats)");
  for (int i = 0; i < 5; ++i)
    if (!d.layer[i].w16 || !d.layer[i].scale16) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: layer %d is not packed", i);
  const int64_t plane = (int64_t)d.batch * d.height * d.width * 32;
  float *A = d.scratch, *B = A + plane;
  if (d.planes) {                                                                                           // x = relu(conv1(cat))           fusion_net.py:38-40
    if (int rc = mivos_fusion_conv1_planes(d.planes, &d.layer[0], A, d.batch, d.height, d.width, stream)) return rc;
  } else if (int rc = conv(d, d.layer[0], d.x16, 16, A, 32, 3, nullptr, 1, stream)) return rc;
  if (int rc = mivos_fusion_resblock(A, B, &d.layer[1], &d.layer[2], d.batch, d.height, d.width, stream)) return rc;   // x = relu(x + conv2(x))  :42-43
  if (int rc = mivos_fusion_resblock(B, A, &d.layer[3], &d.layer[4], d.batch, d.height, d.width, stream)) return rc;   // x = relu(x + conv3(x))  :45-46
  return mivos_fusion_head(A, d.final_w, d.final_bias, d.logits, d.batch, d.height, d.width, stream);     // final_conv               :49
}
