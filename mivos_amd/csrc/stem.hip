// ResNet stem for gfx950: 7x7 / stride 2 / pad 3 convolution of up to 8 PLANAR input channels to 64 channels + folded BN + ReLU
// (modules.py:56-58 / 81-83: conv1, bn1, relu of the mask / query encoders; mod_resnet.py; torchvision resnet50).
//
// As an implicit GEMM (conv_f16x3_kernel<128,64>, K = 7*7*8 = 392) this layer ran at 0.16 of the f16x3 roofline: every input
// element is converted and staged 49/4 times, and the channel concatenation cat([frame, mask, others]) (modules.py:54) was a
// separate interleave pass.  Here a workgroup (8 waves) owns an 8 x 32 tile of output pixels, all 64 channels:
//   * the 21 x 69 input patch is gathered from the planes themselves (3 shared image planes + per-object planes: batch strides,
//     like mivos_interleave_planes), split to fp16 hi / lo ONCE into LDS as [pixel][8 channels] (16 bytes per pixel and image);
//   * a K step = one kernel row position pair (kh, kw = 2p + {0, 1}) x 8 channels = 16: the MFMA B fragment of lane (i, h) is the
//     16-byte pixel 2i + 2p + h of patch row 2r + kh - 64 lanes read one contiguous KB, conflict-free, no im2col arithmetic;
//     28 steps (the 8th tap of a row has zero weights);
//   * the packed weights of all steps ([step][64][16] hi and lo, 112 KB) are LDS-resident for the whole launch (the workgroup
//     is persistent); the A fragment of lane (i, h) is the 16-byte half h of row (step, channel i): again one contiguous KB;
//   * wave r computes output row r for both 32-channel blocks (the pixel fragments are shared): 6 fragment reads per 6 MFMAs,
//     software pipelined one step ahead; three v_mfma_f32_32x32x16_f16 per block (lo*hi + hi*lo + hi*hi, fp32 accumulate);
//   * epilogue: acc * scale + bias, ReLU, 16-byte stores (a lane owns 4 consecutive channels of one pixel);
//   * the next tile's patch is in flight in registers during the tile.
#include "conv_common.h"

namespace mivos {

typedef _Float16 sh8 __attribute__((ext_vector_type(8)));

constexpr int ST_TR = 8, ST_TC = 32;                       // output tile
constexpr int ST_PR = 2 * ST_TR + 5, ST_PC = 2 * ST_TC + 5;   // input patch 21 x 69
constexpr int ST_NPX = ST_PR * ST_PC;                      // 1449 pixels
constexpr int ST_STEPS = 28;                               // (kh, kw pair)
constexpr int ST_PATCH_HALVES = (ST_NPX + 8) * 8;          // one image (hi or lo) + slack for the zero-weight 8th tap
constexpr int ST_W_HALVES = ST_STEPS * 64 * 16;            // one weight image
constexpr int ST_LDS_BYTES = (2 * ST_PATCH_HALVES + 2 * ST_W_HALVES) * 2 + 2 * 64 * 4;
constexpr int ST_PPT = (ST_NPX + 511) / 512;               // patch pixels per thread
static_assert(ST_LDS_BYTES <= 160 * 1024, "LDS budget of one CU");

struct StemP {
  const float *plane[8];
  long long nstride[8];
  const float *w;                 // fp32 OHWI [64][7][7][cin]
  const float *scale, *bias;      // epilogue: acc * scale + bias (scale includes 1 / mult)
  float *y;                       // [N][Ho][Wo][64]
  float mult;                     // power-of-two pre-scaling of the fp16 weight split
  int n_planes, cin, N, H, W, Ho, Wo, tiles_x, tiles_y, n_tiles, contig;
};

__global__ __launch_bounds__(512) void stem7x7s2_kernel(const StemP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
  _Float16 *Ph = reinterpret_cast<_Float16 *>(st_smem), *Pl = Ph + ST_PATCH_HALVES;
  _Float16 *Wh = Pl + ST_PATCH_HALVES, *Wl = Wh + ST_W_HALVES;
  float *SB = reinterpret_cast<float *>(Wl + ST_W_HALVES);      // scale[64] | bias[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;

  // weights: fp32 OHWI -> [step = kh * 4 + p][n][h * 8 + c] hi / lo, tap kw = 2p + h (kw = 7: zero), channels >= cin: zero
  for (int e = tid; e < ST_STEPS * 64 * 2; e += 512) {
    const int hh = e & 1, n = (e >> 1) & 63, step = e >> 7;
    const int kh = step >> 2, kw = 2 * (step & 3) + hh;
    sh8 vh, vl;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = 0.f;
      if (kw < 7 && c < p.cin) v = p.w[((long long)(n * 7 + kh) * 7 + kw) * p.cin + c] * p.mult;
      const _Float16 hi = (_Float16)v;
      vh[c] = hi;
      vl[c] = (_Float16)(v - (float)hi);
    }
    *reinterpret_cast<sh8 *>(Wh + (step * 64 + n) * 16 + 8 * hh) = vh;
    *reinterpret_cast<sh8 *>(Wl + (step * 64 + n) * 16 + 8 * hh) = vl;
  }
  if (tid < 128) SB[tid] = tid < 64 ? p.scale[tid] : (p.bias ? p.bias[tid - 64] : 0.f);
  if (tid < 128) {                                            // slack pixels behind the patch images (read with zero weights only)
    reinterpret_cast<uint32_t *>(Ph + ST_NPX * 8)[tid & 31] = 0u;
    reinterpret_cast<uint32_t *>(Pl + ST_NPX * 8)[tid & 31] = 0u;
  }

  float pre[ST_PPT][8];
  auto tile_coords = [&](int tile, int &img, int &y0, int &x0) {
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    img = tile / p.tiles_y;
    y0 = ty * ST_TR;
    x0 = tx * ST_TC;
  };
  // thread t owns patch pixels t, t + 512, t + 1024: one 4-byte read per plane (neighbouring threads: neighbouring pixels)
  auto load_patch = [&](int tile) {
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
#pragma unroll
    for (int l = 0; l < ST_PPT; ++l) {
      const int px = tid + 512 * l;
      const int pr = px / ST_PC, pc = px - pr * ST_PC;
      const int iy = 2 * y0 - 3 + pr, ix = 2 * x0 - 3 + pc;
      const bool ok = px < ST_NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long long pix = ok ? (long long)iy * p.W + ix : 0ll;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = 0.f;
        if (ok && c < p.n_planes) v = p.plane[c][(long long)img * p.nstride[c] + pix];
        pre[l][c] = v;
      }
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int l = 0; l < ST_PPT; ++l) {
      const int px = tid + 512 * l;
      if (px < ST_NPX) {
        sh8 vh, vl;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const _Float16 hi = (_Float16)pre[l][c];
          vh[c] = hi;
          vl[c] = (_Float16)(pre[l][c] - (float)hi);
        }
        *reinterpret_cast<sh8 *>(Ph + px * 8) = vh;
        *reinterpret_cast<sh8 *>(Pl + px * 8) = vl;
      }
    }
  };

  // (see fusion_resblock_kernel) Persistent walk, XCD-contiguous: workgroup b takes ONE contiguous run of tiles and the runs of the workgroups sharing an XCD (hardware
  // block b runs on XCD b % 8) are contiguous, so the halo rows / columns neighbouring tiles share are re-read from that XCD's L2 (or by
  // the same CU) instead of through the fabric by another XCD (PMC FETCH_SIZE 2.5 - 5 x the input before, profiles/r04d_config3_pmc_traffic.json)
  const int per_wg = (p.n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tstep = p.contig ? 1 : (int)gridDim.x;               // (contig = 0: the round-robin walk of rounds 1-3, A/B only)
  int tile = p.contig ? xcd_remap((int)blockIdx.x, (int)gridDim.x) * per_wg : (int)blockIdx.x;
  const int tile_end = !p.contig ? p.n_tiles : (tile + per_wg < p.n_tiles ? tile + per_wg : p.n_tiles);
  if (tile < tile_end) { load_patch(tile); write_patch(); }
  if (tile + tstep < tile_end) load_patch(tile + tstep);
  __syncthreads();
  for (; tile < tile_end; tile += tstep) {
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    // step s = kh * 4 + pp: pixel fragment = patch pixel (2 wave + kh, 2 i + 2 pp + h), weight fragments = rows (s, cb * 32 + i), half h
    sh8 xh[2], xl[2], wh[2][2], wl[2][2];
    auto load = [&](int s, int set) {
      const int kh = s >> 2, pp = s & 3;
      const int aoff = ((2 * wave + kh) * ST_PC + 2 * i + 2 * pp + h) * 8;
      xh[set] = *reinterpret_cast<const sh8 *>(Ph + aoff);
      xl[set] = *reinterpret_cast<const sh8 *>(Pl + aoff);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int boff = (s * 64 + cb * 32 + i) * 16 + 8 * h;
        wh[set][cb] = *reinterpret_cast<const sh8 *>(Wh + boff);
        wl[set][cb] = *reinterpret_cast<const sh8 *>(Wl + boff);
      }
    };
    load(0, 0);
#pragma unroll
    for (int s = 0; s < ST_STEPS; ++s) {
      const int cur = s & 1;
      if (s + 1 < ST_STEPS) load(s + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {                      // weights are the MFMA "A" operand: D[channel][pixel]; small terms first
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur][cb], xl[cur], acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cur][cb], xh[cur], acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur][cb], xh[cur], acc[cb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    const int oy = y0 + wave, ox = x0 + i;
    if (oy < p.Ho && ox < p.Wo) {
      float *yp = p.y + (((long long)img * p.Ho + oy) * p.Wo + ox) * 64;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cb * 32 + 8 * g + 4 * h;
          const f32x4 s4 = *reinterpret_cast<const f32x4 *>(SB + c), b4 = *reinterpret_cast<const f32x4 *>(SB + 64 + c);
          f32x4 v;
          v.x = fmaxf(__builtin_fmaf(acc[cb][4 * g], s4.x, b4.x), 0.f);
          v.y = fmaxf(__builtin_fmaf(acc[cb][4 * g + 1], s4.y, b4.y), 0.f);
          v.z = fmaxf(__builtin_fmaf(acc[cb][4 * g + 2], s4.z, b4.z), 0.f);
          v.w = fmaxf(__builtin_fmaf(acc[cb][4 * g + 3], s4.w, b4.w), 0.f);
          *reinterpret_cast<f32x4 *>(yp + c) = v;
        }
    }
    __syncthreads();                                        // every wave is done reading the patch
    if (tile + tstep < tile_end) write_patch();
    if (tile + 2 * tstep < tile_end) load_patch(tile + 2 * tstep);
    __syncthreads();
  }
}

}  // namespace mivos

using namespace mivos;

extern "C" int mivos_stem7x7s2_planes(const mivos_interleave_desc *planes, int n_planes, const float *w_ohwi, int cin, float mult, const float *scale,
                                      const float *bias, float *y, int N, int H, int W, void *stream) {
  if (!planes || !w_ohwi || !scale || !y || n_planes < 1 || n_planes > 8 || cin < n_planes || cin > 8 || N < 1 || H < 1 || W < 1 || ((uintptr_t)y & 15) ||
      ((uintptr_t)scale & 15) || ((uintptr_t)bias & 15) || !(mult > 0.f))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "stem7x7s2_planes: null / misaligned pointer or bad sizes (1..8 planes, cin <= 8)");
  StemP p;
  for (int c = 0; c < 8; ++c) {
    p.plane[c] = c < n_planes ? planes->plane[c] : nullptr;
    p.nstride[c] = c < n_planes ? planes->nstride[c] : 0;
    if (c < n_planes && !p.plane[c]) return fail(MIVOS_ERR_INVALID_ARGUMENT, "stem7x7s2_planes: plane %d is NULL (constant planes are not supported here)", c);
  }
  p.w = w_ohwi; p.scale = scale; p.bias = bias; p.y = y; p.mult = mult;
  p.n_planes = n_planes; p.cin = cin; p.N = N; p.H = H; p.W = W;
  p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1;
  p.tiles_x = cdiv(p.Wo, ST_TC); p.tiles_y = cdiv(p.Ho, ST_TR);
  const long long n_tiles = (long long)p.tiles_x * p.tiles_y * N;
  if (n_tiles > 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "stem7x7s2_planes: too many tiles");
  p.n_tiles = (int)n_tiles;
  p.contig = xcd_contig() >= 2;
  static std::atomic<uint64_t> attr_mask{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(stem7x7s2_kernel), ST_LDS_BYTES, attr_mask, "stem7x7s2")) return rc;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
  const int grid = p.n_tiles < cus ? p.n_tiles : cus;       // persistent: the 112 KB of packed weights are built once per workgroup
  hipLaunchKernelGGL(stem7x7s2_kernel, dim3(grid), dim3(512), ST_LDS_BYTES, (hipStream_t)stream, p);
  return check_launch("stem7x7s2");
}
