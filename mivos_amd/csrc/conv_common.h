// Kernel-side parameter block shared by the convolution kernels.
#pragma once
#include "common.h"

namespace mivos {

struct ConvP {
  const float *x, *w, *scale, *bias, *res;
  float *y, *y2;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, split, relu_in, relu_out;
  int log2Cin, M, Ktot, HoWo, tiles_n;
  long long x_ns, x_ps, y_ns, y_ps, y2_ns, y2_ps, r_ns, r_ps;
};


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

// fills ConvP from the public descriptor after validating it; returns a status code
int conv_params_from_desc(const mivos_conv_desc *d, ConvP &p);
int launch_conv_f16x3(ConvP &p, hipStream_t st);
int select_variant_f16x3(int M, int Cout);

}  // namespace mivos
