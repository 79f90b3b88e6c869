// How v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 round: D = C + sum_k a_k b_k with hand-picked operands (every row of A holds
// a_k, every column of B holds b_k, so all outputs are the same number).  The f16x3 kernels rely on small products (hi * lo, 2^-11 of
// the hi * hi terms) surviving next to a large accumulator; this prints what the matrix pipe does with them.
// hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_f16_rounding.hip -o scripts/ubench/mfma_f16_rounding && scripts/ubench/mfma_f16_rounding
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ void k32(const float *a, const float *b, float c, float *out) {   // a, b: 16 values each
  const int lane = threadIdx.x & 63;
  h8 A, B;
  for (int j = 0; j < 8; ++j) { A[j] = (_Float16)a[8 * (lane >> 5) + j]; B[j] = (_Float16)b[8 * (lane >> 5) + j]; }
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = c;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
  if (lane == 37) out[1] = acc[9];
}
__global__ void k16(const float *a, const float *b, float c, float *out) {   // a, b: 32 values each
  const int lane = threadIdx.x & 63;
  h8 A, B;
  for (int j = 0; j < 8; ++j) { A[j] = (_Float16)a[8 * (lane >> 4) + j]; B[j] = (_Float16)b[8 * (lane >> 4) + j]; }
  f4v acc = {c, c, c, c};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
  if (lane == 37) out[1] = acc[3];
}

static void run(const char *name, const float *a, const float *b, float c, double exact) {
  float *da, *db, *dout, h[4];
  hipMalloc(&da, 32 * 4); hipMalloc(&db, 32 * 4); hipMalloc(&dout, 16);
  hipMemcpy(da, a, 32 * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, da, db, c, dout);
  hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, da, db, c, dout);
  hipMemcpy(h + 2, dout, 8, hipMemcpyDeviceToHost);
  const float rne = (float)exact;
  printf("%-58s exact %.10e (RNE %a)  32x32x16: %a %a  16x16x32(first 32 k): %a %a\n", name, exact, rne, h[0], h[1], h[2], h[3]);
  hipFree(da); hipFree(db); hipFree(dout);
}

int main() {
  float a[32], b[32];
  auto zero = [&]() { for (int i = 0; i < 32; ++i) a[i] = b[i] = 0.f; };
  zero(); a[0] = 1.5f * ldexpf(1.f, -12); b[0] = ldexpf(1.f, -12);
  run("C=1, one product 1.5*2^-24 (RNE: 1+2^-23, truncation: 1)", a, b, 1.f, 1.0 + 1.5 * ldexp(1.0, -24));
  zero(); for (int i = 0; i < 16; ++i) { a[i] = ldexpf(1.f, -13); b[i] = ldexpf(1.f, -13); }
  run("C=1, 16 products of 2^-26 (exact sum: 1+2^-22)", a, b, 1.f, 1.0 + ldexp(1.0, -22));
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = ldexpf(1.f, -13); b[1] = ldexpf(1.f, -12); a[2] = -1.f; b[2] = 1.f;
  run("C=0, products 1, 2^-25, -1 (exact: 2^-25)", a, b, 0.f, ldexp(1.0, -25));
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = ldexpf(1.f, -13); b[1] = ldexpf(1.f, -12); a[8] = -1.f; b[8] = 1.f;
  run("  the same with the -1 in the second 8-group", a, b, 0.f, ldexp(1.0, -25));
  zero(); a[0] = 1.f; b[0] = 1.f; for (int i = 1; i < 16; ++i) { a[i] = ldexpf(1.f, -14); b[i] = ldexpf(1.f, -14); }
  run("C=0, 1 + 15 products of 2^-28 (exact: 1+15*2^-28)", a, b, 0.f, 1.0 + 15 * ldexp(1.0, -28));
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = -1.5f * ldexpf(1.f, -12); b[1] = ldexpf(1.f, -13);
  run("C=0, 1 - 1.5*2^-25 (RNE: 1-2^-24, trunc toward 0: 1-2^-24, toward -inf ...)", a, b, 0.f, 1.0 - 1.5 * ldexp(1.0, -25));
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = -1.f * ldexpf(1.f, -13); b[1] = ldexpf(1.f, -13);
  run("C=0, 1 - 2^-26 (RNE: 1, truncation toward zero: 1-2^-24)", a, b, 0.f, 1.0 - ldexp(1.0, -26));
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = 1.f * ldexpf(1.f, -13); b[1] = ldexpf(1.f, -13);
  run("C=0, 1 + 2^-26 (RNE: 1, round up: 1+2^-23)", a, b, 0.f, 1.0 + ldexp(1.0, -26));
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = 1.f * ldexpf(1.f, -12); b[1] = ldexpf(1.f, -12); a[2] = 1.f * ldexpf(1.f, -13); b[2] = ldexpf(1.f, -13);
  run("C=0, 1 + 2^-24 + 2^-26 (RNE: 1+2^-23; sticky lost: 1)", a, b, 0.f, 1.0 + ldexp(1.0, -24) + ldexp(1.0, -26));
  // fp16 subnormal operands (|v| < 2^-14): the lo halves of the f16x3 split are subnormal for every |x| < 2^-3
  zero(); a[0] = ldexpf(1.f, -20); b[0] = 1.f;
  run("C=0, subnormal a = 2^-20 times 1 (kept: 2^-20, flushed: 0)", a, b, 0.f, ldexp(1.0, -20));
  zero(); a[0] = 1.f; b[0] = ldexpf(410.f, -24);
  run("C=0, 1 times subnormal b = 410 * 2^-24 (lo of x = 0.1)", a, b, 0.f, 410.0 * ldexp(1.0, -24));
  zero(); a[0] = 0.0999755859375f; b[0] = 1.f; a[1] = ldexpf(410.f, -24); b[1] = 1.f;
  run("C=0, hi(0.1) + lo(0.1) as two products (kept: 0.1)", a, b, 0.f, 0.0999755859375 + 410.0 * ldexp(1.0, -24));
  zero(); a[0] = ldexpf(1.f, -20); b[0] = ldexpf(1.f, -4);
  run("C=0, subnormal 2^-20 times 2^-4 (product 2^-24)", a, b, 0.f, ldexp(1.0, -24));
  return 0;
}
