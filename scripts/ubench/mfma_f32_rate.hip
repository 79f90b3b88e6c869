// Issue rate of the exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32) from ONE wave per SIMD as a
// function of the number of independent accumulators, with distinct A/B operand registers per instruction (as in a real
// K loop) and optionally two ds_read_b128 per 8 MFMAs (the memory-read kernel's fragment prefetch).
// hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_f32_rate.hip -o /tmp/mfma_f32_rate && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256, 1) void k16(const float *in, long long *out, float *sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63;
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[lane + 64 * i]; b[i] = in[512 + lane + 64 * i]; }
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  f4 g0 = {0, 0, 0, 0}, g1 = g0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {               // 8 MFMAs per group, round-robin over the accumulators
      if (LDS && u == 0) { g0 = *(const f4 *)&lds[lane * 4 + (it & 7) * 256]; g1 = *(const f4 *)&lds[4096 + lane * 4 + (it & 7) * 256]; }
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[u % NACC], 0, 0, 0);
    }
    if (LDS) { a[0] += g0.x * 1e-30f + g1.y * 1e-30f; }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].w;
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void k32(const float *in, long long *out, float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[lane + 64 * i]; b[i] = in[512 + lane + 64 * i]; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u % NACC], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  float *in, *sink;
  long long *out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&sink, 4); hipMalloc(&out, 8);
  hipMemset(in, 0, 8192 * 4);
  const int iters = 2000;
  long long h;
#define RUN(name, kern, per)                                                                  \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, out, sink, iters); hipDeviceSynchronize(); } \
  hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);                                               \
  printf("%-44s %7.1f cycles per MFMA (issue floor %d)\n", name, (double)h / (iters * 8.0), per);
  RUN("16x16x4, 1 accumulator", (k16<1, false>), 32)
  RUN("16x16x4, 2 accumulators", (k16<2, false>), 32)
  RUN("16x16x4, 4 accumulators", (k16<4, false>), 32)
  RUN("16x16x4, 8 accumulators", (k16<8, false>), 32)
  RUN("16x16x4, 2 accumulators + 2 ds_read_b128 / 8", (k16<2, true>), 32)
  RUN("16x16x4, 4 accumulators + 2 ds_read_b128 / 8", (k16<4, true>), 32)
  RUN("32x32x2, 1 accumulator", (k32<1>), 64)
  RUN("32x32x2, 2 accumulators", (k32<2>), 64)
  RUN("32x32x2, 4 accumulators", (k32<4>), 64)
  return 0;
}
