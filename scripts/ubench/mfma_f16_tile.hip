// What a "tile" of the fp16 memory-read kernel costs, piece by piece: 24 v_mfma_f32_32x32x16_f16 from ONE wave per SIMD
// (4-wave workgroup, one workgroup per CU, all CUs busy) with distinct operand registers, NACC accumulator chains, and
// optionally the 16 ds_read_b128 of the fragment prefetch (results consumed one iteration later), a workgroup barrier per
// iteration, and 16 compare+branch pairs.  Reports s_memtime ticks per iteration of workgroup 0 AND the wall time per
// iteration, i.e. the clock the chip sustains under the load.
// hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_f16_tile.hip -o /tmp/mfma_f16_tile && /tmp/mfma_f16_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NACC, bool LDS, bool BAR, bool CMP, bool SMALL>
__global__ __launch_bounds__(256, 1) void tile(const float *in, long long *out, float *sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2][32 * 132];   // 528-byte row pitch: conflict-free ds_read_b128
  __shared__ float big[SMALL ? 1 : 28000];              // SMALL = false: one workgroup per CU by LDS, like the real kernel
  const int lane = threadIdx.x & 63;
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f4 a[16], b[16], g[16];
  for (int i = 0; i < 16; ++i) { a[i] = *(const f4 *)&in[(lane + 64 * i) * 4]; b[i] = *(const f4 *)&in[4096 + (lane + 64 * i) * 4]; g[i] = a[i]; }
  for (int i = threadIdx.x; i < 2 * 32 * 132; i += 256) (&lds[0][0])[i] = in[i & 4095];
  if (!SMALL) big[threadIdx.x] = 0.f;
  __syncthreads();
  float tau = in[0] + 1e30f;
  int cnt = 0;
  const long long t0 = __builtin_readcyclecounter();
  auto body = [&](int it, f4 (&F)[16], f4 (&G)[16]) {     // multiply fragment set F, prefetch the next tile's fragments into G
    const float *row = &lds[it & 1][(lane & 31) * 132 + 8 * (lane >> 5)];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (LDS) G[2 * ks] = *(const f4 *)(row + 16 * ks);
      __builtin_amdgcn_sched_barrier(0);
      acc[(3 * ks) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, F[2 * ks + 1]), __builtin_bit_cast(h8, b[2 * ks]), acc[(3 * ks) % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (CMP) { if (__ballot(acc[0][ks] > tau) != 0ull) cnt++; }
      __builtin_amdgcn_sched_barrier(0);
      acc[(3 * ks + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, F[2 * ks]), __builtin_bit_cast(h8, b[2 * ks + 1]), acc[(3 * ks + 1) % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (LDS) G[2 * ks + 1] = *(const f4 *)(row + 16 * ks + 4);
      __builtin_amdgcn_sched_barrier(0);
      acc[(3 * ks + 2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, F[2 * ks]), __builtin_bit_cast(h8, b[2 * ks]), acc[(3 * ks + 2) % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (CMP) { if (__ballot(acc[0][8 + ks] > tau) != 0ull) cnt++; }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  };
  for (int it = 0; it < iters; it += 2) {
    body(it, a, g);
    body(it + 1, g, a);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = (float)cnt;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.f) sink[0] = s + big[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}


// The same tile loop fed like the real kernel: every iteration requests 16 KB (4 LDS-DMA pieces of 1 KB per wave) two
// iterations ahead into a 3-buffer ring and waits for the older request before its barrier.  `span_tiles`: how many distinct
// tiles a workgroup walks before wrapping (small: L2 resident; large: streamed from HBM / MALL); SHARE workgroups read the
// same addresses (the real kernel: ~11 per XCD walk one object's keys together).
typedef __attribute__((address_space(3))) void *lds_ptr_t;
template <bool MFMA>
__global__ __launch_bounds__(256, 1) void stream(const float *in, const unsigned char *keys, long long *out, float *sink, int iters, int span_tiles, int share) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char ring[3][16384];
  __shared__ float big[28000];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f16v acc[3];
  for (int i = 0; i < 3; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f4 a[16], b[16], g[16];
  for (int i = 0; i < 16; ++i) { a[i] = *(const f4 *)&in[(lane + 64 * i) * 4]; b[i] = *(const f4 *)&in[4096 + (lane + 64 * i) * 4]; g[i] = a[i]; }
  big[threadIdx.x] = 0.f;
  const long long group = blockIdx.x / share;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(keys + group * (long long)span_tiles * 16384), 0, (unsigned)(span_tiles * 16384), 0x00020000);
  const int lds0 = __builtin_amdgcn_readfirstlane((int)(size_t)&ring[0][0]) + wave * 4096;
  const unsigned voff = (unsigned)(wave * 4096 + lane * 16);
  auto issue = [&](int it, int buf) {
    const int soff = (it % span_tiles) * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(size_t)(lds0 + buf * 16384 + i * 1024), 16, voff + i * 1024, soff, 0, 0);
  };
  issue(0, 0); issue(1, 1); issue(2, 2);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  const long long t0 = __builtin_readcyclecounter();
  auto body = [&](int it, f4 (&F)[16], f4 (&G)[16]) {
    issue(it + 3, buf);
    const int nb = buf == 2 ? 0 : buf + 1;
    const float *row = (const float *)&ring[nb][0] + (lane & 31) * 128 + (((8 * (lane >> 5)) ^ (4 * (lane & 15))));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      G[2 * ks] = *(const f4 *)(row + (16 * ks));
      __builtin_amdgcn_sched_barrier(0);
      if (MFMA) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, F[2 * ks + 1]), __builtin_bit_cast(h8, b[2 * ks]), acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MFMA) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, F[2 * ks]), __builtin_bit_cast(h8, b[2 * ks + 1]), acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      G[2 * ks + 1] = *(const f4 *)(row + (16 * ks) + 4);
      __builtin_amdgcn_sched_barrier(0);
      if (MFMA) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, F[2 * ks]), __builtin_bit_cast(h8, b[2 * ks]), acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = nb;
  };
  for (int it = 0; it < iters; it += 2) {
    body(it, a, g);
    body(it + 1, g, a);
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][15];
  for (int i = 0; i < 16; ++i) s += a[i].x + g[i].y;
  if (s == 12345.f) sink[0] = s + big[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
#endif
}

int main() {
  float *in, *sink;
  long long *out;
  hipMalloc(&in, 16384 * 4); hipMalloc(&sink, 4); hipMalloc(&out, 8);
  hipMemset(in, 0, 16384 * 4);
  const int iters = 4000;
  long long h;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(name, kern, grid)                                                                                   \
  { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, sink, iters); hipDeviceSynchronize();        \
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, sink, iters); hipEventRecord(e1); hipDeviceSynchronize(); \
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);                   \
    printf("%-64s %7.1f ticks / tile of 24 MFMAs (floor 768)   %6.3f us / tile  => %5.2f GHz   %6.0f TFLOP/s fp16 on %d CUs\n", name, (double)h / iters, \
           ms * 1e3 / iters, (double)h / iters / (ms * 1e3 / iters) * 1e-3, 24.0 * 32768 * 4 * grid / (ms * 1e-3 / iters) * 1e-12, grid); }
  RUN("MFMA only, 1 chain, 256 WGs", (tile<1, false, false, false, true>), 256)
  RUN("MFMA only, 2 chains, 256 WGs", (tile<2, false, false, false, true>), 256)
  RUN("MFMA only, 3 chains, 256 WGs", (tile<3, false, false, false, true>), 256)
  RUN("MFMA only, 3 chains, 1 WG (idle chip)", (tile<3, false, false, false, true>), 1)
  RUN("MFMA only, 3 chains, 64 WGs", (tile<3, false, false, false, true>), 64)
  RUN("+ 16 ds_read_b128, 3 chains, 256 WGs", (tile<3, true, false, false, true>), 256)
  RUN("+ 16 ds_read_b128 + barrier, 3 chains, 256 WGs", (tile<3, true, true, false, true>), 256)
  RUN("+ 16 ds_read_b128 + barrier + 16 cmp/branch, 3 chains, 256 WGs", (tile<3, true, true, true, true>), 256)
  RUN("same, 160 KB of LDS per workgroup", (tile<3, true, true, true, false>), 256)
  RUN("same, 1 WG (idle chip)", (tile<3, true, true, true, false>), 1)
  unsigned char *keys;
  const long long key_bytes = 3ll << 30;
  hipMalloc(&keys, key_bytes);
  hipMemset(keys, 0, key_bytes);
#define RUNS(name, kern, span, share)                                                                           \
  { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, keys, out, sink, iters, span, share); hipDeviceSynchronize();        \
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, keys, out, sink, iters, span, share); hipEventRecord(e1); hipDeviceSynchronize(); \
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);                   \
    printf("%-72s %7.1f ticks / tile   %6.3f us / tile  => %5.2f TB/s into the CUs\n", name, (double)h / iters, ms * 1e3 / iters, 256 * 16384.0 / (ms * 1e-3 / iters) * 1e-12); }
  RUNS("stream 16 KB / tile + MFMA, every WG its own 64 tiles (L2 resident)", (stream<true>), 64, 1)
  RUNS("stream only (no MFMA), every WG its own 64 tiles (L2 resident)", (stream<false>), 64, 1)
  RUNS("stream + MFMA, 256 WGs share one 64-tile span", (stream<true>), 64, 256)
  RUNS("stream + MFMA, groups of 11 WGs share a 4000-tile span (64 MB each)", (stream<true>), 4000, 11)
  RUNS("stream only, groups of 11 WGs share a 4000-tile span", (stream<false>), 4000, 11)
  RUNS("stream + MFMA, every WG its own 700-tile span (11 MB each, 2.9 GB total)", (stream<true>), 700, 1)
  RUNS("stream only, every WG its own 700-tile span", (stream<false>), 700, 1)
  return 0;
}
