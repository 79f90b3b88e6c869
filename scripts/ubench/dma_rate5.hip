// LDS-DMA issue rate of 4 waves (one per SIMD) while the other 4 waves of the workgroup run back-to-back MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void *lds_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mfma_mode 0: partner waves idle (exit); 1: MFMA loop, prio 0; 2: MFMA loop with s_setprio(1); 3: MFMA loop, 8 independent accumulators, prio 1
// valu_pad: extra VALU instructions between DMA pieces (address math stand-in)
__global__ __launch_bounds__(512) void k(const unsigned char *base, int mfma_mode, int valu_pad_in, int iters, long long *out, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int valu_pad = valu_pad_in;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (mfma_mode == 0) return;
    f32x16 a0, a1, a2, a3, a4, a5, a6, a7;
    for (int r = 0; r < 16; ++r) { a0[r] = a1[r] = a2[r] = a3[r] = a4[r] = a5[r] = a6[r] = a7[r] = 0.f; }
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 2, 2, 2, 2};
    if (mfma_mode >= 2) __builtin_amdgcn_s_setprio(1);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters * 4; ++it) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a5, 0, 0, 0);
      a6 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a6, 0, 0, 0); a7 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, a7, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    return;
  }
  const unsigned char *priv = base + (long long)blockIdx.x * (1 << 20);
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)priv, 0, 1 << 20, 0x00020000);
  long long t0 = __builtin_readcyclecounter();
  int pos = wave * 8;
  int junk = lane;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 racc = {0.f, 0.f, 0.f, 0.f};
  const int ds_mode = valu_pad >= 100 ? valu_pad - 99 : 0;     // 1: ds_reads before the pieces, 2: after, 3: ds_reads only
  if (ds_mode) valu_pad = 0;
  auto ds12 = [&]() {
    f4 v[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) v[q] = *reinterpret_cast<const f4 *>(smem + 32768 + ((q * 64 + lane) * 16 + (junk & 16)));
#pragma unroll
    for (int q = 0; q < 12; ++q) racc += v[q];
  };
  for (int it = 0; it < iters; ++it) {
    if (ds_mode == 1 || ds_mode == 3) ds12();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ds_mode == 3) break;
      for (int v = 0; v < valu_pad; ++v) junk = (junk + 1) ^ v;
      const int off = (((pos + j) * 1024 + lane * 16) & 65535) + (junk == 0x7fffffff ? 16 : 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_t)(smem + (wave * 8 + j) * 1024), 16, off, 0, 0, 0);
    }
    if (ds_mode == 2) ds12();
    pos += 32;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  if (racc.x + racc.y + racc.z + racc.w == 12345.f) sink[1] = racc.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned char *buf; long long *out; float *sink;
  hipMalloc(&buf, 256ull << 20); hipMemset(buf, 1, 256ull << 20); hipMalloc(&out, 64 * 8); hipMalloc(&sink, 16);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 400;
  const char *mn[] = {"partners idle", "partners MFMA (8 acc) prio 0", "partners MFMA (8 acc) prio 1"};
  for (int mode : {0, 1, 2})
    for (int pad : {0, 100, 101, 102}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(out, 0, 64);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, buf, mode, pad, iters, out, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      const double n = (double)iters * 8;
      printf("%-30s valu_pad %2d: DMA wave %6.1f cyc/piece (%5.1f B/clk/CU for 4 waves) | MFMA wave %6.1f cyc/MFMA | wall %7.1f us\n", mn[mode], pad, h[0] / n,
             4 * 1024 * n / h[0], mode ? h[4] / (double)(iters * 32) : 0.0, ms * 1e3);
    }
  return 0;
}
