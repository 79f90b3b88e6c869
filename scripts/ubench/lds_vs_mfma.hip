// How fast can a wave run ds_read_b128 / VALU / LDS-DMA while its SIMD partner issues back-to-back MFMAs?
// (the LOAD-phase / MFMA-phase ping-pong of the f16x3 GEMM).  Partner accumulators in VGPRs or AGPRs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA_V(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// partner: 0 idle, 1 MFMA acc in VGPR, 2 MFMA acc in AGPR.   work: 0 = 12 ds_read_b128, 1 = 48 v_add_f32, 2 = 12 ds_read_b64 x2?? (unused), 3 = 12 ds_reads + 48 VALU
__global__ __launch_bounds__(512) void k(const unsigned char *gbuf, int partner, int work, int iters, long long *out, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (partner == 0) return;
    f32x16 a0, a1, a2, a3, a4, a5, a6, a7;
    for (int r = 0; r < 16; ++r) { a0[r] = a1[r] = a2[r] = a3[r] = a4[r] = a5[r] = a6[r] = a7[r] = 0.f; }
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 2, 2, 2, 2};
    long long t0 = __builtin_readcyclecounter();
    if (partner == 1) {
      for (int it = 0; it < iters * 6; ++it) { MFMA_V(a0); MFMA_V(a1); MFMA_V(a2); MFMA_V(a3); MFMA_V(a4); MFMA_V(a5); MFMA_V(a6); MFMA_V(a7); }
    } else {
      for (int it = 0; it < iters * 6; ++it) { MFMA_A(a0); MFMA_A(a1); MFMA_A(a2); MFMA_A(a3); MFMA_A(a4); MFMA_A(a5); MFMA_A(a6); MFMA_A(a7); }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    return;
  }
  const unsigned addr = (unsigned)(size_t)(smem) + lane * 16;
  f4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11;
  float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f, w0 = 0.f, w1 = 1.f, w2 = 2.f, w3 = 3.f;
  int sreg = 0;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(gbuf + (long long)blockIdx.x * 65536), 0, 1 << 20, 0x00020000);
  const int voff = lane * 16;
  int soff = __builtin_amdgcn_readfirstlane(wave * 4096);
  const int lds_s = __builtin_amdgcn_readfirstlane((int)(size_t)smem + wave * 8192);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (work == 0 || work == 3) {
      asm volatile("ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:1024\n ds_read_b128 %2, %12 offset:2048\n ds_read_b128 %3, %12 offset:3072\n"
                   "ds_read_b128 %4, %12 offset:4096\n ds_read_b128 %5, %12 offset:5120\n ds_read_b128 %6, %12 offset:6144\n ds_read_b128 %7, %12 offset:7168\n"
                   "ds_read_b128 %8, %12 offset:8192\n ds_read_b128 %9, %12 offset:9216\n ds_read_b128 %10, %12 offset:10240\n ds_read_b128 %11, %12 offset:11264\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7), "=v"(r8), "=v"(r9), "=v"(r10), "=v"(r11)
                   : "v"(addr) : "memory");
    }
    if (work == 1 || work == 3) {
#pragma unroll
      for (int q = 0; q < 12; ++q) { v0 += v1; v1 += v2; v2 += v3; v3 += v0; }
    }
    if (work == 4) {      // 48 independent VALU (8 chains)
#pragma unroll
      for (int q = 0; q < 6; ++q) { v0 += 1.f; v1 += 1.f; v2 += 1.f; v3 += 1.f; w0 += 1.f; w1 += 1.f; w2 += 1.f; w3 += 1.f; }
    }
    if (work == 5) {      // 48 SALU
#pragma unroll
      for (int q = 0; q < 48; ++q) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sreg));
    }
    if (work == 6 || work == 7) {      // planned LOAD phase: 12 ds_reads + 4 buffer LDS-DMA pieces with scalar offsets, no VALU
      asm volatile("ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:1024\n ds_read_b128 %2, %12 offset:2048\n ds_read_b128 %3, %12 offset:3072\n"
                   "ds_read_b128 %4, %12 offset:4096\n ds_read_b128 %5, %12 offset:5120\n ds_read_b128 %6, %12 offset:6144\n ds_read_b128 %7, %12 offset:7168\n"
                   "ds_read_b128 %8, %12 offset:8192\n ds_read_b128 %9, %12 offset:9216\n ds_read_b128 %10, %12 offset:10240\n ds_read_b128 %11, %12 offset:11264\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7), "=v"(r8), "=v"(r9), "=v"(r10), "=v"(r11)
                   : "v"(addr) : "memory");
      const int np = work == 6 ? 4 : 8;
      for (int j = 0; j < np; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(size_t)(lds_s + 32768 + j * 1024), 16, voff, soff, 0, 0);
        soff = (soff + 1024) & 0xffff;
      }
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (v0 + v1 + v2 + v3 + w0 + w1 + w2 + w3 + r0.x + r11.x + sreg == 12345.f) sink[1] = v0;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
  long long *out; float *sink; unsigned char *gbuf; hipMalloc(&gbuf, 64 << 20); hipMemset(gbuf, 1, 64 << 20);
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 16);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  const char *pn[] = {"partner idle", "partner MFMA acc=VGPR", "partner MFMA acc=AGPR"};
  const char *wn[] = {"12 ds_read_b128 + wait", "48 dependent v_add_f32", "", "12 ds_read_b128 + 48 v_add", "48 independent v_add_f32", "48 s_add_u32", "12 ds_read + 4 DMA (scalar addr)", "12 ds_read + 8 DMA (scalar addr)"};
  for (int partner : {0, 1})
    for (int work : {0, 1, 4, 5, 6, 7}) {
      for (int rep = 0; rep < 2; ++rep) { hipMemset(out, 0, 64); hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, gbuf, partner, work, iters, out, sink); hipDeviceSynchronize(); }
      long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      printf("%-24s %-28s: %7.1f cycles/iteration | partner %5.1f cyc/MFMA\n", pn[partner], wn[work], h[0] / (double)iters, partner ? h[4] / (double)(iters * 48) : 0.0);
    }
  return 0;
}
