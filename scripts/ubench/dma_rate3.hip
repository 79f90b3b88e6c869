// L2 -> LDS bandwidth of LDS-DMA for the two streams of the f16x3 GEMM: a private L2-resident region per
// workgroup (activations) and one region streamed by every workgroup at once (weights).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void *lds_t;

// pattern 0: private `span` bytes per WG (wrap); 1: all WGs walk the same `span` bytes; 2: alternate (even pieces private, odd shared)
__global__ __launch_bounds__(512) void k(const unsigned char *base, int pattern, int span, int iters, int active_waves, long long *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= active_waves) return;
  const unsigned char *priv = base + (64 << 20) + (long long)blockIdx.x * (1 << 20);
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)priv, 0, 1 << 20, 0x00020000);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 64 << 20, 0x00020000);
  const int mask = span - 1;
  long long t0 = __builtin_readcyclecounter();
  int pos = wave * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int off = ((pos + j) * 1024 + lane * 16) & mask;
      lds_t dst = (lds_t)(smem + (wave * 8 + j) * 1024);
      const bool shared = pattern == 1 || (pattern == 2 && (j & 1));
      if (shared) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, dst, 16, off, 0, 0, 0);
    }
    pos += active_waves * 8;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned char *buf; long long *out;
  hipMalloc(&buf, (64ull << 20) + (256ull << 20)); hipMemset(buf, 1, (64ull << 20) + (256ull << 20)); hipMalloc(&out, 64 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 400;
  const char *pn[] = {"private region per WG", "one region, all WGs", "alternating private/shared"};
  for (int pattern : {0, 1, 2})
    for (int span : {16 << 10, 64 << 10, 128 << 10, 512 << 10, 4 << 20})
      for (int waves : {4, 8}) {
        if (pattern == 0 && span > (1 << 20)) continue;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, buf, pattern, span, iters, waves, out);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8 * waves;
        printf("%-28s span %5d KB waves %d: %6.1f cyc/KB/CU = %5.1f B/clk/CU, wall %7.1f us = %5.2f TB/s chip\n", pn[pattern], span >> 10, waves, h[0] / n, 1024 * n / h[0],
               ms * 1e3, 256 * n * 1024 / (ms * 1e-3) / 1e12);
      }
  return 0;
}
