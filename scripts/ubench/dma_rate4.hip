// L2 channel camping test: every workgroup streams 512 distinct 128-B lines (64 KB: misses the 32 KB L1, 2 MB per XCD: L2
// resident) laid out (a) contiguously, (b) one line per KB (NHWC pixel stride of a 256-channel tensor, same 128-B column in
// every workgroup), (c) as (b) but every workgroup uses its own 128-B column (slab rotation).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void *lds_t;

__global__ __launch_bounds__(512) void k(const unsigned char *base, int pattern, int iters, long long *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char *priv = base + (long long)blockIdx.x * (1 << 20);
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)priv, 0, 1 << 20, 0x00020000);
  long long t0 = __builtin_readcyclecounter();
  int pos = wave * 8;
  const int col = pattern == 2 ? (blockIdx.x & 7) * 128 : 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int line = ((pos + j) * 8 + (lane >> 3)) & 511;
      const int off = (pattern == 0 ? line * 128 : line * 1024 + col) + (lane & 7) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_t)(smem + (wave * 8 + j) * 1024), 16, off, 0, 0, 0);
    }
    pos += 64;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned char *buf; long long *out;
  hipMalloc(&buf, 256ull << 20); hipMemset(buf, 1, 256ull << 20); hipMalloc(&out, 64 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 400;
  const char *pn[] = {"contiguous 64 KB", "1 line per KB, same column", "1 line per KB, column = WG % 8"};
  for (int pattern : {0, 1, 2}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, buf, pattern, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64;
    printf("%-32s: %6.1f cyc/KB/CU = %5.1f B/clk/CU, wall %7.1f us = %5.2f TB/s chip\n", pn[pattern], h[0] / n, 1024 * n / h[0], ms * 1e3, 256 * n * 1024 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
