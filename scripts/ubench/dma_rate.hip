// LDS-DMA (global_load_lds_dwordx4) throughput per CU vs address pattern / waves issuing.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const unsigned char *g, unsigned char *l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// mode 0: piece = 1 KB contiguous; 1: piece = 8 lines at `stride` bytes (lane>>3 selects the line); 2: all lanes one line
// every workgroup walks its own `span` bytes region (wrap) `iters` times, 8 pieces per wave per iteration
// regular loads (use_dma == 0): same addresses with global_load_dwordx4 into registers
template <int USE_DMA>
__global__ __launch_bounds__(512) void k(const unsigned char *base, long long wg_stride, int mode, int stride, int span, int iters,
                                         int active_waves, long long *out, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= active_waves) return;
  const unsigned char *b = base + (long long)blockIdx.x * wg_stride;
  float acc = 0.f;
  long long t0 = __builtin_readcyclecounter();
  int pos = wave * 8;   // piece index
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long long off;
      const int piece = pos + j;
      if (mode == 0) off = ((long long)piece * 1024 + lane * 16) % span;
      else if (mode == 1) off = (((long long)piece * 8 + (lane >> 3)) * stride + (lane & 7) * 16) % span;
      else off = (lane & 7) * 16;
      if (USE_DMA) glds16(b + off, smem + (wave * 8 + j) * 1024);
      else { float4 v = *reinterpret_cast<const float4 *>(b + off); acc += v.x + v.y + v.z + v.w; }
    }
    pos += active_waves * 8;
    if (USE_DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  const size_t bytes = 1ull << 30;
  unsigned char *buf; long long *out; float *sink;
  hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  struct { const char *name; int mode, stride, span; long long wg_stride; } cfg[] = {
      {"contig 1KB pieces, 32 KB span/WG (L1?)", 0, 0, 32 << 10, 1 << 20},
      {"contig 1KB pieces, 1 MB span/WG (L2)", 0, 0, 1 << 20, 1 << 20},
      {"contig, all WGs same 32 KB (weights)", 0, 0, 32 << 10, 0},
      {"8 lines @1KB stride, 1 MB span/WG", 1, 1024, 1 << 20, 1 << 20},
      {"8 lines @1KB stride, 256 KB span/WG", 1, 1024, 256 << 10, 1 << 20},
      {"8 lines @1152B stride, 1 MB span/WG", 1, 1152, 1 << 20, 1 << 20},
      {"8 lines @128B stride(=contig), 256 KB", 1, 128, 256 << 10, 1 << 20},
      {"all lanes one line", 2, 0, 1024, 1 << 20},
  };
  const int iters = 200;
  for (int dma = 1; dma >= 0; --dma)
    for (auto &c : cfg)
      for (int waves : {1, 4, 8}) {
        for (int grid : {1, 256}) {
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (dma) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 65536, 0, buf, c.wg_stride, c.mode, c.stride, c.span, iters, waves, out, sink);
            else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 65536, 0, buf, c.wg_stride, c.mode, c.stride, c.span, iters, waves, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
          }
          float ms; hipEventElapsedTime(&ms, e0, e1);
          long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
          const double pieces = (double)iters * 8 * waves;
          printf("%s %-40s waves %d grid %3d: %7.1f cyc/piece/CU  (%5.1f B/clk/CU, wall %6.1f us, %6.2f TB/s chip)\n", dma ? "DMA " : "LOAD", c.name, waves, grid,
                 h[0] / pieces, 1024.0 * pieces / h[0], ms * 1e3, grid * pieces * 1024 / (ms * 1e-3) / 1e12);
        }
      }
  return 0;
}
