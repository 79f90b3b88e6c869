// Per-wave VMEM issue cost vs addressing form and width (follow-up of dma_rate.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void *lds_t;

// form 0: global_load_lds (64-bit vaddr) x16B; 1: buffer_load offen lds x16B; 2: buffer x4B (dword); 3: buffer x16B, only 32 lanes active;
// 4: two half-size pieces (lanes 0-31 each 16B) -- n/a ; 5: buffer x16B regular load (VGPR dest)
template <int FORM>
__global__ __launch_bounds__(512) void k(const unsigned char *base, int pattern, int span, int iters, int active_waves, long long *out, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= active_waves) return;
  const unsigned char *b = base + (long long)blockIdx.x * (1 << 20);
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)b, 0, 1 << 20, 0x00020000);
  float acc = 0.f;
  if (FORM == 3 && lane >= 32) return;
  long long t0 = __builtin_readcyclecounter();
  int pos = wave * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece = pos + j;
      int off;
      if (pattern == 0) off = (piece * 1024 + lane * 16) % span;                                   // contiguous 1 KB
      else off = ((piece * 8 + (lane >> 3)) * 1024 + (lane & 7) * 16) % span;                      // 8 lines @ 1 KB stride
      lds_t dst = (lds_t)(smem + (wave * 8 + j) * 1024);
      if (FORM == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + off), dst, 16, 0, 0);
      else if (FORM == 1 || FORM == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, off, 0, 0, 0);
      else if (FORM == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, off, 0, 0, 0);
      else { float4 v = *reinterpret_cast<const float4 *>(b + off); acc += v.x + v.y + v.z + v.w; }
    }
    pos += active_waves * 8;
    if (FORM != 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  if (acc == 12345.f) sink[0] = acc;
}

template <int FORM>
void run(const char *name, unsigned char *buf, long long *out, float *sink) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 200;
  for (int pattern : {0, 1})
    for (int span : {32 << 10, 256 << 10})
      for (int waves : {1, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(512), 65536, 0, buf, pattern, span, iters, waves, out, sink);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8 * waves;
        printf("%-34s %s span %3d KB waves %d: %6.1f cyc/instr/CU, per wave %6.1f, wall %6.1f us\n", name, pattern ? "8 lines@1KB" : "contig 1KB ", span >> 10, waves,
               h[0] / n, h[0] / n * waves, ms * 1e3);
      }
}

int main() {
  unsigned char *buf; long long *out; float *sink;
  hipMalloc(&buf, 1ull << 30); hipMemset(buf, 1, 1ull << 30); hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4);
  run<0>("global_load_lds x16 (64-bit vaddr)", buf, out, sink);
  run<1>("buffer_load lds x16 offen", buf, out, sink);
  run<2>("buffer_load lds x4 offen", buf, out, sink);
  run<3>("buffer_load lds x16, 32 lanes", buf, out, sink);
  run<5>("global_load_dwordx4 -> VGPR", buf, out, sink);
  return 0;
}
