// Faithful skeleton of the ping-pong GEMM loop (conv_f16x3_dma.hip): two groups of 4 waves alternate LOAD phases (NR
// ds_read_b128 into fragment registers + ND LDS-DMA pieces) and MFMA phases (NM MFMAs ON THE LOADED FRAGMENTS), one
// barrier per phase, group 1 one phase behind.  Reports cycles per (LOAD + MFMA) pair to compare with max(L, M) and L + M.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_t;
typedef const __attribute__((address_space(3))) h8 *lds_h8_t;

// mode 0: full ping-pong; 1: no DMA; 2: no ds_reads (fragments constant); 3: MFMA + barriers only; 4: both groups in phase (no ping-pong)
template <int NM>   // MFMAs per phase: 12 or 24 (NM/3 accumulators... uses 4 A/B fragment pairs)
__global__ __launch_bounds__(512) void k(const unsigned char *gbuf, int mode, int nd, int iters, long long *out, float *sink, int pat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<float *>(smem)[i] = 0.001f * (i & 63);
  __syncthreads();
  const unsigned lbase = (unsigned)(size_t)(lds_t)smem;
  const unsigned ra = lbase + (wave >> 2) * 8192 + (lane & 31) * 128 + ((lane >> 5) ^ ((lane >> 1) & 7)) * 16;
  const long long wg_bytes = pat == 2 ? (4ll << 20) : 262144;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(gbuf + (pat == 3 ? 0ll : (long long)blockIdx.x * wg_bytes)), 0, pat == 3 ? (4 << 20) : (int)wg_bytes, 0x00020000);
  const int wrap = pat == 2 || pat == 3 ? (pat == 3 ? 0x1fffff : 0x3fffff) : 0x3ffff;
  const int lds_s = __builtin_amdgcn_readfirstlane((int)lbase + 65536 + wave * 8192);
  const unsigned voff = pat == 1 ? (lane >> 3) * 1024 + (lane & 7) * 16 : lane * 16;
  int soff = __builtin_amdgcn_readfirstlane(wave * 8192);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  h8 fa[4], fb[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { fa[i][e] = (_Float16)(0.01f * (e + i)); fb[i][e] = (_Float16)(0.02f * e); }
  auto barrier = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
  auto load_phase = [&]() {
    if (mode != 2 && mode != 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[i] = *(lds_h8_t)(size_t)(ra + i * 4096); fb[i] = *(lds_h8_t)(size_t)(ra + 32768 + i * 4096); }
      if (NM == 24) {   // 16 reads per phase: model the merged / 256x256 fragment count with 8 extra reads
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i] += *(lds_h8_t)(size_t)(ra + 16384 + i * 4096); }
      }
    }
    if (mode != 1 && mode != 3)
      for (int j = 0; j < nd; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_t)(size_t)(lds_s + j * 1024), 16, voff, soff, 0, 0);
        soff = (soff + (pat == 1 ? 8192 + 128 : 1024)) & wrap;
      }
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  };
  auto mfma_phase = [&]() {
#pragma unroll
    for (int q = 0; q < NM; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[q & 3], fb[(q >> 1) & 3], acc[q & 7], 0, 0, 0);
  };
  barrier();
  if (mode != 4 && wave >= 4) barrier();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    load_phase();
    barrier();
    mfma_phase();
    barrier();
  }
  long long t1 = __builtin_readcyclecounter();
  if (mode != 4 && wave < 4) barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.f) sink[0] = s;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned char *buf; long long *out; float *sink;
  hipMalloc(&buf, 1ull << 30); hipMemset(buf, 1, 1ull << 30); hipMalloc(&out, 64); hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 2000;
  const char *mn[] = {"full ping-pong", "no DMA", "no ds_reads", "MFMA + barriers only", "both groups in phase"};
  const char *pn[] = {"contiguous, 256 KB/WG", "8 lines @1KB stride, 256 KB/WG", "contiguous, 4 MB/WG (beyond L2)", "one 2 MB region for all WGs"};
  for (int nm : {24})
    for (int nd : {4})
      for (int pat : {0, 1, 2, 3})
      for (int mode : {0}) {
        for (int rep = 0; rep < 2; ++rep) {
          if (nm == 12) hipLaunchKernelGGL(k<12>, dim3(256), dim3(512), 160 * 1024, 0, buf, mode, nd, iters, out, sink, pat);
          else hipLaunchKernelGGL(k<24>, dim3(256), dim3(512), 160 * 1024, 0, buf, mode, nd, iters, out, sink, pat);
          hipDeviceSynchronize();
        }
        long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("[%s] NM %2d (%4d MFMA cycles)  ND %d  %-22s: %7.1f cycles per LOAD+MFMA pair (wave 0), %7.1f (wave 4)\n", pn[pat], nm, nm * 32, nd, mn[mode], h[0] / (double)iters, h[4] / (double)iters);
      }
  return 0;
}
