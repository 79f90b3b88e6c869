// Shared host/device helpers for libmivos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/mivos_hip.h"

namespace mivos {

// thread-local last error string (mivos_last_error)
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MIVOS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MIVOS_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device); `mask` holds one bit per
// device (idempotent when two host threads race, so no lock is needed)
inline int ensure_dynamic_lds(const void *kern, size_t lds, std::atomic<uint64_t> &mask, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(MIVOS_ERR_LAUNCH, "%s: hipGetDevice failed", what);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(mask.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(MIVOS_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    mask.fetch_or(bit, std::memory_order_release);
  }
  return MIVOS_OK;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// How the tile- / element-walking kernels deal their work to the workgroups (hardware block b runs on XCD b % 8; the eight L2s are not
// shared): round-robin (rounds 1-3) or in XCD-contiguous runs, so that neighbouring outputs - which share input lines - meet in one L2.
// Measured per kernel (round 4, profiles/r04e_xcd_contig_ab.txt, r04e_config3_pmc_traffic.json; fabric reads per launch / kernel time):
//   upsample2x_add_multi 304 -> 182 MB, 77.9 -> 77.8 us;  maxpool_sh32 214 -> 204 MB, same time;  fusion_head 351 -> 267 MB, 155 -> 140 us   => level 1 (default)
//   stem 108 -> 70 MB, 104 -> 105 us;  fusion_conv1 212 -> 73 MB but 352 -> 373 us;  fusion_resblock 663 -> 604 MB, 355 -> 374 us;
//   memread_finalize 208 -> 201 MB, 81 -> 84 us                                                                                  => level 2 only (A/B)
// MIVOS_XCD_CONTIG = 0 / 1 / 2 (tuning only).
inline int xcd_contig() {
  static const int v = getenv("MIVOS_XCD_CONTIG") ? atoi(getenv("MIVOS_XCD_CONTIG")) : 1;
  return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32x32 MFMA C/D layout (cdna guide §3): lane l, register r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace mivos
