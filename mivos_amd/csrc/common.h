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

// tuning / A-B only (MIVOS_XCD_CONTIG=0): the tile- and element-walking kernels (stem, FusionNet, upsample, maxpool) deal their work
// round-robin over the workgroups like rounds 1-3 instead of in XCD-contiguous runs
inline int xcd_contig() {
  static const int v = getenv("MIVOS_XCD_CONTIG") ? atoi(getenv("MIVOS_XCD_CONTIG")) : 1;
  return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32x32 MFMA C/D layout (cdna guide §3): lane l, register r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace mivos
