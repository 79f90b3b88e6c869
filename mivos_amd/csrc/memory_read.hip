// Space-time memory read for gfx950: fp32-MFMA affinity tiles + streaming per-query top-k in LDS +
// softmax over the k survivors + sparse value readout.  The [T*H*W x H*W] affinity of the reference
// (prop_net.py:85-88, 52 MB/object at 480p T=5, 160 GB at 1080p T=200) is never materialised.
//
// Kernel 1 (memread_select): workgroup = 4 waves = 128 queries x one chunk of memory positions.
//   * each wave keeps its 32 query keys (pre-scaled by 1/sqrt(128)) in 64 VGPRs as the MFMA B operand;
//   * memory-key tiles (32 positions x 128 ch, 16 KB) are staged through LDS once per workgroup (shared
//     by the 4 waves; row pitch 132 floats => the ds_read_b128 fragment reads are bank-conflict-free)
//     with the next tile's global loads in flight while the current one is multiplied;
//   * 64 x v_mfma_f32_32x32x2_f32 give a 32x32 score tile; lane (j, h) owns 16 scores of query j;
//   * scores above the query's running threshold tau are appended to a per-query candidate buffer that
//     lives in GLOBAL memory (L2-resident workspace; counters and thresholds stay in LDS): with the buffers
//     in LDS (133 KB) only one workgroup fit per CU and every barrier, LDS round trip and compaction stalled
//     the only wave of its SIMD (affinity MFMA 15 % busy); now LDS is 18 KB, 3 workgroups share a CU and
//     their MFMA / selection phases overlap
//     (packed 64-bit {orderable score, ~index}; one LDS atomic per lane and half-tile); when a buffer may
//     overflow the owning wave compacts it to the exact top-k with a wave-level radix select (ballot
//     bisection of the k-th score) and raises tau (no workgroup barrier).
//   The k best of every (object, chunk, query) go to the workspace.
// Kernel 2 (memread_finalize): one wave per (object, query): merge the per-chunk lists to the exact
//   top-k, exp(s - s_max)/sum in the reference's order, then gather the k value rows (2 KB each) in
//   ascending memory index — the order in which the reference's dense bmm meets its non-zeros.
// Ties: the lower memory index wins (torch.topk leaves ties unspecified).
#include <stdlib.h>

#include "common.h"

namespace mivos {

constexpr int CK = 128, CV = 512;
constexpr int QT = 128;     // queries per workgroup
constexpr int KT = 32;      // memory positions per tile
constexpr int KLD = 132;    // LDS pitch of a key row (floats)
constexpr int CAP = 256;    // candidate slots per query (global memory)
constexpr int CAP_TRIGGER = CAP - 16;   // a half-tile adds at most 16 candidates per query
constexpr int MAX_SPLIT = 8;
constexpr int MAX_TOPK = 64;

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ uint64_t pack_cand(float s, uint32_t idx) {
  return ((uint64_t)f2ord(s) << 32) | (uint64_t)(0xffffffffu - idx);
}
__device__ __forceinline__ float cand_score(uint64_t c) { return ord2f((uint32_t)(c >> 32)); }
__device__ __forceinline__ uint32_t cand_index(uint64_t c) { return 0xffffffffu - (uint32_t)c; }

// Exact top-k of one query's candidate buffer, executed by the owning wave (all 64 lanes; lane l owns
// entries l and l+64).  Wave-level radix select: the k-th largest score is found by bisecting its 32
// orderable bits MSB-first with two ballots per bit, survivors are compacted with ballot prefix sums
// (order inside the buffer is irrelevant, the finalize kernel ranks).  Exact score ties at the threshold
// are resolved toward the lower memory index.  Only the owning wave touches a query's buffer and the LDS
// ops of one wave execute in order, so no barrier is needed — the asm statements only stop the compiler
// from caching LDS values across the wave-level hand-offs.
__device__ __forceinline__ uint64_t ld_cand(const uint64_t *p) {
  // candidates were appended with plain stores by lanes of THIS wave: read them back through L2 (sc1),
  // never from a possibly stale L1 line
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// EPL entries per lane (CAP = 64 * EPL).  Returns the survivors in e[] with keep flags; when WRITE_BACK they
// are also compacted in place.
constexpr int EPL = CAP / 64;
template <bool WRITE_BACK>
__device__ __forceinline__ void compact_query(uint64_t *buf, int *cnt, float *tau, int k, int lane, uint64_t (&e)[EPL],
                                              bool (&keep)[EPL]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's appends have reached L2
  const int n = *cnt;
  uint32_t a[EPL];
#pragma unroll
  for (int t = 0; t < EPL; ++t) {
    keep[t] = lane + 64 * t < n;
    e[t] = keep[t] ? ld_cand(buf + lane + 64 * t) : 0ull;
    a[t] = (uint32_t)(e[t] >> 32);                        // 0 = empty (below every valid score)
  }
  if (n <= k) return;                                     // nothing to drop, threshold unchanged
  // k-th largest orderable score = the largest v with #(a >= v) >= k: bisect its 32 bits MSB-first, ONE vector compare
  // per entry and bit (the counts come from scalar popcounts of the ballots: this code runs beside other waves' MFMA
  // streams, where every vector-ALU instruction costs a whole MFMA slot)
  uint32_t prefix = 0;
#pragma unroll 1
  for (int b = 31; b >= 0; --b) {
    const uint32_t trial = prefix | (1u << b);
    int c = 0;
#pragma unroll
    for (int t = 0; t < EPL; ++t) c += __popcll(__ballot(a[t] >= trial));
    if (c >= k) prefix = trial;
  }
  // prefix = k-th largest score; rem = how many entries equal to it must be kept
  bool eq[EPL];
  int neq = 0, ngt = 0;
#pragma unroll
  for (int t = 0; t < EPL; ++t) {
    eq[t] = a[t] == prefix;
    neq += __popcll(__ballot(eq[t]));
    ngt += __popcll(__ballot(a[t] > prefix));
  }
  const int rem = k - ngt;
  if (neq != rem) {                                        // rare: exact ties straddle the cut -> lowest indices win
    int r[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) r[t] = 0;
    for (int i = 0; i < n; ++i) {
      const uint64_t c = ld_cand(buf + i);
      const bool ceq = (uint32_t)(c >> 32) == prefix;
#pragma unroll
      for (int t = 0; t < EPL; ++t) r[t] += ceq && (uint32_t)c > (uint32_t)e[t];   // ~index: larger = lower index
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) eq[t] = eq[t] && r[t] < rem;
  }
  int base = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int t = 0; t < EPL; ++t) {
    keep[t] = a[t] > prefix || eq[t];
    if (WRITE_BACK) {
      const unsigned long long m = __ballot(keep[t]);
      if (keep[t]) buf[base + __popcll(m & below)] = e[t];
      base += __popcll(m);
    }
  }
  asm volatile("" ::: "memory");
  if (lane == 0) { *cnt = k; *tau = ord2f(prefix); }
  asm volatile("" ::: "memory");
}

template <int ABL>   // ablation switch for profiling builds (0 = product)
__global__ __launch_bounds__(256) void memread_select_kernel(const float *__restrict__ keys, long long keys_ostride,
                                                            const float *__restrict__ qk, uint64_t *__restrict__ cand_out,
                                                            uint64_t *__restrict__ cand_ws, long long n_mem, int n_q,
                                                            int top_k, long long chunk, int n_split) {
  __shared__ __attribute__((aligned(16))) float ktile[KT * KLD];
  __shared__ int cnt[QT];
  __shared__ float tau[QT];
  // this workgroup's candidate buffers in the global workspace: QT x CAP entries
  uint64_t *cand = cand_ws + (((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (QT * CAP);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int obj = blockIdx.z, split = blockIdx.y;
  const int qslot = wave * 32 + j;
  const int q = blockIdx.x * QT + qslot;
  const long long c0 = (long long)split * chunk;
  const long long c1 = (c0 + chunk < n_mem) ? c0 + chunk : n_mem;
  const float *kbase = keys + (long long)obj * keys_ostride;

  if (tid < QT) { cnt[tid] = 0; tau[tid] = -INFINITY; }

  // B operand: this lane's query row, k = 8u + 4h + s, scaled like prop_net.py:86 (qk / sqrt(CK))
  f32x4 qreg[16];
  {
    const float *qrow = qk + (long long)(q < n_q ? q : n_q - 1) * CK + 4 * h;
    const float d = sqrtf((float)CK);  // torch divides by float(math.sqrt(CK)), correctly rounded
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + 8 * u);
      v.x /= d; v.y /= d; v.z /= d; v.w /= d;
      qreg[u] = v;
    }
  }

  // key-tile loader: thread -> row tid>>3, float4 columns (tid&7) + 8 jj
  const int lrow = tid >> 3, lc = tid & 7;
  f32x4 kr[4];
  auto gload = [&](long long kb) {
    const long long m = kb + lrow;
    const bool ok = m < c1;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(kbase + (ok ? m : c0) * CK) + lc;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      f32x4 v = src[8 * jj];
      if (!ok) { v.x = v.y = v.z = v.w = 0.f; }
      kr[jj] = v;
    }
  };
  gload(c0);
  float my_tau = -INFINITY;
  for (long long kb = c0; kb < c1; kb += KT) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4 *>(ktile + lrow * KLD + 4 * (lc + 8 * jj)) = kr[jj];
    __syncthreads();
    if (kb + KT < c1) gload(kb + KT);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *arow = ktile + j * KLD + 4 * h;
    // all 16 fragment reads in flight first (one LDS latency per tile instead of 16 exposed ones: with
    // 1-2 waves per SIMD nothing else hides them), then 64 back-to-back MFMAs
    f32x4 afrag[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) afrag[u] = *reinterpret_cast<const f32x4 *>(arow + 8 * u);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[u][s], qreg[u][s], acc, 0, 0, 0);
    }

    if (ABL == 1 && kb >= c0 + 2 * KT) {   // MFMA + staging only (first two tiles select normally so the lists are valid)
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[r];
      if (t == 123.456f) my_tau = t;
      continue;
    }
    // append in two halves of 8 registers; before each half make room: a half adds <= 16 per query
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      asm volatile("" ::: "memory");
      const int mycnt = cnt[qslot];
      unsigned long long need = __ballot(mycnt > CAP_TRIGGER) & 0xffffffffull;
      if (need) {
        while (need) {
          const int s = wave * 32 + __builtin_ctzll(need);
          need &= need - 1;
          uint64_t e[EPL];
          bool kp[EPL];
          compact_query<true>(cand + s * CAP, cnt + s, tau + s, top_k, lane, e, kp);
        }
        my_tau = tau[qslot];
      }
      // one LDS atomic per lane and half-tile: reserve as many slots as this lane has passing scores.
      // (selection VALU is the bound here - one VALU issue per MFMA of the co-resident waves - so: 32-bit row
      // arithmetic, and the row-in-range test only on the last, partial tile of a chunk)
      uint32_t passmask = 0;
      const int rows_left = (int)(c1 - kb);                  // >= KT on every tile but the last
      if (rows_left >= KT) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) passmask |= (uint32_t)(acc[half * 8 + rr] > my_tau) << rr;
      } else {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int r = half * 8 + rr;
          passmask |= (uint32_t)(mfma32_row(r, lane) < rows_left && acc[r] > my_tau) << rr;
        }
      }
      if (passmask) {
        const uint32_t kb32 = (uint32_t)kb;
        int pos = atomicAdd(cnt + qslot, __popc(passmask));
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          if (passmask & (1u << rr)) {
            const int r = half * 8 + rr;
            cand[qslot * CAP + pos++] = pack_cand(acc[r], kb32 + (uint32_t)mfma32_row(r, lane));
          }
        }
      }
    }
  }
  // final exact top-k of this chunk for the wave's 32 queries, written (unsorted) to the chunk list
  for (int qs = 0; qs < 32; ++qs) {
    const int s = wave * 32 + qs;
    uint64_t e[EPL];
    bool kp[EPL];
    compact_query<false>(cand + s * CAP, cnt + s, tau + s, top_k, lane, e, kp);
    const int qq = blockIdx.x * QT + s;
    if (qq < n_q) {
      uint64_t *dst = cand_out + (((long long)obj * n_split + split) * n_q + qq) * top_k;
      const unsigned long long below = (1ull << lane) - 1ull;
      int kept = 0;
#pragma unroll
      for (int t = 0; t < EPL; ++t) {
        const unsigned long long m = __ballot(kp[t]);
        if (kp[t]) dst[kept + __popcll(m & below)] = e[t];
        kept += __popcll(m);
      }
      if (lane >= kept && lane < top_k) dst[lane] = 0ull;            // chunk had fewer than k positions
    }
  }
}

// one single-wave workgroup per (object, query); __syncthreads() on a 64-thread block is just the
// LDS ordering fence between the phases
template <bool INDICES>
__global__ __launch_bounds__(64) void memread_finalize_kernel(const uint64_t *__restrict__ cand_in,
                                                             const float *__restrict__ values, long long values_ostride,
                                                             float *__restrict__ out, long long out_ostride,
                                                             long long out_pstride, int32_t *__restrict__ idx_out,
                                                             float *__restrict__ w_out, int n_q, int top_k, int n_split) {
  __shared__ uint64_t all[MAX_SPLIT * MAX_TOPK];
  __shared__ uint64_t sel[MAX_TOPK];
  __shared__ float wv[MAX_TOPK];
  __shared__ uint32_t oi[MAX_TOPK];
  __shared__ float ow[MAX_TOPK];
  const int lane = threadIdx.x;
  const int q = blockIdx.x, obj = blockIdx.y;
  const int n = n_split * top_k;
  uint64_t mine[MAX_SPLIT];
  int rank[MAX_SPLIT];
#pragma unroll
  for (int e = 0; e < MAX_SPLIT; ++e) {
    const int i = lane + 64 * e;
    uint64_t v = 0ull;
    if (i < n) {
      const int sp = i / top_k, t = i - sp * top_k;
      v = cand_in[(((long long)obj * n_split + sp) * n_q + q) * top_k + t];
      all[i] = v;
    }
    mine[e] = v;
    rank[e] = 0;
  }
  __syncthreads();
  // exact merge of the per-chunk survivor lists (unsorted): all-pairs rank, best first
  for (int i = 0; i < n; ++i) {
    const uint64_t c = all[i];
#pragma unroll
    for (int e = 0; e < MAX_SPLIT; ++e) rank[e] += (c > mine[e]) ? 1 : 0;
  }
#pragma unroll
  for (int e = 0; e < MAX_SPLIT; ++e)
    if (lane + 64 * e < n && mine[e] != 0ull && rank[e] < top_k) sel[rank[e]] = mine[e];
  __syncthreads();
  // softmax over the k survivors, max = best score (prop_net.py:55), sum in rank order
  const uint64_t c = lane < top_k ? sel[lane] : 0ull;
  const float smax = cand_score(sel[0]);
  const float e = lane < top_k ? expf(cand_score(c) - smax) : 0.f;
  if (lane < top_k) wv[lane] = e;
  __syncthreads();
  float sum = 0.f;
  for (int i = 0; i < top_k; ++i) sum += wv[i];
  const float w = e / sum;
  const uint32_t idx = cand_index(c);
  if (INDICES) {
    if (lane < top_k) {
      idx_out[((long long)obj * n_q + q) * top_k + lane] = (int32_t)idx;
      w_out[((long long)obj * n_q + q) * top_k + lane] = w;
    }
    return;
  }
  // order by memory index (ascending) for the readout
  int r2 = 0;
  for (int i = 0; i < top_k; ++i) r2 += (cand_index(sel[i]) < idx) ? 1 : 0;
  if (lane < top_k) { oi[r2] = idx; ow[r2] = w; }
  __syncthreads();
  const float *vb = values + (long long)obj * values_ostride + 4 * lane;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < top_k; ++t) {
    const float wt = ow[t];
    const f32x4 *row = reinterpret_cast<const f32x4 *>(vb + (long long)oi[t] * CV);
    const f32x4 v0 = row[0], v1 = row[64];
    a0.x = fmaf(wt, v0.x, a0.x); a0.y = fmaf(wt, v0.y, a0.y); a0.z = fmaf(wt, v0.z, a0.z); a0.w = fmaf(wt, v0.w, a0.w);
    a1.x = fmaf(wt, v1.x, a1.x); a1.y = fmaf(wt, v1.y, a1.y); a1.z = fmaf(wt, v1.z, a1.z); a1.w = fmaf(wt, v1.w, a1.w);
  }
  float *o = out + (long long)obj * out_ostride + (long long)q * out_pstride + 4 * lane;
  *reinterpret_cast<f32x4 *>(o) = a0;
  *reinterpret_cast<f32x4 *>(o + 256) = a1;
}

struct SplitPlan { int n_split; long long chunk; };
static SplitPlan plan_split(int n_obj, long long n_mem, int n_q) {
  const long long q_tiles = cdiv(n_q, QT), tiles = cdiv(n_mem, KT);
  // Selection work grows with the number of chunks (every chunk keeps its own top-k: k(1 + ln(n/k)) appends),
  // MFMA work does not.  Short memories (480p, T <= ~30): one round of <= 512 workgroups
  // (two per CU: measured 1.2x over one per CU at K=5, T=12).  Long memories
  // (>= 512 tiles per chunk, where selection is negligible): up to 3 workgroups per CU so that barriers and
  // compactions of one workgroup hide behind the MFMAs of the others.
  const long long wg = q_tiles * n_obj;
  static const int target = getenv("MIVOS_MEMREAD_WGS") ? atoi(getenv("MIVOS_MEMREAD_WGS")) : 512;   // tuning only (two 4-wave workgroups per CU)
  long long s = target / wg;
  if (s < 1) s = 1;
  const long long s3 = cdiv(768, wg), by_len = tiles / 512;
  if (by_len > s) s = by_len < s3 ? by_len : s3;
  if (s > MAX_SPLIT) s = MAX_SPLIT;
  if (s > tiles) s = tiles;
  if (s < 1) s = 1;
  SplitPlan p;
  p.chunk = (long long)cdiv(tiles, s) * KT;
  p.n_split = cdiv(n_mem, p.chunk);
  return p;
}

static int run_select(const float *keys, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem, int n_q,
                      int top_k, void *workspace, int64_t workspace_bytes, hipStream_t st, SplitPlan &pl) {
  if (!keys || !qk || !workspace) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: null pointer");
  if (top_k < 1 || top_k > MAX_TOPK) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: top_k=%d unsupported (1..%d)", top_k, MAX_TOPK);
  if (n_mem < top_k) return fail(MIVOS_ERR_TOPK_RANGE, "selected index k out of range (top_k=%d > %lld memory positions)", top_k, (long long)n_mem);
  if (n_obj < 1 || n_q < 1 || n_mem >= 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: bad sizes");
  if (((uintptr_t)keys & 15) || ((uintptr_t)qk & 15) || (keys_ostride & 3)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: keys/qk must be 16-byte aligned");
  if (workspace_bytes < mivos_memory_read_workspace_bytes(n_obj, n_mem, n_q, top_k)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: workspace too small");
  pl = plan_split(n_obj, n_mem, n_q);
  uint64_t *lists = (uint64_t *)workspace;                                   // [obj][split][q][top_k] chunk survivors
  uint64_t *bufs = lists + (long long)n_obj * MAX_SPLIT * n_q * top_k;       // [obj][split][q_tile][QT][CAP] candidates
  static const int abl = getenv("MIVOS_ABL") ? atoi(getenv("MIVOS_ABL")) : 0;   // profiling only
  if (abl == 1)
    hipLaunchKernelGGL(memread_select_kernel<1>, dim3(cdiv(n_q, QT), pl.n_split, n_obj), dim3(256), 0, st, keys,
                       (long long)keys_ostride, qk, lists, bufs, (long long)n_mem, n_q, top_k, pl.chunk, pl.n_split);
  else
    hipLaunchKernelGGL(memread_select_kernel<0>, dim3(cdiv(n_q, QT), pl.n_split, n_obj), dim3(256), 0, st, keys,
                       (long long)keys_ostride, qk, lists, bufs, (long long)n_mem, n_q, top_k, pl.chunk, pl.n_split);
  return check_launch("memread_select");
}

}  // namespace mivos

using namespace mivos;

extern "C" int64_t mivos_memory_read_workspace_bytes(int n_obj, int64_t n_mem, int n_q, int top_k) {
  (void)n_mem;
  return (int64_t)n_obj * MAX_SPLIT * n_q * top_k * 8 + (int64_t)n_obj * MAX_SPLIT * cdiv(n_q, QT) * QT * CAP * 8;
}

extern "C" int mivos_memory_read_topk(const float *keys, int64_t keys_ostride, const float *values, int64_t values_ostride,
                                      const float *qk, float *out, int64_t out_ostride, int64_t out_pstride, int n_obj,
                                      int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes,
                                      void *stream) {
  if (!values || !out || ((uintptr_t)values & 15) || ((uintptr_t)out & 15) || (out_pstride & 3) || (out_ostride & 3) || (values_ostride & 3))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: values/out must be 16-byte aligned");
  SplitPlan pl;
  int rc = run_select(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, (hipStream_t)stream, pl);
  if (rc) return rc;
  hipLaunchKernelGGL(memread_finalize_kernel<false>, dim3(n_q, n_obj), dim3(64), 0, (hipStream_t)stream,
                     (const uint64_t *)workspace, values, (long long)values_ostride, out, (long long)out_ostride,
                     (long long)out_pstride, (int32_t *)nullptr, (float *)nullptr, n_q, top_k, pl.n_split);
  return check_launch("memread_finalize");
}

extern "C" int mivos_memory_read_topk_indices(const float *keys, int64_t keys_ostride, const float *qk, int32_t *idx_out,
                                              float *weight_out, int n_obj, int64_t n_mem, int n_q, int top_k,
                                              void *workspace, int64_t workspace_bytes, void *stream) {
  if (!idx_out || !weight_out) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_indices: null pointer");
  SplitPlan pl;
  int rc = run_select(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, (hipStream_t)stream, pl);
  if (rc) return rc;
  hipLaunchKernelGGL(memread_finalize_kernel<true>, dim3(n_q, n_obj), dim3(64), 0, (hipStream_t)stream,
                     (const uint64_t *)workspace, (const float *)nullptr, 0ll, (float *)nullptr, 0ll, 0ll, idx_out,
                     weight_out, n_q, top_k, pl.n_split);
  return check_launch("memread_finalize_indices");
}
