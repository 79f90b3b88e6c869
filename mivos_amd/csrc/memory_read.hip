// Space-time memory read for gfx950: exact-fp32 MFMA affinity tiles + streaming per-query top-k in LDS + softmax over
// the k survivors + sparse value readout.  The [T*H*W x H*W] affinity of the reference (prop_net.py:85-88, 52 MB/object
// at 480p T=5, 160 GB at 1080p T=200) is never materialised.
//
// Kernel 1 (memread_select): PERSISTENT, one 4-wave workgroup per CU (one wave per SIMD, 154 KB of LDS).
//   Work = streams x tiles: a stream is (object, tile of 64 queries) against the whole memory, cut into tiles of 32
//   memory positions.  The streams' tiles are laid end to end and dealt to the workgroups in equal contiguous runs
//   (stream-K style): perfect balance for any object count / frame size / bank depth, and a run that crosses a stream
//   boundary is processed as two segments.  Every segment leaves its own candidate list (>= its exact top-k) in the
//   workspace; kernel 2 merges the 1-3 lists of a query exactly.
//   Per wave: 16 queries (pre-scaled by 1/sqrt(128)) live in 32 VGPRs as the MFMA B operand; key tiles are staged in
//   LDS once per workgroup (double buffered, one barrier per tile, 528-byte row pitch => conflict-free ds_read_b128);
//   2 x 32 v_mfma_f32_16x16x4_f32 (two independent accumulators: rows 0-15 / 16-31 of the tile) give the 32x16 score
//   tile; lane (j, g) owns 8 scores of query j.
//   Selection runs IN THE SHADOW of the next tile's MFMAs, in the same wave: measured in round 1
//   (scripts/ubench/lds_vs_mfma.hip) a co-resident wave gets ~1 vector instruction issued per MFMA of its partner,
//   while a wave's own independent instructions issue freely between its MFMAs (7 slots per 32-cycle MFMA).  So the
//   loop is software pipelined: the 8 score registers of tile t-1 are compared / appended one per group of 8 MFMAs of
//   tile t, the LDS atomic that reserves the slots is issued one group before its result is needed.
//   Candidates: per query 240 packed {orderable score, ~index} entries in LDS.  Scores above the query's threshold tau
//   are appended; when a buffer may overflow the owning wave compacts it: bisection of the packed 64-bit keys (all
//   distinct) with two ballots per bit, stopped as soon as between k and k+16 entries survive (an exact cut is not
//   needed until the end), tau := the cut.  Ties: lower memory index wins (torch.topk leaves ties unspecified).
// Kernel 2 (memread_finalize): one wave per (object, query): exact k-th largest of the merged lists (same bisection,
//   run to the end), exp(s - s_max)/sum in rank order like the reference, then the k value rows (2 KB each) are gathered
//   in ascending memory index - the order in which the reference's dense bmm meets its non-zeros.
#include <stdlib.h>

#include "common.h"

namespace mivos {

constexpr int CK = 128, CV = 512;
constexpr int QW = 16;      // queries per wave (MFMA N)
constexpr int QT = 64;      // queries per workgroup
constexpr int KT = 32;      // memory positions per LDS tile (two 16-row MFMA sub-tiles)
constexpr int KLD = 132;    // LDS pitch of a key row (floats)
constexpr int CAP = 240;    // candidate slots per query (LDS)
constexpr int CAP_TRIGGER = CAP - KT;   // a tile adds at most 32 candidates per query
constexpr int SLACK = 16;   // a compaction leaves between k and k + SLACK survivors
constexpr int MAX_TOPK = 64;
constexpr int MAX_SLOTS = 12;                        // candidate lists (segments) per stream
constexpr int EPL = (CAP + 63) / 64;                 // candidate entries per lane during a compaction
constexpr int FIN_EPL_MAX = (MAX_SLOTS * (MAX_TOPK + SLACK) + 63) / 64;   // entries per lane in the finalize merge

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

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

// Largest 64-bit prefix p with #(e >= p) >= k, found MSB first; stops early once k <= #(e >= p) <= k + slack.
// N entries per lane, empty entries are 0.  Wave-uniform result; `count` = #(e >= p).
template <int N>
__device__ __forceinline__ uint64_t bisect_kth(const uint64_t (&e)[N], int k, int slack, int &count) {
  uint64_t prefix = 0;
  int c_at = 0;
#pragma unroll 1
  for (int b = 63; b >= 0; --b) {
    const uint64_t trial = prefix | (1ull << b);
    int c = 0;
#pragma unroll
    for (int t = 0; t < N; ++t) c += __popcll(__ballot(e[t] >= trial));
    if (c >= k) {
      prefix = trial;
      c_at = c;
      if (c <= k + slack) break;
    }
  }
  count = c_at;
  return prefix;
}

// Compaction of one query's LDS candidate buffer by the owning wave (all 64 lanes).  n > k + SLACK entries -> between k
// and k + SLACK survivors, compacted in place; *cnt and *tau updated.  Only the owning wave touches the buffer and the LDS
// operations of one wave execute in order, so no barrier is needed.
__device__ __forceinline__ void compact_query(uint64_t *buf, int *cnt, float *tau, int k, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(*cnt);
  if (n <= k + SLACK) return;
  uint64_t e[EPL];
#pragma unroll
  for (int t = 0; t < EPL; ++t) e[t] = (lane + 64 * t < n) ? buf[lane + 64 * t] : 0ull;
  int c;
  const uint64_t p = bisect_kth<EPL>(e, k, SLACK, c);
  int base = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int t = 0; t < EPL; ++t) {
    const bool keep = e[t] >= p;
    const unsigned long long m = __ballot(keep);
    if (keep) buf[base + __popcll(m & below)] = e[t];
    base += __popcll(m);
  }
  // later positions have higher indices: one whose score EQUALS the cut's score loses the tie against all c >= k survivors
  if (lane == 0) { *cnt = c; *tau = ord2f((uint32_t)(p >> 32)); }
}

struct SelectArgs {
  const float *keys;
  long long keys_ostride;
  const float *qk;
  uint64_t *lists;        // [stream][slot][QT][L]
  long long n_mem;
  int n_q, top_k, n_qtiles;
  int tps;                // tiles per stream
  long long total_tiles;
  int tiles_per_wg, slots, L;
};

template <int ABL>   // ablation switch for profiling builds (0 = product, 1 = MFMA + staging only)
__global__ __launch_bounds__(256, 1) void memread_select_kernel(const SelectArgs a) {
  __shared__ __attribute__((aligned(16))) float ktile[2][KT * KLD];
  __shared__ uint64_t cand[QT * CAP];
  __shared__ int cnt[QT];
  __shared__ float tau[QT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jq = lane & 15, g = lane >> 4;
  const int qslot = wave * QW + jq;
  // channel of (MFMA step u, k-slot g, element s): 64 (g&1) + 32 (g>>1) + 4u + s.  Any bijection works as long as keys
  // and queries use the same one; this one makes the two k-slots that share a ds_read_b128 lane group (g, g^1) hit the
  // same banks with different rows => conflict-free fragment reads (row pitch 132 floats).
  const int coff = 64 * (g & 1) + 32 * (g >> 1);
  const int lrow = tid >> 3, lc = tid & 7;           // key-tile loader: row tid>>3, float4 columns lc + 8 jj

  long long t_begin = (long long)blockIdx.x * a.tiles_per_wg;
  const long long t_end = (t_begin + a.tiles_per_wg < a.total_tiles) ? t_begin + a.tiles_per_wg : a.total_tiles;

  while (t_begin < t_end) {
    const int stream = (int)(t_begin / a.tps);
    const int seg_lo = (int)(t_begin - (long long)stream * a.tps);
    int seg_hi = seg_lo + (int)(t_end - t_begin);
    if (seg_hi > a.tps) seg_hi = a.tps;
    const int nt = seg_hi - seg_lo;
    const int obj = stream / a.n_qtiles, qtile = stream - obj * a.n_qtiles;
    const int slot = (int)blockIdx.x - (int)(((long long)stream * a.tps) / a.tiles_per_wg);
    // memory positions are 32-bit (n_mem < 2^31 is checked on the host)
    const int r0 = seg_lo * KT;
    const int r1 = ((long long)seg_hi * KT < a.n_mem) ? seg_hi * KT : (int)a.n_mem;
    const float *kbase = a.keys + (long long)obj * a.keys_ostride;

    __syncthreads();                                  // previous segment completely done with LDS
    if (tid < QT) { cnt[tid] = 0; tau[tid] = -INFINITY; }

    // B operand: this lane's query row pieces, scaled like prop_net.py:86 (qk / sqrt(CK), a true division)
    f32x4_t qreg[8];
    {
      const int q = qtile * QT + qslot;
      const float *qrow = a.qk + (long long)(q < a.n_q ? q : a.n_q - 1) * CK + coff;
      const float d = sqrtf((float)CK);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4_t v = *reinterpret_cast<const f32x4_t *>(qrow + 4 * u);
        v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        qreg[u] = v;
      }
    }

    f32x4_t kr[4];
    auto gload = [&](int kb) {
      const int m = kb + lrow;
      const bool ok = m < r1;
      const f32x4_t *src = reinterpret_cast<const f32x4_t *>(kbase + (long long)(ok ? m : r0) * CK) + lc;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4_t v = src[8 * jj];
        if (!ok) { v.x = v.y = v.z = v.w = 0.f; }
        kr[jj] = v;
      }
    };
    auto lds_store = [&](int buf) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4_t *>(&ktile[buf][lrow * KLD + 4 * (lc + 8 * jj)]) = kr[jj];
    };
    gload(r0);
    lds_store(0);
    if (nt > 1) gload(r0 + KT);
    __syncthreads();

    // scores of the previous tile (software pipeline): -inf = nothing to select on the first tile
    f32x4_t p0 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, p1 = p0;
    int pb = r0;                                      // base row of the previous tile
    float my_tau = -INFINITY;

    // one selection step: score register (sub, r) of the previous tile; the slot reservation (LDS atomic) issued in step i
    // is consumed in step i + 1, after the 8 MFMAs in between
    int pend_pos = 0;
    bool pend = false;
    uint64_t pend_ent = 0ull;
    auto flush_pending = [&]() {
      if (pend) cand[qslot * CAP + pend_pos] = pend_ent;
      pend = false;
    };
    auto select_step = [&](int i) {
      flush_pending();
      const float sc = (i < 4) ? p0[i & 3] : p1[i & 3];
      const int rowoff = 16 * (i >> 2) + 4 * g + (i & 3);
      const bool pass = sc > my_tau && (pb + rowoff < r1);
      if (pass) {
        pend_pos = atomicAdd(&cnt[qslot], 1);
        pend_ent = pack_cand(sc, (uint32_t)(pb + rowoff));
      }
      pend = pass;
    };
    auto make_room = [&]() {
      // before the (up to 32 per query) appends of a tile: compact every buffer of this wave that might overflow
      const int mycnt = cnt[qslot];
      unsigned long long need = __ballot(mycnt > CAP_TRIGGER) & 0xffffull;
      if (need) {
        while (need) {
          const int s = wave * QW + __builtin_ctzll(need);
          need &= need - 1;
          compact_query(cand + s * CAP, cnt + s, tau + s, a.top_k, lane);
        }
        my_tau = tau[qslot];
      }
    };

    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      const float *arow0 = &ktile[cur][jq * KLD + coff];
      const float *arow1 = arow0 + 16 * KLD;
      f32x4_t af0[8], af1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        af0[u] = *reinterpret_cast<const f32x4_t *>(arow0 + 4 * u);
        af1[u] = *reinterpret_cast<const f32x4_t *>(arow1 + 4 * u);
      }
      if (t + 1 < nt) lds_store(cur ^ 1);             // tile t+1 (its buffer was last read in iteration t-1)
      if (t + 2 < nt) gload(r0 + (t + 2) * KT);
      if (ABL == 0) make_room();

      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (ABL == 0) select_step(u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af0[u][s], qreg[u][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af1[u][s], qreg[u][s], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ABL == 0) flush_pending();
      if (ABL == 1 && t >= 2) {                        // keep the MFMA results alive without selecting
        const float sum = (acc0.x + acc0.y) + (acc1.z + acc1.w);
        if (sum == 123.456f) my_tau = sum;
      } else {
        p0 = acc0; p1 = acc1;
        pb = r0 + t * KT;
      }
      __syncthreads();
    }
    // drain the pipeline: select on the last tile
    make_room();
#pragma unroll
    for (int i = 0; i < 8; ++i) select_step(i);
    flush_pending();

    // this segment's candidate lists: for each of the wave's 16 queries between min(n, k) and k + SLACK entries
    for (int qs = 0; qs < QW; ++qs) {
      const int s = wave * QW + qs;
      compact_query(cand + s * CAP, cnt + s, tau + s, a.top_k, lane);
      const int n = cnt[s];                           // <= k + SLACK = L
      uint64_t *dst = a.lists + (((long long)stream * a.slots + slot) * QT + s) * a.L;
      for (int i = lane; i < a.L; i += 64) dst[i] = (i < n) ? cand[s * CAP + i] : 0ull;
    }
    t_begin += nt;
  }
}

// one single-wave workgroup per (object, query); __syncthreads() on a 64-thread block is just the LDS ordering fence
// between the phases
template <bool INDICES, int FIN_EPL>   // FIN_EPL: merged candidates per lane (64 FIN_EPL >= segments x L)
__global__ __launch_bounds__(64) void memread_finalize_kernel(const uint64_t *__restrict__ lists,
                                                             const float *__restrict__ values, long long values_ostride,
                                                             float *__restrict__ out, long long out_ostride,
                                                             long long out_pstride, int32_t *__restrict__ idx_out,
                                                             float *__restrict__ w_out, int n_q, int top_k, int n_qtiles,
                                                             int tps, int tiles_per_wg, int slots, int L) {
  __shared__ uint64_t sel[MAX_TOPK];
  __shared__ float wv[MAX_TOPK];
  __shared__ uint32_t oi[MAX_TOPK];
  __shared__ float ow[MAX_TOPK];
  const int lane = threadIdx.x;
  const int q = blockIdx.x, obj = blockIdx.y;
  const int stream = obj * n_qtiles + q / QT, qs = q % QT;
  // the segments of this stream: workgroups w_first .. w_last of the select kernel
  const int w_first = (int)(((long long)stream * tps) / tiles_per_wg);
  const int w_last = (int)((((long long)stream + 1) * tps - 1) / tiles_per_wg);
  const int n = (w_last - w_first + 1) * L;
  uint64_t e[FIN_EPL];
#pragma unroll
  for (int t = 0; t < FIN_EPL; ++t) {
    const int i = lane + 64 * t;
    uint64_t v = 0ull;
    if (i < n) {
      const int sl = i / L, k = i - sl * L;
      v = lists[(((long long)stream * slots + sl) * QT + qs) * L + k];
    }
    e[t] = v;
  }
  // exact k-th largest of the merged candidates (keys are distinct: slack 0 ends with exactly k survivors)
  int c;
  const uint64_t p = bisect_kth<FIN_EPL>(e, top_k, 0, c);
  {
    int base = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int t = 0; t < FIN_EPL; ++t) {
      const bool keep = e[t] >= p && e[t] != 0ull;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = base + __popcll(m & below);
        if (pos < MAX_TOPK) sel[pos] = e[t];
      }
      base += __popcll(m);
    }
  }
  __syncthreads();
  // rank the k survivors, best first
  const uint64_t mine = lane < top_k ? sel[lane] : 0ull;
  int rank = 0;
  for (int i = 0; i < top_k; ++i) rank += (sel[i] > mine) ? 1 : 0;
  __syncthreads();
  if (lane < top_k) sel[rank] = mine;
  __syncthreads();
  // softmax over the k survivors, max = best score (prop_net.py:55), sum in rank order
  const uint64_t cc = lane < top_k ? sel[lane] : 0ull;
  const float smax = cand_score(sel[0]);
  const float ex = lane < top_k ? expf(cand_score(cc) - smax) : 0.f;
  if (lane < top_k) wv[lane] = ex;
  __syncthreads();
  float sum = 0.f;
  for (int i = 0; i < top_k; ++i) sum += wv[i];
  const float w = ex / sum;
  const uint32_t idx = cand_index(cc);
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

// ---- host side --------------------------------------------------------------------------------------------------
struct Plan {
  int n_qtiles, streams, tps, n_wg, tiles_per_wg, slots, L;
  long long total;
};

static int compute_units() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int &c = cus[dev & 63];
  if (c == 0) {
    int v = 0;
    c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return c;
}

static Plan make_plan(int n_obj, long long n_mem, int n_q, int top_k) {
  Plan p;
  p.n_qtiles = cdiv(n_q, QT);
  p.streams = n_obj * p.n_qtiles;
  p.tps = cdiv(n_mem, KT);
  p.total = (long long)p.streams * p.tps;
  static const int forced = getenv("MIVOS_MEMREAD_WGS") ? atoi(getenv("MIVOS_MEMREAD_WGS")) : 0;   // tuning only
  long long n_wg = forced > 0 ? forced : compute_units();
  // a stream is cut into at most MAX_SLOTS segments: ceil(tps / tiles_per_wg) + 1 <= MAX_SLOTS
  if (n_wg > (long long)p.streams * (MAX_SLOTS - 2)) n_wg = (long long)p.streams * (MAX_SLOTS - 2);
  if (n_wg > p.total) n_wg = p.total;
  if (n_wg < 1) n_wg = 1;
  p.tiles_per_wg = cdiv(p.total, n_wg);
  p.n_wg = cdiv(p.total, p.tiles_per_wg);
  p.slots = cdiv(p.tps, p.tiles_per_wg) + 1;
  if (p.slots > MAX_SLOTS) p.slots = MAX_SLOTS;
  p.L = top_k + SLACK;
  return p;
}

static int check_select_args(const float *keys, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem, int n_q,
                             int top_k, void *workspace, int64_t workspace_bytes) {
  if (!keys || !qk || !workspace) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: null pointer");
  if (top_k < 1 || top_k > MAX_TOPK) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: top_k=%d unsupported (1..%d)", top_k, MAX_TOPK);
  if (n_obj < 1 || n_q < 1 || n_mem < 1 || n_mem >= 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: bad sizes");
  if (n_mem < top_k) return fail(MIVOS_ERR_TOPK_RANGE, "selected index k out of range (top_k=%d > %lld memory positions)", top_k, (long long)n_mem);
  if (((uintptr_t)keys & 15) || ((uintptr_t)qk & 15) || (keys_ostride & 3)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: keys/qk must be 16-byte aligned");
  if (workspace_bytes < mivos_memory_read_workspace_bytes(n_obj, n_mem, n_q, top_k)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: workspace too small");
  return MIVOS_OK;
}

template <bool INDICES, int N>
static void launch_finalize_n(const Plan &pl, void *workspace, const float *values, int64_t values_ostride, float *out, int64_t out_ostride,
                              int64_t out_pstride, int32_t *idx_out, float *w_out, int n_obj, int n_q, int top_k, hipStream_t st) {
  hipLaunchKernelGGL((memread_finalize_kernel<INDICES, N>), dim3(n_q, n_obj), dim3(64), 0, st, (const uint64_t *)workspace, values,
                     (long long)values_ostride, out, (long long)out_ostride, (long long)out_pstride, idx_out, w_out, n_q, top_k,
                     pl.n_qtiles, pl.tps, pl.tiles_per_wg, pl.slots, pl.L);
}

static int launch_finalize(bool indices, const Plan &pl, void *workspace, const float *values, int64_t values_ostride, float *out,
                           int64_t out_ostride, int64_t out_pstride, int32_t *idx_out, float *w_out, int n_obj, int n_q, int top_k,
                           hipStream_t st) {
  const int per_lane = cdiv((long long)pl.slots * pl.L, 64);       // slots bounds the segments of any stream
#define MIVOS_FIN(N)                                                                                                              \
  (indices ? launch_finalize_n<true, N>(pl, workspace, values, values_ostride, out, out_ostride, out_pstride, idx_out, w_out, n_obj, n_q, top_k, st) \
           : launch_finalize_n<false, N>(pl, workspace, values, values_ostride, out, out_ostride, out_pstride, idx_out, w_out, n_obj, n_q, top_k, st))
  if (per_lane <= 4) MIVOS_FIN(4);
  else if (per_lane <= 8) MIVOS_FIN(8);
  else MIVOS_FIN(FIN_EPL_MAX);
#undef MIVOS_FIN
  return check_launch("memread_finalize");
}

}  // namespace mivos

using namespace mivos;

extern "C" int64_t mivos_memory_read_workspace_bytes(int n_obj, int64_t n_mem, int n_q, int top_k) {
  if (n_obj < 1 || n_q < 1 || n_mem < 1 || top_k < 1) return 0;
  const Plan p = make_plan(n_obj, n_mem, n_q, top_k);
  return (int64_t)p.streams * p.slots * QT * p.L * 8;
}

extern "C" int mivos_memory_read_plan(int n_obj, int64_t n_mem, int n_q, int top_k, int32_t *plan_out) {
  if (!plan_out || n_obj < 1 || n_q < 1 || n_mem < 1 || top_k < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_plan: bad arguments");
  const Plan p = make_plan(n_obj, n_mem, n_q, top_k);
  plan_out[0] = p.n_wg; plan_out[1] = p.tiles_per_wg; plan_out[2] = p.slots; plan_out[3] = p.tps; plan_out[4] = p.streams; plan_out[5] = p.L;
  return MIVOS_OK;
}

extern "C" int mivos_memory_read_select(const float *keys, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem,
                                        int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream) {
  if (int rc = check_select_args(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes)) return rc;
  const Plan pl = make_plan(n_obj, n_mem, n_q, top_k);
  SelectArgs a;
  a.keys = keys; a.keys_ostride = keys_ostride; a.qk = qk; a.lists = (uint64_t *)workspace; a.n_mem = n_mem; a.n_q = n_q;
  a.top_k = top_k; a.n_qtiles = pl.n_qtiles; a.tps = pl.tps; a.total_tiles = pl.total; a.tiles_per_wg = pl.tiles_per_wg;
  a.slots = pl.slots; a.L = pl.L;
  static const int abl = getenv("MIVOS_ABL") ? atoi(getenv("MIVOS_ABL")) : 0;   // profiling only
  if (abl == 1)
    hipLaunchKernelGGL(memread_select_kernel<1>, dim3(pl.n_wg), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(memread_select_kernel<0>, dim3(pl.n_wg), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("memread_select");
}

extern "C" int mivos_memory_read_finalize(const float *values, int64_t values_ostride, float *out, int64_t out_ostride,
                                          int64_t out_pstride, int n_obj, int64_t n_mem, int n_q, int top_k, void *workspace,
                                          int64_t workspace_bytes, void *stream) {
  if (!values || !out || !workspace || ((uintptr_t)values & 15) || ((uintptr_t)out & 15) || (out_pstride & 3) || (out_ostride & 3) || (values_ostride & 3))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: values/out must be non-null and 16-byte aligned");
  if (top_k < 1 || top_k > MAX_TOPK || n_mem < top_k || workspace_bytes < mivos_memory_read_workspace_bytes(n_obj, n_mem, n_q, top_k))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_finalize: arguments do not match the select call");
  const Plan pl = make_plan(n_obj, n_mem, n_q, top_k);
  return launch_finalize(false, pl, workspace, values, values_ostride, out, out_ostride, out_pstride, nullptr, nullptr, n_obj, n_q,
                         top_k, (hipStream_t)stream);
}

extern "C" int mivos_memory_read_topk(const float *keys, int64_t keys_ostride, const float *values, int64_t values_ostride,
                                      const float *qk, float *out, int64_t out_ostride, int64_t out_pstride, int n_obj,
                                      int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes,
                                      void *stream) {
  if (int rc = mivos_memory_read_select(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, stream)) return rc;
  return mivos_memory_read_finalize(values, values_ostride, out, out_ostride, out_pstride, n_obj, n_mem, n_q, top_k, workspace,
                                    workspace_bytes, stream);
}

extern "C" int mivos_memory_read_topk_indices(const float *keys, int64_t keys_ostride, const float *qk, int32_t *idx_out,
                                              float *weight_out, int n_obj, int64_t n_mem, int n_q, int top_k,
                                              void *workspace, int64_t workspace_bytes, void *stream) {
  if (!idx_out || !weight_out) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_indices: null pointer");
  if (int rc = mivos_memory_read_select(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, stream)) return rc;
  const Plan pl = make_plan(n_obj, n_mem, n_q, top_k);
  return launch_finalize(true, pl, workspace, nullptr, 0, nullptr, 0, 0, idx_out, weight_out, n_obj, n_q, top_k, (hipStream_t)stream);
}
