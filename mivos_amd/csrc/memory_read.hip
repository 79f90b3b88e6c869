// Space-time memory read for gfx950: MFMA affinity tiles (error-compensated fp16, or exact fp32) + streaming per-query
// top-k in LDS + softmax over the k survivors + sparse value readout.  The [T*H*W x H*W] affinity of the reference
// (prop_net.py:85-88, 52 MB/object at 480p T=5, 160 GB at 1080p T=200) is never materialised.
//
// Kernel 1 (memread_select): PERSISTENT, one 4-wave workgroup per CU (one wave per SIMD, 155 KB of LDS).
//   Work = streams x tiles: a stream is (object, tile of 64 queries) against the whole memory, cut into tiles of 32
//   memory positions.  The streams' tiles are laid end to end and dealt to the workgroups in equal contiguous runs
//   (stream-K style): perfect balance for any object count / frame size / bank depth, and a run that crosses a stream
//   boundary is processed as two segments.  Every segment leaves its own candidate list (>= its exact top-k) in the
//   workspace; kernel 2 merges the lists of a query exactly (the select launch leaves its plan in the workspace header).
//   Per wave: 16 queries (pre-scaled by 1/sqrt(128)) live in 32 VGPRs as the MFMA B operand; key tiles go global ->
//   registers -> LDS once per workgroup (four tiles in flight, requests in inline assembly with an explicit vmcnt; two LDS
//   buffers, one barrier per tile, 528-byte row pitch => conflict-free ds_read_b128; the fragments of tile t+1 are read
//   into a second register set during tile t); the MFMAs of a tile give the 32x16 score tile on two accumulators (rows
//   0-15 / 16-31); lane (j, g) owns 8 scores of query j.
//   Selection is software pipelined: the 8 score registers of tile t-1 are compared / appended between the MFMAs of tile
//   t.  (Beside fp32 MFMAs nothing a wave issues is hidden - measured, scripts/ubench/README.md - so the append path is
//   as short as it can be: compare, index, address, one exec-masked ds_write2_b32, fill level; beside fp16 MFMAs it is.)
//   Candidates: lane (q, g) appends raw {score bits, index} entries to ITS region of REG entries of query q's buffer
//   (private fill level in a VGPR: no atomics).  When a region may overflow the owning wave compacts the query's four
//   regions: entries -> orderable keys {score, ~index}, bisection (common high bits skipped, score words first) stopped as
//   soon as between k and k+16 entries survive (an exact cut is not needed until the end), survivors dealt back round
//   robin, tau := the cut.  Ties: lower memory index wins (torch.topk leaves ties unspecified).
//   F16 variant (the engine's default precision, "f16x3"): the same kernel with the affinity on the fp16 matrix pipe, error
//   compensated like the convolutions: keys and queries are split x = hi + lo (two fp16, 22 significant bits), a k-step of
//   32 channels is three v_mfma_f32_16x16x32_f16 (lo*hi + hi*lo + hi*hi, fp32 accumulate; the lo*lo term is below fp32
//   rounding).  24 MFMAs of ~17 cycles per tile instead of 64 of 32: the kernel is then bound by the selection at 480p and by
//   the key stream at 1080p.  Keys come PRE-SPLIT (mivos_memory_split_keys, once per memorised frame): a row is the same
//   512 bytes, laid out so that the loader, the LDS tile and the fragment reads are byte-for-byte those of the fp32 variant -
//   block b (64 halves) = for ks in 0..3: hi[8] | lo[8] of channels 32 ks + 8 b + e.
// Kernel 2 (memread_finalize): one wave per (object, query): exact k-th largest of the merged lists (same bisection,
//   run to the end), exp(s - s_max)/sum in rank order like the reference, then the k value rows (2 KB each) are gathered
//   in ascending memory index - the order in which the reference's dense bmm meets its non-zeros.
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "conv_common.h"

namespace mivos {

constexpr int CK = 128, CV = 512;
constexpr int QW = 16;      // queries per wave (MFMA N)
constexpr int QT = 64;      // queries per workgroup
constexpr int KT = 32;      // memory positions per LDS tile (two 16-row MFMA sub-tiles)
constexpr int KLD = 132;    // LDS pitch of a key row (floats)
constexpr int STAGE_DEPTH2 = 4;  // ... of the 32-queries-per-wave kernel (its tile loop is written out for four sets)
constexpr int STAGE_DEPTH = 4;   // key tiles in flight per workgroup on their way global -> registers -> LDS (16 VGPRs each)
constexpr int REG = 61;                 // lane (q, g) appends to ITS region of REG entries: private fill level in a VGPR, no
                                        // atomics.  Regions are laid out [wave][g][q][REG]: the 16 lanes of a ds_write_b64 lane
                                        // group (same g, q = 0..15) are REG entries = 122 dwords apart, and with REG odd that
                                        // walks all 32 bank pairs (REG = 60, or the query-major order, gave 8-way conflicts)
constexpr int CAP = 4 * REG;            // candidate slots per query
constexpr int REG_TRIGGER = REG - 1 - 8;   // a tile adds at most 8 candidates per lane (one slot of the 61 is padding)
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

// max over the 64 lanes (wave-uniform): rotate-and-max butterfly inside the four 16-lane rows (DPP row_ror, no LDS),
// then the four row results through SGPRs.
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
  const uint32_t a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

// Largest 64-bit prefix p with #(e >= p) >= k, found MSB first; stops early once k <= #(e >= p) <= k + slack.
// N entries per lane, empty entries are 0; at least k entries are valid.  Wave-uniform result; `count` = #(e >= p).
// The bits every valid entry shares (sign, exponent, ... of scores in a narrow band) are skipped, and the search runs on the
// 32-bit score words alone (single-rate compares) until the cut is found; only exact score ties straddling the cut make it
// continue into the index words.
template <int N>
__device__ __forceinline__ uint64_t bisect_kth(const uint64_t (&e)[N], int k, int slack, int &count) {
  uint32_t h[N];
  uint32_t mx = 0u, nmn = 0u;                        // max of the score words, max of their complements (= ~min) over valid entries
  int valid = 0;
#pragma unroll
  for (int t = 0; t < N; ++t) {
    h[t] = (uint32_t)(e[t] >> 32);
    mx = max(mx, h[t]);
    nmn = max(nmn, e[t] != 0ull ? ~h[t] : 0u);
    valid += __popcll(__ballot(e[t] != 0ull));
  }
  mx = wave_umax(mx);
  const uint32_t mn = ~wave_umax(nmn);
  const uint32_t diff = mx ^ mn;
  uint32_t ph = diff ? (mx & ~((2u << (31 - __builtin_clz(diff))) - 1u)) : mx;   // the common high bits
  int c_at = valid;
  bool done = valid <= k + slack;
  if (!done && diff) {
#pragma unroll 1
    for (int b = 31 - __builtin_clz(diff); b >= 0; --b) {
      const uint32_t trial = ph | (1u << b);
      int c = 0;
#pragma unroll
      for (int t = 0; t < N; ++t) c += __popcll(__ballot(h[t] >= trial));
      if (c >= k) {
        ph = trial;
        c_at = c;
        if (c <= k + slack) { done = true; break; }
      }
    }
  }
  uint64_t prefix = (uint64_t)ph << 32;
  if (!done) {                                       // equal scores straddle the cut: decide by index (lower index = larger key)
#pragma unroll 1
    for (int b = 31; b >= 0; --b) {
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
  }
  count = c_at;
  return prefix;
}

// Lanes of one wave hand data to each other through LDS (lane 0 publishes a count / threshold, every lane reads it).
// The hardware executes a wave's LDS operations in order, but the COMPILER reasons per thread: without this barrier it
// forwards a thread's own earlier load across another lane's store ("nobody in this thread wrote it").
__device__ __forceinline__ void wave_lds_handoff() { asm volatile("" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes: shifts inside the rows of 16 (DPP row_shr), then the row totals travel with the
// two row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
__device__ __forceinline__ int wave_prefix_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31
  return v;
}

// Same contract as bisect_kth (a prefix p with k <= #(e >= p) <= k + slack, callers guarantee more than k + slack valid
// entries), found by COUNTING instead of bit by bit: the score words of the valid entries span [mn, mx]; that range is cut into
// 64 buckets of a power-of-two width (mn = the caller's current threshold, below which there are no candidates), every entry adds one to its bucket's counter in `hist` (64 words of LDS owned by this
// wave; lane L reads back bucket 63 - L), a prefix sum over the lanes gives "entries in this bucket or above", and the first
// lane whose sum reaches k names the bucket the cut lies in.  If that leaves more than k + slack survivors the bucket is cut
// into 64 again (6 more bits of the score per level).  bisect_kth needs one dependent compare / count / branch round per BIT
// between the highest bit in which the scores differ and the cut - 15 rounds when the pool holds scores of both signs, as it
// does at the first compaction of every segment; measured 3 160 cycles per compaction at one wave per SIMD,
// profiles/r02f_memread_cycles.txt - this needs two or three levels.  Exact score ties straddling the cut fall through to the
// index bisection, like there.
template <int N>
__device__ __forceinline__ uint64_t count_kth(const uint64_t (&e)[N], int k, int slack, int &count, uint32_t *hist, int lane, uint32_t lo) {
  // `lo`: a score word no valid entry lies below (the caller's current threshold: candidates were appended because they beat it)
  uint32_t h[N];
  uint32_t mx = 0u;
#pragma unroll
  for (int t = 0; t < N; ++t) {
    h[t] = (uint32_t)(e[t] >> 32);
    mx = max(mx, h[t]);
  }
  mx = wave_umax(mx);
  const uint32_t range = mx - lo;
  int shift = range ? 26 - __builtin_clz(range) : 0;       // (range >> shift) < 64
  if (shift < 0) shift = 0;
  uint32_t width_m1 = 0xffffffffu;                          // entries of the current range: lo <= h <= lo + width_m1
  int above = 0;                                            // entries above the current range (all survive)
  int total;
  for (;;) {
    hist[lane] = 0u;
    wave_lds_handoff();
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const uint32_t d = h[t] - lo;
      if (e[t] != 0ull && h[t] >= lo && d <= width_m1) atomicAdd(&hist[63 - (int)(d >> shift)], 1u);
    }
    wave_lds_handoff();
    const int sum = wave_prefix_sum((int)hist[lane]);       // entries of the range in buckets >= 63 - lane
    const int need = k - above;                             // >= 1, and the range holds at least that many
    const unsigned long long m = __ballot(sum >= need);
    const int l = __builtin_ctzll(m);
    total = above + __builtin_amdgcn_readlane(sum, l);
    lo += (uint32_t)(63 - l) << shift;                      // lower edge of the bucket the cut lies in
    if (total <= k + slack || shift == 0) break;
    above += l ? __builtin_amdgcn_readlane(sum, l - 1) : 0;
    width_m1 = (1u << shift) - 1u;
    shift = shift > 6 ? shift - 6 : 0;
  }
  uint64_t prefix = (uint64_t)lo << 32;
  if (total > k + slack) {                                  // equal scores straddle the cut: decide by index (lower index = larger key)
#pragma unroll 1
    for (int b = 31; b >= 0; --b) {
      const uint64_t trial = prefix | (1ull << b);
      int c = 0;
#pragma unroll
      for (int t = 0; t < N; ++t) c += __popcll(__ballot(e[t] >= trial));
      if (c >= k) {
        prefix = trial;
        total = c;
        if (c <= k + slack) break;
      }
    }
  }
  count = total;
  return prefix;
}

// Compaction of one query's LDS candidate buffer by the owning wave (all 64 lanes).  The buffer is four lane-private
// regions of REG entries (region g is appended to by lane (q, g) only, whose VGPR `my_cnt` is its fill level); n0..n3 are
// the four levels.  More than k + SLACK entries -> between k and k + SLACK survivors, dealt round-robin back to the four
// regions; returns the survivor count c (region g then holds (c - g + 3) / 4) and the new threshold.  Only the owning wave
// touches the buffer and the LDS operations of one wave execute in order, so no barrier is needed.
// LDS candidate entries are RAW {score bits (high word), memory index (low word)}: the append path runs for every score of
// every tile (and a wave's vector instructions are not hidden behind its own fp32 MFMAs - they share the SIMD's FMA lanes),
// so the conversion to the orderable key {f2ord(score), ~index} happens where entries are read back (compaction, list copy).
__device__ __forceinline__ uint64_t raw_to_key(uint64_t raw) {
  return ((uint64_t)f2ord(__uint_as_float((uint32_t)(raw >> 32))) << 32) | (uint64_t)(~(uint32_t)raw);
}

__device__ __forceinline__ int compact_query(uint64_t *buf, int n0, int n1, int n2, int n3, int k, int lane, float &new_tau, uint32_t *hist) {
  constexpr int GS = QW * REG;                         // region g of this query starts at buf + g * GS
  wave_lds_handoff();
  // unconditional loads (lanes past a region's fill level read stale slots of the same region, lanes 61-63 its last slot) and
  // selects instead of four exec-masked load + convert blocks: the compaction is a chain of dependent instructions issued by
  // one wave, so every branch and exec round trip in it is paid in full
  const int li = lane < REG ? lane : REG - 1;
  uint64_t raw[4], e[4];
  raw[0] = buf[li];
  raw[1] = buf[GS + li];
  raw[2] = buf[2 * GS + li];
  raw[3] = buf[3 * GS + li];
  const uint64_t k0 = raw_to_key(raw[0]), k1 = raw_to_key(raw[1]), k2 = raw_to_key(raw[2]), k3 = raw_to_key(raw[3]);
  e[0] = lane < n0 ? k0 : 0ull;
  e[1] = lane < n1 ? k1 : 0ull;
  e[2] = lane < n2 ? k2 : 0ull;
  e[3] = lane < n3 ? k3 : 0ull;
  int c = n0 + n1 + n2 + n3;
  uint64_t p = 1ull;                                   // one region nearly full, few entries overall: only rebalance
  if (c > k + SLACK) p = count_kth<4>(e, k, SLACK, c, hist, lane, f2ord(new_tau));   // new_tau comes in as the current threshold
  int base = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool keep = e[t] >= p;
    const unsigned long long m = __ballot(keep);
    const int r = base + __popcll(m & below);         // rank among the survivors -> region r & 3, slot r >> 2
    if (keep) buf[(r & 3) * GS + (r >> 2)] = raw[t];
    base += __popcll(m);
  }
  wave_lds_handoff();
  // later positions have higher indices: one whose score EQUALS the cut's score loses the tie against all c >= k survivors
  if (p != 1ull) new_tau = ord2f((uint32_t)(p >> 32));
  return c;
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
  int *header;               // first 64 bytes of the workspace: the plan this launch used, for the finalize kernel
  int qt;                    // queries per workgroup (QT, QT2 for the 32-queries-per-wave kernel, QT3 for the 8-wave one)
  uint64_t *cand;            // QT3 kernel: global candidate regions, CAND3_PER_WG entries per workgroup
  const float *kmax2;        // QT3 kernel, hi-first variant: per object, the largest squared norm of a key row (key_norm2_max_kernel)
  unsigned long long *dbg;   // profiling builds: {shader cycles, tiles} of workgroup 0 / wave 0 (NULL otherwise)
  int contig;                // 64-query kernel: chunks dealt XCD-major (1) or in block order (0; tuning / A-B only, MIVOS_XCD_CONTIG)
};

// The finalize kernel takes the work partition from the workspace header the select launch left there (the two select
// kernels cut the work differently; a finalize call only repeats the sizes).
constexpr int HEADER_BYTES = 64;
__device__ __forceinline__ void write_plan_header(const SelectArgs &a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.header[0] = a.qt; a.header[1] = a.n_qtiles; a.header[2] = a.tps; a.header[3] = a.tiles_per_wg; a.header[4] = a.slots; a.header[5] = a.L;
  }
}

// ABL: ablation switch for profiling builds (0 = product, 1 = MFMA + staging only).
// BR: long memories (1080p, hundreds of frames): once the threshold has converged almost no score passes, so the append
//     (address, store, fill level) is skipped by a wave-uniform branch on the compare's mask; with short memories (480p) a
//     step nearly always has a passing lane and the branch would only add to the instruction count.
// F16: error-compensated fp16 MFMA affinity on pre-split keys (see the header); false = exact fp32 MFMA on fp32 keys.
template <int ABL, bool BR, bool F16>
__global__ __launch_bounds__(256, 1) void memread_select_kernel(const SelectArgs a) {
  __shared__ __attribute__((aligned(16))) float ktile[2][KT * KLD];
  __shared__ uint64_t cand[QT * CAP];                 // [wave][g][q][REG]
  __shared__ uint32_t hist[4][64];                    // bucket counters of a compaction (count_kth), one set per wave

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jq = lane & 15, g = lane >> 4;
  const int qslot = wave * QW + jq;
  // channel of (MFMA step u, k-slot g, element s): 64 (g&1) + 32 (g>>1) + 4u + s.  Any bijection works as long as keys
  // and queries use the same one; this one makes the two k-slots that share a ds_read_b128 lane group (g, g^1) hit the
  // same banks with different rows => conflict-free fragment reads (row pitch 132 floats).
  const int coff = 64 * (g & 1) + 32 * (g >> 1);
  const int lrow = tid >> 3, lc = tid & 7;           // key-tile loader: row tid>>3, float4 columns lc + 8 jj

  // Chunks are dealt XCD-major: workgroup b (which runs on XCD b % 8) takes chunk (b % 8) * (n / 8) + b / 8, so that the 32 workgroups
  // of an XCD walk neighbouring streams = the SAME object's key tiles at about the same time and one fetch through the fabric serves them
  // all (the eight L2s are not shared).  Round 3 measured this form in isolation - 283.4 vs 283.9 us, the kernel does not wait for keys -
  // and dropped it.  Round 4 measured the fabric: 916 -> 376 MB of reads per launch (51 x -> 21 x the keys) at equal time in situ
  // (353 vs 356 us, 206.2 vs 205.3 frames/s on the driver's window; profiles/r04f_select_xcd_ab.txt, r04f_config3_pmc_traffic.json):
  // kept for the traffic - the FusionNet kernels of the previous frame share the chip and the fabric with this kernel.
  const int chunk = a.contig ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  long long t_begin = (long long)chunk * a.tiles_per_wg;
  const long long t_end = (t_begin + a.tiles_per_wg < a.total_tiles) ? t_begin + a.tiles_per_wg : a.total_tiles;
  const bool prof = a.dbg && blockIdx.x == 0;       // profiling builds only (MIVOS_MEMREAD_DBG)
  const unsigned long long clk0 = prof ? __builtin_readcyclecounter() : 0ull;
  unsigned long long clk_room = 0ull, clk_final = 0ull, clk_pro = 0ull, n_compact = 0ull;
  write_plan_header(a);

  while (t_begin < t_end) {
    const int stream = (int)(t_begin / a.tps);
    const int seg_lo = (int)(t_begin - (long long)stream * a.tps);
    int seg_hi = seg_lo + (int)(t_end - t_begin);
    if (seg_hi > a.tps) seg_hi = a.tps;
    const int nt = seg_hi - seg_lo;
    const int obj = stream / a.n_qtiles, qtile = stream - obj * a.n_qtiles;
    const int slot = chunk - (int)(((long long)stream * a.tps) / a.tiles_per_wg);
    // memory positions are 32-bit (n_mem < 2^31 is checked on the host)
    const int r0 = seg_lo * KT;
    const int r1 = ((long long)seg_hi * KT < a.n_mem) ? seg_hi * KT : (int)a.n_mem;
    const float *kbase = a.keys + (long long)obj * a.keys_ostride;

    __syncthreads();                                  // previous segment completely done with LDS

    // B operand: this lane's query row pieces, scaled like prop_net.py:86 (qk / sqrt(CK), a true division).
    // fp32: qreg[u] = channels coff + 4u .. +3.  F16: block b = coff / 32 of the split layout - qreg[2 ks] = hi, qreg[2 ks + 1]
    // = lo of channels 32 ks + 8 b + e (e = 0..7), as two packed half8 vectors in the same 32 registers.
    f32x4_t qreg[8];
    {
      const int q = qtile * QT + qslot;
      const float *qrow = a.qk + (long long)(q < a.n_q ? q : a.n_q - 1) * CK;
      const float d = sqrtf((float)CK);
      if (F16) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float *src = qrow + 32 * ks + (coff >> 2);
          const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(src), v1 = *reinterpret_cast<const f32x4_t *>(src + 4);
          half8_t hi, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (e < 4 ? v0[e & 3] : v1[e & 3]) / d;
            hi[e] = (_Float16)x;
            lo[e] = (_Float16)(x - (float)hi[e]);
          }
          qreg[2 * ks] = __builtin_bit_cast(f32x4_t, hi);
          qreg[2 * ks + 1] = __builtin_bit_cast(f32x4_t, lo);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          f32x4_t v = *reinterpret_cast<const f32x4_t *>(qrow + coff + 4 * u);
          v.x /= d; v.y /= d; v.z /= d; v.w /= d;
          qreg[u] = v;
        }
      }
      // the query fragments are finished HERE, before the key requests below go out: the compiler waits for its own loads
      // with counts that do not know about those requests, so a later use would wait for the whole prefetch as well
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(qreg[u]));
    }

    // global -> register -> LDS staging of key tiles, STAGE_DEPTH register sets in flight: a tile is requested STAGE_DEPTH
    // iterations before it is written to LDS.  What bounds the tile loop is the rate at which a CU can pull its 16 KB per tile
    // through the L2 (scripts/ubench/mfma_f16_tile.hip: with two tiles in flight 0.40 - 0.67 us per tile, more than the MFMAs
    // take), i.e. latency x bytes in flight.  The requests are inline assembly on purpose: the compiler's vmcnt bookkeeping
    // loses the order of loads carried around the loop and waits for the NEWEST request whenever the oldest is needed
    // (measured: one tile in flight); here nothing is tracked and the wait is written out (requests complete in order).
    f32x4_t kr[STAGE_DEPTH][4];
    auto gload = [&](f32x4_t (&krs)[4], int kb) {
      // rows past the end of the segment are read from its first row instead (always a valid address): their scores are
      // never selected, and the request count per iteration stays constant, which the explicit vmcnt below relies on
      const int m = kb + lrow;
      const f32x4_t *src = reinterpret_cast<const f32x4_t *>(kbase + (long long)(m < r1 ? m : r0) * CK) + lc;
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:128\n\t"
                   "global_load_dwordx4 %2, %4, off offset:256\n\tglobal_load_dwordx4 %3, %4, off offset:384"
                   : "=&v"(krs[0]), "=&v"(krs[1]), "=&v"(krs[2]), "=&v"(krs[3]) : "v"(src) : "memory");
    };
    auto lds_store = [&](f32x4_t (&krs)[4], int buf) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4_t *>(&ktile[buf][lrow * KLD + 4 * (lc + 8 * jj)]) = krs[jj];
    };

    // scores of the previous tile (software pipeline): -inf = nothing to select on the first tile
    f32x4_t p0 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, p1 = p0;
    int pb = r0;                                      // base row of the previous tile
    float my_tau = -INFINITY;

    // Selection of score register i (sub-tile i>>2, row 4g + (i&3)) of the previous tile, BRANCH-FREE and cut into three
    // slices that are placed between the MFMAs of the running tile: every lane takes part in the LDS atomic (adding 0 when
    // its score does not pass) and in the candidate store (to a private dump slot when it does not pass), so the whole tile
    // is one basic block and the slices issue in the shadow of the matrix pipe.
    // Lane (q, g) appends to ITS region of query q's buffer at its private fill level: no atomic, no LDS round trip.
    uint64_t *const my_region = cand + ((wave * 4 + g) * QW + jq) * REG;
    // Append path, per score register: ONE vector compare, the index (one add to a per-tile base), and - executed by the
    // passing lanes only, exec := the compare's mask, no branch - the advance of the lane's write pointer and one
    // ds_write2_b32 through it.  The fill level IS that pointer (my_top = LDS byte address of the newest entry; the level is
    // only needed as a number when a compaction runs), so there is no address arithmetic and no add-with-carry per score:
    // 3 vector + 1 LDS + 2 scalar instructions (5 + 1 + 2 with a counter).  Measured (MIVOS_ABL=2..4): a wave's VALU /
    // LDS-store instructions cost their full issue time next to its own fp32 MFMAs - nothing here is hidden - so the
    // instruction count is the cost; entries stay raw {score bits, index}, the orderable key is built at compaction time.
    const uint32_t region_lds = (uint32_t)(size_t)my_region;     // LDS byte address of this lane's region
    uint32_t my_top = region_lds - 8u;                // no entry yet
    auto fill_level = [&]() { return (int)((my_top + 8u - region_lds) >> 3); };
    bool s_pass;
    uint32_t idx_base = 0u;                           // pb + 4g: index of (sub, r) = idx_base + 16 sub + r
    auto slice_a = [&](int i) { s_pass = ((i < 4) ? p0[i & 3] : p1[i & 3]) > my_tau; };
    auto slice_b = [&](int i) {
      const unsigned long long m = __ballot(s_pass);
      if (BR && m == 0ull) return;
      const uint32_t idx = idx_base + (uint32_t)(16 * (i >> 2) + (i & 3));
      const uint32_t bits = __float_as_uint((i < 4) ? p0[i & 3] : p1[i & 3]);
      unsigned long long saved;
      asm volatile("s_and_saveexec_b64 %0, %2\n\tv_add_u32_e32 %1, 8, %1\n\tds_write2_b32 %1, %3, %4 offset1:1\n\ts_mov_b64 exec, %0"
                   : "=&s"(saved), "+v"(my_top) : "s"(m), "v"(idx), "v"(bits) : "memory");
    };
    // latch the scores of tile t for the selection that runs during tile t+1; only the last tile of a stream can hold rows
    // past the end of the memory (scores of whatever row was loaded instead): those become -inf here
    auto latch_scores = [&](const f32x4_t &a0, const f32x4_t &a1, int t) {
      p0 = a0; p1 = a1;
      pb = r0 + t * KT;
      idx_base = (uint32_t)(pb + 4 * g);
      if (pb + KT > r1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (pb + 4 * g + i >= r1) p0[i] = -INFINITY;
          if (pb + 16 + 4 * g + i >= r1) p1[i] = -INFINITY;
        }
      }
    };
    // compaction of query `ql` (0..15) of this wave: fill levels come from the four owner lanes, go back to them
    auto compact_one = [&](int ql, bool force) {
      const int my_cnt = fill_level();
      const int n0 = __builtin_amdgcn_readlane(my_cnt, ql), n1 = __builtin_amdgcn_readlane(my_cnt, ql + 16);
      const int n2 = __builtin_amdgcn_readlane(my_cnt, ql + 32), n3 = __builtin_amdgcn_readlane(my_cnt, ql + 48);
      if (force && n0 + n1 + n2 + n3 <= a.top_k + SLACK) return;   // end of a segment: the list takes the regions as they are
      // query ql's threshold, wave-uniform (it stays when nothing is dropped); compact_query also takes it as the lower bound
      float nt_tau = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(my_tau), ql));
      const int c = compact_query(cand + (wave * 4 * QW + ql) * REG, n0, n1, n2, n3, a.top_k, lane, nt_tau, hist[wave]);
      if (jq == ql) { my_top = region_lds - 8u + 8u * (uint32_t)((c - g + 3) >> 2); my_tau = nt_tau; }
    };
    auto make_room = [&]() {
      // before the (up to 8 per lane) appends of a tile: compact every buffer of this wave with a region that might overflow
      const unsigned long long full = __ballot((int)(my_top - region_lds) > 8 * REG_TRIGGER - 8);
      unsigned need = (unsigned)((full | (full >> 16) | (full >> 32) | (full >> 48)) & 0xffffull);
      if (need) {
        const unsigned long long c0 = prof ? __builtin_readcyclecounter() : 0ull;
        while (need) {
          const int ql = __builtin_ctz(need);
          need &= need - 1;
          compact_one(ql, false);
          n_compact += prof ? 1 : 0;
        }
        if (prof) clk_room += __builtin_readcyclecounter() - c0;
      }
    };

    const unsigned long long cpro = prof ? __builtin_readcyclecounter() : 0ull;
    // Prologue.  The A fragments of tile t+1 are read from LDS into a second register set DURING the MFMAs of tile t (two
    // ds_read_b128 per 8 MFMAs) instead of in one burst after the barrier; tile t+2 is then written over tile t's LDS copy
    // (dead once every wave holds it in registers), so two LDS buffers suffice and there is one barrier per tile.
    f32x4_t fa0[8], fa1[8], fb0[8], fb1[8];
    gload(kr[0], r0);
    gload(kr[1], r0 + KT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_store(kr[0], 0);
    lds_store(kr[1], 1);
#pragma unroll
    for (int d = 0; d < STAGE_DEPTH; ++d) gload(kr[d], r0 + (2 + d) * KT);     // set d: tiles 2 + d, 2 + d + STAGE_DEPTH, ...
    __syncthreads();
    {
      const float *arow0 = &ktile[0][jq * KLD + coff];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        fa0[u] = *reinterpret_cast<const f32x4_t *>(arow0 + 4 * u);
        fa1[u] = *reinterpret_cast<const f32x4_t *>(arow0 + 16 * KLD + 4 * u);
      }
    }
    __syncthreads();                                  // every wave holds tile 0 in registers: its LDS copy is dead

    if (prof) clk_pro += __builtin_readcyclecounter() - cpro;
    // one tile: MFMAs on fragment set F (tile t) while G receives tile t+1's fragments and tile t-1's scores are selected
    auto tile_iter = [&](int t, f32x4_t (&F0)[8], f32x4_t (&F1)[8], f32x4_t (&G0)[8], f32x4_t (&G1)[8], f32x4_t (&kr)[4]) {
      // tile t+2 (requested STAGE_DEPTH iterations ago: everything but the STAGE_DEPTH - 1 younger sets has landed) goes over
      // tile t's dead LDS copy, its registers take the request for tile t + 2 + STAGE_DEPTH.  Unconditional: past the end of
      // the segment the rows are clamped and the LDS copy is never read.
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (STAGE_DEPTH - 1)) : "memory");
      lds_store(kr, t & 1);
      gload(kr, r0 + (t + 2 + STAGE_DEPTH) * KT);
      if (ABL == 0) make_room();
      const float *nrow0 = &ktile[(t + 1) & 1][jq * KLD + coff];   // tile t+1 (stale data past the segment's end: unused)
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      // A single wave issues a vector instruction every ~8 cycles, an MFMA occupies the matrix pipe for 32: at most ~3 other
      // instructions fit behind each MFMA, and everything beyond that in one gap idles the pipe (measured: clusters of 5-6
      // cost their full issue time).  So the selection is dealt out ONE OR TWO instructions per MFMA, pinned by scheduling
      // barriers (the compiler would otherwise regroup them).
#define MIVOS_SB __builtin_amdgcn_sched_barrier(0);
      if (F16) {
        // k-step ks: F[2 ks] = hi, F[2 ks + 1] = lo halves of the key rows; small terms first.  Score registers 2 ks and
        // 2 ks + 1 of the previous tile are selected in the gaps (an fp16 MFMA leaves the vector ALU free, unlike fp32).
#define MIVOS_HF(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, A), __builtin_bit_cast(half8_t, B), ACC, 0, 0, 0); MIVOS_SB
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          G0[2 * ks] = *reinterpret_cast<const f32x4_t *>(nrow0 + 8 * ks); MIVOS_SB
          MIVOS_HF(acc0, F0[2 * ks + 1], qreg[2 * ks])
          if (ABL != 1) slice_a(2 * ks);
          MIVOS_SB
          MIVOS_HF(acc1, F1[2 * ks + 1], qreg[2 * ks])
          G0[2 * ks + 1] = *reinterpret_cast<const f32x4_t *>(nrow0 + 8 * ks + 4); MIVOS_SB
          MIVOS_HF(acc0, F0[2 * ks], qreg[2 * ks + 1])
          if (ABL != 1) slice_b(2 * ks);
          MIVOS_SB
          MIVOS_HF(acc1, F1[2 * ks], qreg[2 * ks + 1])
          G1[2 * ks] = *reinterpret_cast<const f32x4_t *>(nrow0 + 16 * KLD + 8 * ks); MIVOS_SB
          MIVOS_HF(acc0, F0[2 * ks], qreg[2 * ks])
          if (ABL != 1) slice_a(2 * ks + 1);
          MIVOS_SB
          G1[2 * ks + 1] = *reinterpret_cast<const f32x4_t *>(nrow0 + 16 * KLD + 8 * ks + 4); MIVOS_SB
          MIVOS_HF(acc1, F1[2 * ks], qreg[2 * ks])
          if (ABL != 1) slice_b(2 * ks + 1);
          MIVOS_SB
        }
#undef MIVOS_HF
      } else {
#define MIVOS_MF0(U, S) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(F0[U][S], qreg[U][S], acc0, 0, 0, 0); MIVOS_SB
#define MIVOS_MF1(U, S) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(F1[U][S], qreg[U][S], acc1, 0, 0, 0); MIVOS_SB
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          G0[u] = *reinterpret_cast<const f32x4_t *>(nrow0 + 4 * u); MIVOS_SB
          MIVOS_MF0(u, 0)
          if (ABL != 1) slice_a(u);
          MIVOS_SB
          MIVOS_MF1(u, 0)
          G1[u] = *reinterpret_cast<const f32x4_t *>(nrow0 + 16 * KLD + 4 * u); MIVOS_SB
          MIVOS_MF0(u, 1)
          MIVOS_MF1(u, 1)
          if (ABL != 1) slice_b(u);
          MIVOS_SB
          MIVOS_MF0(u, 2)
          MIVOS_MF1(u, 2)
          MIVOS_MF0(u, 3)
          MIVOS_MF1(u, 3)
        }
#undef MIVOS_MF0
#undef MIVOS_MF1
      }
#undef MIVOS_SB
      if (ABL == 1 && t >= 2) {                 // keep the MFMA results alive without selecting
        const float sum = (acc0.x + acc0.y) + (acc1.z + acc1.w);
        if (sum == 123.456f) my_tau = sum;
      } else {
        latch_scores(acc0, acc1, t);
      }
      __syncthreads();
    };
    static_assert(STAGE_DEPTH % 2 == 0, "fragment sets alternate with the tile parity");
    for (int t = 0; t < nt; t += STAGE_DEPTH) {
#pragma unroll
      for (int d = 0; d < STAGE_DEPTH; ++d) {         // unrolled: staging set d and the fragment ping-pong are compile-time
        if (t + d < nt) {
          if (d & 1) tile_iter(t + d, fb0, fb1, fa0, fa1, kr[d]);
          else tile_iter(t + d, fa0, fa1, fb0, fb1, kr[d]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests past the end of the segment: their registers are reused
    // drain the pipeline: select on the last tile
    const unsigned long long cfin = prof ? __builtin_readcyclecounter() : 0ull;
    make_room();
#pragma unroll
    for (int i = 0; i < 8; ++i) { slice_a(i); slice_b(i); }

    // this segment's candidate lists: for each of the wave's 16 queries between min(n, k) and k + SLACK entries
    for (int ql = 0; ql < QW; ++ql) {
      compact_one(ql, true);
      wave_lds_handoff();
      const int my_cnt = fill_level();
      const int n0 = __builtin_amdgcn_readlane(my_cnt, ql), n1 = __builtin_amdgcn_readlane(my_cnt, ql + 16);
      const int n2 = __builtin_amdgcn_readlane(my_cnt, ql + 32), n3 = __builtin_amdgcn_readlane(my_cnt, ql + 48);
      const int s = wave * QW + ql;
      const uint64_t *src = cand + (wave * 4 * QW + ql) * REG;     // region g at src + g * QW * REG
      uint64_t *dst = a.lists + (((long long)stream * a.slots + slot) * QT + s) * a.L;
      // n0 + n1 + n2 + n3 <= k + SLACK = L <= 80: lanes 0..REG-1 copy one entry of each region, the rest is zero-filled
      if (lane < n0) dst[lane] = raw_to_key(src[lane]);
      if (lane < n1) dst[n0 + lane] = raw_to_key(src[QW * REG + lane]);
      if (lane < n2) dst[n0 + n1 + lane] = raw_to_key(src[2 * QW * REG + lane]);
      if (lane < n3) dst[n0 + n1 + n2 + lane] = raw_to_key(src[3 * QW * REG + lane]);
      for (int i = n0 + n1 + n2 + n3 + lane; i < a.L; i += 64) dst[i] = 0ull;
    }
    if (prof) clk_final += __builtin_readcyclecounter() - cfin;
    t_begin += nt;
  }
  if (prof && tid == 0) {
    a.dbg[0] = __builtin_readcyclecounter() - clk0;
    a.dbg[1] = (unsigned long long)(t_end - (long long)chunk * a.tiles_per_wg);
    a.dbg[2] = clk_room; a.dbg[3] = n_compact; a.dbg[4] = clk_final; a.dbg[5] = clk_pro;
  }
}

// ---- the 32-queries-per-wave variant of the fp16 kernel ------------------------------------------------------------
// Same work partition, staging, candidate handling and lists as memread_select_kernel<.., F16 = true>, but a workgroup
// covers QT2 = 128 queries: each wave holds 32 queries (64 VGPRs of hi/lo fp16 B fragments) and runs
// v_mfma_f32_32x32x16_f16 on the 32-key tile, so a key fragment read from LDS serves twice as many queries - the fp16
// kernel above is bound by the LDS traffic of the key tile (80 KB moved per 32 keys x 64 queries), not by its 24 MFMAs.
//   Per tile and wave: 8 k-steps of 16 channels x 3 products = 24 MFMAs (32 cycles each), 16 ds_read_b128.
//   Lane (j, h) = (lane & 31, lane >> 5) owns the 16 scores of query j against keys 8 (r >> 2) + 4 h + (r & 3), r = 0..15,
//   one threshold and ONE candidate region of REG2 entries; a query's two regions (h = 0, 1) are compacted together.
//   The pre-split key rows are the same bytes: chunk (ks, h) of the fp16 k-step is at float offset 16 ks + 8 h (hi; lo at
//   + 4) of the row and holds channels 64 (ks & 1) + 32 h + 8 (ks >> 1) + e.
constexpr int QW2 = 32;                       // queries per wave
constexpr int QT2 = 128;                      // queries per workgroup
constexpr int REG2 = 61;                      // entries per region (odd: 16 consecutive lanes' regions walk all bank pairs)
constexpr int REG2_TRIGGER = REG2 - 1 - 8;    // room is made before each group of 8 appends per lane (twice per tile)

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ int compact_query2(uint64_t *buf, int n0, int n1, int k, int lane, float &new_tau) {
  constexpr int GS = QW2 * REG2;                       // region h of this query starts at buf + h * GS
  wave_lds_handoff();
  uint64_t raw[2], e[2];
  raw[0] = lane < n0 ? buf[lane] : 0ull;
  raw[1] = lane < n1 ? buf[GS + lane] : 0ull;
  e[0] = lane < n0 ? raw_to_key(raw[0]) : 0ull;
  e[1] = lane < n1 ? raw_to_key(raw[1]) : 0ull;
  int c = n0 + n1;
  uint64_t p = 1ull;                                   // one region nearly full, few entries overall: only rebalance
  if (c > k + SLACK) p = bisect_kth<2>(e, k, SLACK, c);
  int base = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const bool keep = e[t] >= p;
    const unsigned long long m = __ballot(keep);
    const int r = base + __popcll(m & below);         // rank among the survivors -> region r & 1, slot r >> 1
    if (keep) buf[(r & 1) * GS + (r >> 1)] = raw[t];
    base += __popcll(m);
  }
  wave_lds_handoff();
  if (p != 1ull) new_tau = ord2f((uint32_t)(p >> 32));
  return c;
}

template <int ABL, bool BR>
__global__ __launch_bounds__(256, 1) void memread_select32_kernel(const SelectArgs a) {
  __shared__ __attribute__((aligned(16))) float ktile[2][KT * KLD];
  __shared__ uint64_t cand[QT2 * 2 * REG2];           // [wave][h][j][REG2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int qslot = wave * QW2 + j;
  const int lrow = tid >> 3, lc = tid & 7;           // key-tile loader: row tid>>3, float4 columns lc + 8 jj

  long long t_begin = (long long)blockIdx.x * a.tiles_per_wg;
  const long long t_end = (t_begin + a.tiles_per_wg < a.total_tiles) ? t_begin + a.tiles_per_wg : a.total_tiles;
  const bool prof = a.dbg && blockIdx.x == 0;       // profiling builds only (MIVOS_MEMREAD_DBG)
  const unsigned long long clk0 = prof ? __builtin_readcyclecounter() : 0ull;
  unsigned long long clk_room = 0ull, clk_final = 0ull, clk_pro = 0ull, n_compact = 0ull;
  write_plan_header(a);

  while (t_begin < t_end) {
    const int stream = (int)(t_begin / a.tps);
    const int seg_lo = (int)(t_begin - (long long)stream * a.tps);
    int seg_hi = seg_lo + (int)(t_end - t_begin);
    if (seg_hi > a.tps) seg_hi = a.tps;
    const int nt = seg_hi - seg_lo;
    const int obj = stream / a.n_qtiles, qtile = stream - obj * a.n_qtiles;
    const int slot = (int)blockIdx.x - (int)(((long long)stream * a.tps) / a.tiles_per_wg);
    const int r0 = seg_lo * KT;
    const int r1 = ((long long)seg_hi * KT < a.n_mem) ? seg_hi * KT : (int)a.n_mem;
    const float *kbase = a.keys + (long long)obj * a.keys_ostride;

    __syncthreads();                                  // previous segment completely done with LDS

    // B operand: query j, scaled like prop_net.py:86, split hi/lo; qf[2 ks] = hi, qf[2 ks + 1] = lo of chunk (ks, h)
    f32x4_t qf[16];
    {
      const int q = qtile * QT2 + qslot;
      const float *qrow = a.qk + (long long)(q < a.n_q ? q : a.n_q - 1) * CK + 32 * h;
      const float d = sqrtf((float)CK);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float *src = qrow + 64 * (ks & 1) + 8 * (ks >> 1);
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(src), v1 = *reinterpret_cast<const f32x4_t *>(src + 4);
        half8_t hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (e < 4 ? v0[e & 3] : v1[e & 3]) / d;
          hi[e] = (_Float16)x;
          lo[e] = (_Float16)(x - (float)hi[e]);
        }
        qf[2 * ks] = __builtin_bit_cast(f32x4_t, hi);
        qf[2 * ks + 1] = __builtin_bit_cast(f32x4_t, lo);
      }
#pragma unroll
      for (int x = 0; x < 16; ++x) asm volatile("" : "+v"(qf[x]));      // finished before the key requests go out
    }

    // global -> register -> LDS staging, STAGE_DEPTH2 tiles in flight, requests in inline assembly with explicit vmcnt (see
    // memread_select_kernel)
    f32x4_t kr[STAGE_DEPTH2][4];
    auto gload = [&](f32x4_t (&krs)[4], int kb) {
      const int m = kb + lrow;                        // rows past the segment's end: its first row instead (never selected)
      const f32x4_t *src = reinterpret_cast<const f32x4_t *>(kbase + (long long)(m < r1 ? m : r0) * CK) + lc;
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:128\n\t"
                   "global_load_dwordx4 %2, %4, off offset:256\n\tglobal_load_dwordx4 %3, %4, off offset:384"
                   : "=&v"(krs[0]), "=&v"(krs[1]), "=&v"(krs[2]), "=&v"(krs[3]) : "v"(src) : "memory");
    };
    auto lds_store = [&](f32x4_t (&krs)[4], int buf) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4_t *>(&ktile[buf][lrow * KLD + 4 * (lc + 8 * jj)]) = krs[jj];
    };

    float my_tau = -INFINITY;
    uint64_t *const my_region = cand + ((wave * 2 + h) * QW2 + j) * REG2;
    const uint32_t region_lds = (uint32_t)(size_t)my_region;
    int my_cnt = 0;
    // Append path of one score (see memread_select_kernel): compare, index, LDS address, a store executed by the passing
    // lanes only, fill level.  `sc` is read straight from the PREVIOUS tile's accumulators while the matrix pipe works on the
    // current tile (two accumulator sets, even / odd tiles): no latch phase between tiles, and the previous tile's last MFMA
    // retired a whole tile ago.
    bool s_pass;
    float sc;
    uint32_t idx_base = 0u;                           // previous tile's base row + 4h: index of score r = idx_base + 8 (r >> 2) + (r & 3)
    auto slice_a = [&](const f32x16_t (&pv)[2], int r) { sc = pv[0][r] + pv[1][r]; s_pass = sc > my_tau; };
    auto slice_b = [&](int r) {
      const unsigned long long m = __ballot(s_pass);
      if (BR && m == 0ull) return;
      const uint32_t addr = region_lds + 8u * (uint32_t)my_cnt;
      const uint32_t idx = idx_base + (uint32_t)(8 * (r >> 2) + (r & 3));
      const uint32_t bits = __float_as_uint(sc);
      unsigned long long saved;
      asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write2_b32 %2, %3, %4 offset1:1\n\ts_mov_b64 exec, %0"
                   : "=&s"(saved) : "s"(m), "v"(addr), "v"(idx), "v"(bits) : "memory");
      my_cnt += s_pass ? 1 : 0;
    };
    auto compact_one = [&](int ql, bool force) {
      const int n0 = __builtin_amdgcn_readlane(my_cnt, ql), n1 = __builtin_amdgcn_readlane(my_cnt, ql + 32);
      if (force && n0 + n1 <= a.top_k + SLACK) return;
      float nt_tau = my_tau;
      const int c = compact_query2(cand + (wave * 2 * QW2 + ql) * REG2, n0, n1, a.top_k, lane, nt_tau);
      if (j == ql) { my_cnt = (c - h + 1) >> 1; my_tau = nt_tau; }
    };
    auto make_room = [&]() {
      const unsigned long long full = __ballot(my_cnt > REG2_TRIGGER);
      unsigned need = (unsigned)(full | (full >> 32));
      if (need) {
        const unsigned long long c0 = prof ? __builtin_readcyclecounter() : 0ull;
        while (need) {
          const int ql = __builtin_ctz(need);
          need &= need - 1;
          compact_one(ql, false);
          n_compact += prof ? 1 : 0;
        }
        if (prof) clk_room += __builtin_readcyclecounter() - c0;
      }
    };

    const unsigned long long cpro = prof ? __builtin_readcyclecounter() : 0ull;
    f32x4_t fa[16], fb[16];
    gload(kr[0], r0);
    gload(kr[1], r0 + KT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_store(kr[0], 0);
    lds_store(kr[1], 1);
#pragma unroll
    for (int d = 0; d < STAGE_DEPTH2; ++d) gload(kr[d], r0 + (2 + d) * KT);     // set d: tiles 2 + d, 2 + d + STAGE_DEPTH2, ...
    __syncthreads();
    {
      const float *arow = &ktile[0][j * KLD + 8 * h];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        fa[2 * ks] = *reinterpret_cast<const f32x4_t *>(arow + 16 * ks);
        fa[2 * ks + 1] = *reinterpret_cast<const f32x4_t *>(arow + 16 * ks + 4);
      }
    }
    __syncthreads();                                  // every wave holds tile 0 in registers: its LDS copy is dead
    if (prof) clk_pro += __builtin_readcyclecounter() - cpro;

    // One tile: MFMAs on fragment set F (tile t) into accumulator set `cur` while G receives tile t+1's fragments, tile
    // t+2 goes from its staging registers to LDS (over tile t's dead copy), tile t+4 is requested, and tile t-1's scores
    // (accumulator set `pv`) are selected.  Everything is dealt out between the MFMAs, pinned by scheduling barriers.
    f32x16_t accA[2], accB[2];
    auto tile_iter = [&](int t, f32x4_t (&F)[16], f32x4_t (&G)[16], f32x4_t (&krs)[4], f32x16_t (&cur)[2], const f32x16_t (&pv)[2], auto first) {
      constexpr bool SELECT = !decltype(first)::value && ABL != 1;
      const float *nrow = &ktile[(t + 1) & 1][j * KLD + 8 * h];   // tile t+1
      idx_base = (uint32_t)(r0 + (t - 1) * KT + 4 * h);
#pragma unroll
      for (int r = 0; r < 16; ++r) { cur[0][r] = 0.f; cur[1][r] = 0.f; }
      // staging: tile t+2 (requested STAGE_DEPTH2 iterations ago) goes to LDS one 16-byte piece per k-step 0..3, over tile t's
      // dead copy; then its registers take the request for tile t + 2 + STAGE_DEPTH2.  Both are unconditional: past the end of
      // the segment the rows are clamped and the LDS copy is never read.
      float *ldst = &ktile[t & 1][lrow * KLD + 4 * lc];
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (STAGE_DEPTH2 - 1)) : "memory");
#define MIVOS_SB __builtin_amdgcn_sched_barrier(0);
#define MIVOS_HF(N, A, B) cur[(N) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, A), __builtin_bit_cast(half8_t, B), cur[(N) & 1], 0, 0, 0); MIVOS_SB
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (SELECT && ABL == 0 && (ks == 0 || ks == 4)) make_room();      // before the appends of scores 0-7 / 8-15
        G[2 * ks] = *reinterpret_cast<const f32x4_t *>(nrow + 16 * ks); MIVOS_SB
        MIVOS_HF(3 * ks, F[2 * ks + 1], qf[2 * ks])             // lo * hi
        if (SELECT) slice_a(pv, 2 * ks);
        MIVOS_SB
        if (SELECT) slice_b(2 * ks);
        MIVOS_SB
        MIVOS_HF(3 * ks + 1, F[2 * ks], qf[2 * ks + 1])         // hi * lo
        G[2 * ks + 1] = *reinterpret_cast<const f32x4_t *>(nrow + 16 * ks + 4); MIVOS_SB
        if (SELECT) slice_a(pv, 2 * ks + 1);
        MIVOS_SB
        MIVOS_HF(3 * ks + 2, F[2 * ks], qf[2 * ks])             // hi * hi
        if (SELECT) slice_b(2 * ks + 1);
        MIVOS_SB
        if (ks < 4) { *reinterpret_cast<f32x4_t *>(ldst + 32 * ks) = krs[ks]; MIVOS_SB }
        if (ks == 3) { gload(krs, r0 + (t + 2 + STAGE_DEPTH2) * KT); MIVOS_SB }
      }
#undef MIVOS_HF
#undef MIVOS_SB
      // ROCm 7.2's hazard recognizer leaves too few wait states between a v_mfma_f32_32x32x16_f16 and the first read of its
      // result by the vector ALU when a branch or barrier lies between them (measured: stale last rows of the tile, i.e.
      // dropped candidates); the operands tie every later use of these accumulators behind the wait
      asm volatile("s_nop 15" : "+a"(cur[0]), "+a"(cur[1]));
      if (ABL == 1) {                           // keep the MFMA results alive without selecting
        const float sum = (cur[0][0] + cur[0][5]) + (cur[1][10] + cur[1][15]);
        if (sum == 123.456f) a.lists[0] = 1ull;
      }
      __syncthreads();
    };
    tile_iter(0, fa, fb, kr[0], accA, accB, std::true_type{});
    if (nt > 1) tile_iter(1, fb, fa, kr[1], accB, accA, std::false_type{});
    if (nt > 2) tile_iter(2, fa, fb, kr[2], accA, accB, std::false_type{});
    if (nt > 3) tile_iter(3, fb, fa, kr[3], accB, accA, std::false_type{});
    for (int t = 4; t < nt; t += 4) {
      tile_iter(t, fa, fb, kr[0], accA, accB, std::false_type{});
      if (t + 1 < nt) tile_iter(t + 1, fb, fa, kr[1], accB, accA, std::false_type{});
      if (t + 2 < nt) tile_iter(t + 2, fa, fb, kr[2], accA, accB, std::false_type{});
      if (t + 3 < nt) tile_iter(t + 3, fb, fa, kr[3], accB, accA, std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests past the end of the segment: their registers are reused
    // drain the pipeline: select on the last tile (the only one that can hold rows past the end of the memory)
    const unsigned long long cfin = prof ? __builtin_readcyclecounter() : 0ull;
    auto drain = [&](const f32x16_t (&last)[2]) {
      const int pb = r0 + (nt - 1) * KT;
      idx_base = (uint32_t)(pb + 4 * h);
      f32x16_t masked[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = pb + 4 * h + 8 * (r >> 2) + (r & 3) < r1;
        masked[0][r] = ok ? last[0][r] : -INFINITY;
        masked[1][r] = ok ? last[1][r] : -INFINITY;
      }
      make_room();
#pragma unroll
      for (int r = 0; r < 8; ++r) { slice_a(masked, r); slice_b(r); }
      make_room();
#pragma unroll
      for (int r = 8; r < 16; ++r) { slice_a(masked, r); slice_b(r); }
    };
    if (ABL != 1) {
      if ((nt - 1) & 1) drain(accB);
      else drain(accA);
    }

    // this segment's candidate lists: for each of the wave's 32 queries between min(n, k) and k + SLACK entries
    for (int ql = 0; ql < QW2; ++ql) {
      compact_one(ql, true);
      wave_lds_handoff();
      const int n0 = __builtin_amdgcn_readlane(my_cnt, ql), n1 = __builtin_amdgcn_readlane(my_cnt, ql + 32);
      const int s = wave * QW2 + ql;
      const uint64_t *src = cand + (wave * 2 * QW2 + ql) * REG2;   // region h at src + h * QW2 * REG2
      uint64_t *dst = a.lists + (((long long)stream * a.slots + slot) * QT2 + s) * a.L;
      if (lane < n0) dst[lane] = raw_to_key(src[lane]);
      if (lane < n1) dst[n0 + lane] = raw_to_key(src[QW2 * REG2 + lane]);
      for (int i = n0 + n1 + lane; i < a.L; i += 64) dst[i] = 0ull;
    }
    if (prof) clk_final += __builtin_readcyclecounter() - cfin;
    t_begin += nt;
  }
  if (prof && tid == 0) {
    a.dbg[0] = __builtin_readcyclecounter() - clk0;
    a.dbg[1] = (unsigned long long)(t_end - (long long)blockIdx.x * a.tiles_per_wg);
    a.dbg[2] = clk_room; a.dbg[3] = n_compact; a.dbg[4] = clk_final; a.dbg[5] = clk_pro;
  }
}

// ---- the 256-queries-per-workgroup variant for very long memories -------------------------------------------------------
// memread_select32_kernel is bound by the L2 -> CU key stream: every 128-query workgroup re-reads its object's whole key bank
// (80 GB of L2 requests per launch at 1080p, T = 100).  More queries per key byte need more candidate storage than LDS has
// (256 queries x 2 regions x 61 entries x 8 B = 250 KB), so here the lane-private candidate regions live in GLOBAL scratch:
// once the thresholds have converged (a few hundred positions into a stream of 10^5 - 10^6) almost no score passes, the
// append is a rare exec-masked global store and a compaction a rare round trip to L2.  LDS then only holds the two key tiles.
//   8 waves x 32 queries per workgroup (two waves per SIMD: one fills the matrix pipe while the other reads its fragments and
//   selects); 256 VGPRs per wave, so one fragment set (read at the start of a tile), two accumulator sets (tile t - 1 is selected
//   between the MFMAs of tile t, like in memread_select32_kernel; one chain of 24 MFMAs per tile) and four key tiles in flight
//   (explicit vmcnt as above); LDS is a ring of eight tiles and the workgroup synchronises once per four tiles (see tile_iter).
//   A wave's own global stores are ordered before its later loads by a workgroup-scope fence (same CU, same L1) and the
//   reads bypass L1 (agent-scope loads) for good measure.  The explicit `s_waitcnt vmcnt(N)` of the key pipeline stays
//   correct with stores in flight: they only add to the count, and the count cannot fall to N before the oldest loads landed.
constexpr int NW3 = 8, QT3 = 256, REG3 = 61, REG3_TRIGGER = REG3 - 1 - 16;   // (room for a whole tile's 16 appends per lane is made once per tile)
constexpr int GROUP3 = 4, NBUF3 = 2 * GROUP3;   // key tiles per barrier; LDS ring of two groups (LDS holds nothing else here: 132 KB)
constexpr long long CAND3_PER_WG = (long long)NW3 * 2 * 32 * REG3;      // candidate entries (8 bytes each) per workgroup

// Two candidate entries read back from the global scratch (agent-scope loads: past the CU's L1).  Inline assembly ON PURPOSE:
// a load the compiler knows of makes it guard every later write of the destination registers - which it reuses for the
// accumulators - with an `s_waitcnt vmcnt(0)` at the join behind the (rarely taken) compaction branch, i.e. at the first MFMA
// of EVERY tile, where that wait also drains the key requests just issued (the tile pipeline then runs at one memory latency
// per tile: the first two versions of this kernel, 17 - 18 ms at 1080p T = 100).  These requests are invisible to the compiler
// like the key requests; the wait is written out and tied to the results.
__device__ __forceinline__ void gload_entries(const uint64_t *p0, const uint64_t *p1, uint64_t &v0, uint64_t &v1) {
  asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
}

__device__ __forceinline__ int compact_query_g(uint64_t *buf, int n0, int n1, int k, int lane, float &new_tau) {
  constexpr int GS = 32 * REG3;                        // region h of this query starts at buf + h * GS
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const int li = lane < REG3 ? lane : REG3 - 1;       // unconditional loads (stale slots of the same region past its fill level)
  uint64_t raw[2], e[2];
  gload_entries(buf + li, buf + GS + li, raw[0], raw[1]);
  const uint64_t k0 = raw_to_key(raw[0]), k1 = raw_to_key(raw[1]);
  e[0] = lane < n0 ? k0 : 0ull;
  e[1] = lane < n1 ? k1 : 0ull;
  int c = n0 + n1;
  uint64_t p = 1ull;                                   // one region nearly full, few entries overall: only rebalance
  if (c > k + SLACK) p = bisect_kth<2>(e, k, SLACK, c);
  int base = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const bool keep = e[t] >= p;
    const unsigned long long m = __ballot(keep);
    const int r = base + __popcll(m & below);         // rank among the survivors -> region r & 1, slot r >> 1
    if (keep) buf[(r & 1) * GS + (r >> 1)] = raw[t];
    base += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (p != 1ull) new_tau = ord2f((uint32_t)(p >> 32));
  return c;
}

// HIFIRST (round 3, last experiment; off unless mivos_memory_read_set_hifirst(1)): a tile is first multiplied with the hi halves only
// (8 MFMAs and 8 fragment reads instead of 24 and 16).  The two dropped products are bounded: x = hi + lo with |lo| <= 2^-11 |x|
// (half an fp16 ulp; 2^-25 absolute below the normal range), so |k.q - kh.qh| <= 2^-10 (1 + 2^-10) sum |k_c| |q_c| <= 2^-10
// (1 + 2^-10) |k| |q| (Cauchy-Schwarz).  With eps = 1.25 x 2^-10 x max_rows |k| x |q| + 1e-6 (|k| + |q|) (the margin covers the fp32
// accumulation of 128 terms; CPU check of the bound: tests/test_host_logic.py::test_hi_first_bound_model) a wave tile in which every
// lane's hi-only scores stay <= threshold - eps cannot hold a candidate - the exact score is <= the threshold and an equal score
// loses its tie to the earlier positions - and is skipped; otherwise the two lo products are added to the same accumulators
// (16 more MFMAs on fragments read then) and the tile is selected on exactly as before.  After convergence the threshold is the
// rank-k score of 10^5 - 10^6 positions and eps ~ 1 % of the score spread: a few per cent of the wave tiles take the second pass.
template <bool HIFIRST>
__global__ __launch_bounds__(512, 1) void memread_select256_kernel(const SelectArgs a) {
  __shared__ __attribute__((aligned(16))) float ktile[NBUF3][KT * KLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int qslot = wave * 32 + j;
  const int lrow = tid >> 4, lc = tid & 15;          // key-tile loader: row tid >> 4, float4 columns lc and lc + 16

  long long t_begin = (long long)blockIdx.x * a.tiles_per_wg;
  const long long t_end = (t_begin + a.tiles_per_wg < a.total_tiles) ? t_begin + a.tiles_per_wg : a.total_tiles;
  write_plan_header(a);
  uint64_t *const wave_regions = a.cand + ((long long)blockIdx.x * NW3 + wave) * (2 * 32 * REG3);   // [h][j][REG3]
  uint64_t *const my_region = wave_regions + (h * 32 + j) * REG3;

  while (t_begin < t_end) {
    const int stream = (int)(t_begin / a.tps);
    const int seg_lo = (int)(t_begin - (long long)stream * a.tps);
    int seg_hi = seg_lo + (int)(t_end - t_begin);
    if (seg_hi > a.tps) seg_hi = a.tps;
    const int nt = seg_hi - seg_lo;
    const int obj = stream / a.n_qtiles, qtile = stream - obj * a.n_qtiles;
    const int slot = (int)blockIdx.x - (int)(((long long)stream * a.tps) / a.tiles_per_wg);
    const int r0 = seg_lo * KT;
    const int r1 = ((long long)seg_hi * KT < a.n_mem) ? seg_hi * KT : (int)a.n_mem;
    const float *kbase = a.keys + (long long)obj * a.keys_ostride;

    __syncthreads();                                  // previous segment completely done with LDS

    f32x4_t qf[16];                                   // B operand: query j, scaled like prop_net.py:86, split hi/lo (see memread_select32_kernel)
    float my_eps = 0.f;                               // HIFIRST: bound of the two dropped products for this lane's query against any key of the object
    {
      const int q = qtile * QT3 + qslot;
      const float *qrow = a.qk + (long long)(q < a.n_q ? q : a.n_q - 1) * CK + 32 * h;
      const float d = sqrtf((float)CK);
      float qn2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float *src = qrow + 64 * (ks & 1) + 8 * (ks >> 1);
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(src), v1 = *reinterpret_cast<const f32x4_t *>(src + 4);
        half8_t hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (e < 4 ? v0[e & 3] : v1[e & 3]) / d;
          hi[e] = (_Float16)x;
          lo[e] = (_Float16)(x - (float)hi[e]);
          if (HIFIRST) { const float xs = (float)hi[e] + (float)lo[e]; qn2 += xs * xs; }
        }
        qf[2 * ks] = __builtin_bit_cast(f32x4_t, hi);
        qf[2 * ks + 1] = __builtin_bit_cast(f32x4_t, lo);
      }
      if (HIFIRST) {
        qn2 += __shfl_xor(qn2, 32);                   // the other half of the query's channels (lane j + 32 h)
        const float qn = sqrtf(qn2) * 1.0001f, kn = sqrtf(a.kmax2[obj]) * 1.0001f;
        my_eps = kn * qn * (1.25f / 1024.f) + 1e-6f * (kn + qn);
        if (!(my_eps < INFINITY)) my_eps = INFINITY;  // (NaN / inf norms: every tile takes the exact pass)
      }
#pragma unroll
      for (int x = 0; x < 16; ++x) asm volatile("" : "+v"(qf[x]));      // finished before the key requests go out
    }

    f32x4_t kr[GROUP3][2];
    auto gload = [&](f32x4_t (&krs)[2], int kb) {
      const int m = kb + lrow;                        // rows past the segment's end: its first row instead (never selected)
      const f32x4_t *src = reinterpret_cast<const f32x4_t *>(kbase + (long long)(m < r1 ? m : r0) * CK) + lc;
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:256"
                   : "=&v"(krs[0]), "=&v"(krs[1]) : "v"(src) : "memory");
    };
    auto lds_store = [&](f32x4_t (&krs)[2], int buf) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) *reinterpret_cast<f32x4_t *>(&ktile[buf][lrow * KLD + 4 * (lc + 16 * jj)]) = krs[jj];
    };

    float my_tau = -INFINITY, my_tau_lo = -INFINITY;  // my_tau_lo = my_tau - my_eps: what a hi-only score has to exceed to matter
    int my_cnt = 0;
    auto compact_one = [&](int ql, bool force) {
      const int n0 = __builtin_amdgcn_readlane(my_cnt, ql), n1 = __builtin_amdgcn_readlane(my_cnt, ql + 32);
      if (force && n0 + n1 <= a.top_k + SLACK) return;
      float nt_tau = my_tau;
      const int c = compact_query_g(wave_regions + ql * REG3, n0, n1, a.top_k, lane, nt_tau);
      if (j == ql) {
        my_cnt = (c - h + 1) >> 1;
        my_tau = nt_tau;
        if (HIFIRST) my_tau_lo = (my_eps < INFINITY) ? nt_tau - my_eps : -INFINITY;
      }
    };
    auto make_room = [&]() {
      const unsigned long long full = __ballot(my_cnt > REG3_TRIGGER);
      unsigned need = (unsigned)(full | (full >> 32));
      while (need) {
        const int ql = __builtin_ctz(need);
        need &= need - 1;
        compact_one(ql, false);
      }
    };

    // prologue: the first group of tiles into LDS slots 0 .. GROUP3 - 1, the second group requested (tile t -> register set t % GROUP3)
#pragma unroll
    for (int d = 0; d < GROUP3; ++d) gload(kr[d], r0 + d * KT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < GROUP3; ++d) lds_store(kr[d], d);
#pragma unroll
    for (int d = 0; d < GROUP3; ++d) gload(kr[d], r0 + (GROUP3 + d) * KT);
    __syncthreads();

    // Append path of one score of the PREVIOUS tile (accumulator set `pv`), dealt out between the MFMAs of the running tile like in
    // memread_select32_kernel: the fp16 matrix pipe leaves the vector ALU free, so a wave selects tile t - 1 while it multiplies
    // tile t (with the workgroup barrier keeping the 8 waves in phase, nothing else would overlap the two).
    uint32_t idx_base = 0u;                           // previous tile's base row + 4h: index of score r = idx_base + 8 (r >> 2) + (r & 3)
    // Scores are tested FOUR at a time: once the thresholds have converged (a few hundred positions into a stream of 10^5 - 10^6)
    // no score passes in almost every wave tile, so the common path per four scores is max3 + max + compare + one wave-uniform
    // branch (3 vector instructions instead of 8: the three products of a k-step go into ONE accumulator chain - measured, 1, 2
    // or 3 chains issue alike, DESIGN "What bounds it now" - so there is no add of two partial sums either); the four scores are
    // looked at one by one only behind that branch.
    auto slice4 = [&](const f32x16_t &pv, int g4) {
      float m4;                                        // (fmaxf would canonicalise every MFMA result first: two more instructions)
      asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(m4) : "v"(pv[4 * g4]), "v"(pv[4 * g4 + 1]), "v"(pv[4 * g4 + 2]), "v"(pv[4 * g4 + 3]));
      if (__builtin_expect(__ballot(m4 > my_tau) != 0ull, 0)) {
#pragma unroll
        for (int r = 4 * g4; r < 4 * g4 + 4; ++r) {
          const float sc = pv[r];
          const bool pass = sc > my_tau;
          if (pass) my_region[my_cnt] = ((uint64_t)__float_as_uint(sc) << 32) | (uint64_t)(idx_base + (uint32_t)(8 * (r >> 2) + (r & 3)));
          my_cnt += pass ? 1 : 0;
        }
      }
    };
    // One tile.  The workgroup synchronises once per GROUP of four tiles, not per tile: LDS is a ring of two groups, during group g
    // every wave writes its rows of the tiles of group g + 1 (requested a group ago; their registers take the requests for group
    // g + 2) into the other half of the ring, and the barrier at the end of the group is the only point where "those tiles are
    // complete" and "nobody reads this half any more" have to hold.  Inside a group the eight waves drift apart, so that one
    // wave's fragment reads (16 ds_read_b128 at the start of a tile; 128 KB per tile for the workgroup = 1 000 LDS cycles) overlap
    // the other waves' MFMAs (1 536 matrix-pipe cycles per tile and SIMD) instead of alternating with them in lockstep, and a
    // wave that has to compact a query stalls the others only if it is still behind at the end of the group.
    // Tile t: its fragments, 24 MFMAs into `cur` with the selection of tile t - 1 (`pv`) between them.
    f32x16_t accA, accB;
    // HIFIRST: tile t is multiplied with the hi halves, tested against my_tau_lo and - rarely - completed and selected on, all in
    // its own iteration (the other wave of the SIMD covers the wait for the accumulators; no second accumulator set).
    auto tile_iter_hf = [&](int t, f32x4_t (&krs)[2]) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (GROUP3 - 1)) : "memory");
      lds_store(krs, (t + GROUP3) & (NBUF3 - 1));
      gload(krs, r0 + (t + 2 * GROUP3) * KT);
      const float *arow = &ktile[t & (NBUF3 - 1)][j * KLD + 8 * h];
      f32x4_t fh[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) fh[ks] = *reinterpret_cast<const f32x4_t *>(arow + 16 * ks);
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fh[ks]), __builtin_bit_cast(half8_t, qf[2 * ks]), acc, 0, 0, 0);   // hi * hi
      asm volatile("s_nop 15" : "+v"(acc));             // (MFMA results read by the vector ALU: the hazard guard of the other kernels)
      const int pb = r0 + t * KT;
      if (pb + KT > r1) {                               // the stream's last tile: rows past the end of the memory never pass
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (pb + 4 * h + 8 * (r >> 2) + (r & 3) >= r1) acc[r] = -INFINITY;
      }
      float m16;
      {
        float ma, mb, mc, md;
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(ma) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(mb) : "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(mc) : "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]));
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(md) : "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]));
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(m16) : "v"(ma), "v"(mb), "v"(mc), "v"(md));
      }
      if (__builtin_expect(__ballot(m16 > my_tau_lo) != 0ull, 0)) {
        f32x4_t fl[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fl[ks] = *reinterpret_cast<const f32x4_t *>(arow + 16 * ks + 4);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fl[ks]), __builtin_bit_cast(half8_t, qf[2 * ks]), acc, 0, 0, 0);       // lo * hi
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fh[ks]), __builtin_bit_cast(half8_t, qf[2 * ks + 1]), acc, 0, 0, 0);   // hi * lo
        }
        asm volatile("s_nop 15" : "+v"(acc));
        idx_base = (uint32_t)(pb + 4 * h);
        make_room();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) slice4(acc, g4);
      }
      if (((t + 1) & (GROUP3 - 1)) == 0) __syncthreads();
    };
    auto tile_iter = [&](int t, f32x4_t (&krs)[2], f32x16_t &cur, const f32x16_t &pv, auto first) {
      constexpr bool SELECT = !decltype(first)::value;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (GROUP3 - 1)) : "memory");     // tile t + GROUP3 has landed (requests complete in order)
      lds_store(krs, (t + GROUP3) & (NBUF3 - 1));
      gload(krs, r0 + (t + 2 * GROUP3) * KT);
      const float *arow = &ktile[t & (NBUF3 - 1)][j * KLD + 8 * h];
      f32x4_t fa[16];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        fa[2 * ks] = *reinterpret_cast<const f32x4_t *>(arow + 16 * ks);
        fa[2 * ks + 1] = *reinterpret_cast<const f32x4_t *>(arow + 16 * ks + 4);
      }
      idx_base = (uint32_t)(r0 + (t - 1) * KT + 4 * h);
#pragma unroll
      for (int r = 0; r < 16; ++r) cur[r] = 0.f;
      if (SELECT) make_room();                                 // before the (up to 16 per lane) appends of the previous tile's scores
#define MIVOS_SB __builtin_amdgcn_sched_barrier(0);
#define MIVOS_HF(A, B) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, A), __builtin_bit_cast(half8_t, B), cur, 0, 0, 0); MIVOS_SB
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        MIVOS_HF(fa[2 * ks + 1], qf[2 * ks])                   // lo * hi
        if (SELECT && (ks & 1) == 0) slice4(pv, ks >> 1);      // scores 4 (ks / 2) .. + 3 of the previous tile
        MIVOS_SB
        MIVOS_HF(fa[2 * ks], qf[2 * ks + 1])                   // hi * lo
        MIVOS_HF(fa[2 * ks], qf[2 * ks])                       // hi * hi
      }
#undef MIVOS_HF
#undef MIVOS_SB
      // (the hazard guard of memread_select32_kernel: MFMA results read by the vector ALU behind a branch / barrier)
      asm volatile("s_nop 15" : "+v"(cur));
      if (((t + 1) & (GROUP3 - 1)) == 0) __syncthreads();
    };
    static_assert(GROUP3 == 4, "the tile loop is written out for four register sets");
    if (HIFIRST) {
      for (int t = 0; t < nt; t += 4) {
        tile_iter_hf(t, kr[0]);
        if (t + 1 < nt) tile_iter_hf(t + 1, kr[1]);
        if (t + 2 < nt) tile_iter_hf(t + 2, kr[2]);
        if (t + 3 < nt) tile_iter_hf(t + 3, kr[3]);
      }
    } else {
    tile_iter(0, kr[0], accA, accA, std::true_type{});            // (no previous tile to select on: `pv` is not read)
    if (nt > 1) tile_iter(1, kr[1], accB, accA, std::false_type{});
    if (nt > 2) tile_iter(2, kr[2], accA, accB, std::false_type{});
    if (nt > 3) tile_iter(3, kr[3], accB, accA, std::false_type{});
    for (int t = 4; t < nt; t += 4) {
      tile_iter(t, kr[0], accA, accB, std::false_type{});
      if (t + 1 < nt) tile_iter(t + 1, kr[1], accB, accA, std::false_type{});
      if (t + 2 < nt) tile_iter(t + 2, kr[2], accA, accB, std::false_type{});
      if (t + 3 < nt) tile_iter(t + 3, kr[3], accB, accA, std::false_type{});
    }
    }
    // drain the pipeline: select on the last tile (the only one that can hold rows past the end of the memory)
    auto drain = [&](const f32x16_t &last) {
      const int pb = r0 + (nt - 1) * KT;
      idx_base = (uint32_t)(pb + 4 * h);
      f32x16_t masked;
#pragma unroll
      for (int r = 0; r < 16; ++r) masked[r] = (pb + 4 * h + 8 * (r >> 2) + (r & 3) < r1) ? last[r] : -INFINITY;
      make_room();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) slice4(masked, g4);
    };
    if (!HIFIRST) {
      if ((nt - 1) & 1) drain(accB);
      else drain(accA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests past the end of the segment: their registers are reused

    // this segment's candidate lists: for each of the wave's 32 queries between min(n, k) and k + SLACK entries
    for (int ql = 0; ql < 32; ++ql) {
      compact_one(ql, true);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const int n0 = __builtin_amdgcn_readlane(my_cnt, ql), n1 = __builtin_amdgcn_readlane(my_cnt, ql + 32);
      const int s = wave * 32 + ql;
      const uint64_t *src = wave_regions + ql * REG3;               // region h at src + h * 32 * REG3
      uint64_t *dst = a.lists + (((long long)stream * a.slots + slot) * QT3 + s) * a.L;
      const int li = lane < REG3 ? lane : REG3 - 1;
      uint64_t e0, e1;
      gload_entries(src + li, src + 32 * REG3 + li, e0, e1);
      if (lane < n0) dst[lane] = raw_to_key(e0);
      if (lane < n1) dst[n0 + lane] = raw_to_key(e1);
      for (int i = n0 + n1 + lane; i < a.L; i += 64) dst[i] = 0ull;
    }
    t_begin += nt;
  }
}

// optional SH32 outputs of the readout (zero-bordered activation buffers of the LDS-DMA convolutions): raw and relu(raw),
// addressed as image `obj`, pixel (q / q_width, q % q_width); strides in floats
struct ShOut {
  float *raw, *relu;
  long long ns, rs, ps;
  int q_width;
};

// one single-wave workgroup per (object, query); __syncthreads() on a 64-thread block is just the LDS ordering fence
// between the phases
template <bool INDICES, int FIN_EPL>   // FIN_EPL: merged candidates per lane (64 FIN_EPL >= segments x L)
__global__ __launch_bounds__(64) void memread_finalize_kernel(const int *__restrict__ header,
                                                             const float *__restrict__ values, long long values_ostride,
                                                             float *__restrict__ out, long long out_ostride,
                                                             long long out_pstride, int32_t *__restrict__ idx_out,
                                                             float *__restrict__ w_out, int n_q, int top_k, ShOut sh) {
  const uint64_t *__restrict__ lists = reinterpret_cast<const uint64_t *>(header + HEADER_BYTES / 4);
  const int qt = header[0], n_qtiles = header[1], tps = header[2], tiles_per_wg = header[3], slots = header[4], L = header[5];
  __shared__ uint64_t sel[MAX_TOPK];
  __shared__ float wv[MAX_TOPK];
  __shared__ uint32_t oi[MAX_TOPK];
  __shared__ float ow[MAX_TOPK];
  const int lane = threadIdx.x;
  const int q = blockIdx.x, obj = blockIdx.y;      // (an XCD-contiguous query order was measured in round 4: 208 -> 201 MB of fabric reads, 81 -> 84 us; not kept)
  const int stream = obj * n_qtiles + q / qt, qs = q % qt;
  // defined contents whatever the lists hold (ablation builds leave them incomplete): position 0, weight 0
  sel[lane] = pack_cand(-INFINITY, 0u); oi[lane] = 0u; ow[lane] = 0.f; wv[lane] = 0.f;
  __syncthreads();
  // the segments of this stream: workgroups w_first .. w_last of the select kernel
  const int w_first = (int)(((long long)stream * tps) / tiles_per_wg);
  const int w_last = (int)((((long long)stream + 1) * tps - 1) / tiles_per_wg);
  int n = (w_last - w_first + 1) * L;
  if (n > 64 * FIN_EPL) n = 64 * FIN_EPL;            // (never: the host sizes FIN_EPL for the larger of the two plans)
  uint64_t e[FIN_EPL];
#pragma unroll
  for (int t = 0; t < FIN_EPL; ++t) {
    const int i = lane + 64 * t;
    uint64_t v = 0ull;
    if (i < n) {
      const int sl = i / L, k = i - sl * L;
      v = lists[(((long long)stream * slots + sl) * qt + qs) * L + k];
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
  if (out) {
    float *o = out + (long long)obj * out_ostride + (long long)q * out_pstride + 4 * lane;
    *reinterpret_cast<f32x4 *>(o) = a0;
    *reinterpret_cast<f32x4 *>(o + 256) = a1;
  }
  if (sh.raw || sh.relu) {      // the decoder's first convolutions read the readout pre-split (and through ReLU): no pack pass
    const int qy = q / sh.q_width, qx = q - qy * sh.q_width;
    const long long pix = (long long)obj * sh.ns + (long long)qy * sh.rs + (long long)qx * sh.ps;
    if (sh.raw) { store_sh32x4(sh.raw, pix, 4 * lane, a0); store_sh32x4(sh.raw, pix, 256 + 4 * lane, a1); }
    if (sh.relu) {
      store_sh32x4(sh.relu, pix, 4 * lane, f32x4{fmaxf(a0.x, 0.f), fmaxf(a0.y, 0.f), fmaxf(a0.z, 0.f), fmaxf(a0.w, 0.f)});
      store_sh32x4(sh.relu, pix, 256 + 4 * lane, f32x4{fmaxf(a1.x, 0.f), fmaxf(a1.y, 0.f), fmaxf(a1.z, 0.f), fmaxf(a1.w, 0.f)});
    }
  }
}

// fp32 key rows [n_obj][n_rows][128] -> the pre-split rows the F16 select kernel streams (same 512 bytes per row): thread
// (row, t = 4 ks + b) converts channels 32 ks + 8 b .. +7 = floats 8 t .. 8 t + 7 into hi[8] | lo[8] at half offset 64 b + 16 ks.
__global__ __launch_bounds__(256) void split_keys_kernel(const float *__restrict__ src, long long src_ostride, float *__restrict__ dst,
                                                         long long dst_ostride, long long n_rows) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = i >> 4;
  if (row >= n_rows) return;
  const int t = (int)(i & 15), ks = t >> 2, b = t & 3;
  const float *p = src + (long long)blockIdx.y * src_ostride + row * CK + 8 * t;
  const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(p), v1 = *reinterpret_cast<const f32x4_t *>(p + 4);
  half8_t hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? v0[e & 3] : v1[e & 3];
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
  float *q = dst + (long long)blockIdx.y * dst_ostride + row * CK + 32 * b + 8 * ks;
  *reinterpret_cast<f32x4_t *>(q) = __builtin_bit_cast(f32x4_t, hi);
  *reinterpret_cast<f32x4_t *>(q + 4) = __builtin_bit_cast(f32x4_t, lo);
}

// Largest squared norm of a (pre-split) key row per object, for the hi-first select kernel's bound: thread (row, chunk) sums
// (hi + lo)^2 over its 8 channels, the 16 chunks of a row are added up across lanes, the block's maximum goes out through
// one atomic per wave (non-negative floats order like their bit patterns).  `out` must be zero before the launch.
__global__ __launch_bounds__(256) void key_norm2_max_kernel(const float *__restrict__ split, long long ostride, long long n_rows,
                                                            float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = i >> 4;
  float s2 = 0.f;
  if (row < n_rows) {
    const float *p = split + (long long)blockIdx.y * ostride + row * CK + 8 * (i & 15);
    const half8_t hi = __builtin_bit_cast(half8_t, *reinterpret_cast<const f32x4_t *>(p));
    const half8_t lo = __builtin_bit_cast(half8_t, *reinterpret_cast<const f32x4_t *>(p + 4));
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float x = (float)hi[e] + (float)lo[e]; s2 += x * x; }
  }
  s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 8);      // the row's 16 chunks
  s2 = fmaxf(s2, __shfl_xor(s2, 16)); s2 = fmaxf(s2, __shfl_xor(s2, 32));                                    // the wave's 4 rows
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int *>(out) + blockIdx.y, __float_as_uint(s2));
}

// ---- host side --------------------------------------------------------------------------------------------------
struct Plan {
  int qt, n_qtiles, streams, tps, n_wg, tiles_per_wg, slots, L;
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

// Workgroups of the persistent select kernels: 0 = one per CU (the kernel then owns the chip for its whole duration: 155 KB of LDS per
// workgroup leave no room for anybody else's); n > 0 = at most n - set by callers that keep several launch streams busy on the GPU
// (mivos_memory_read_set_workgroups; ops.chip_share sets CUs / streams), so that another clip's kernels run beside a select launch.
static std::atomic<int> g_select_wgs{0};

static Plan make_plan(int n_obj, long long n_mem, int n_q, int top_k, int qt) {
  Plan p;
  p.qt = qt;
  p.n_qtiles = cdiv(n_q, qt);
  p.streams = n_obj * p.n_qtiles;
  p.tps = cdiv(n_mem, KT);
  p.total = (long long)p.streams * p.tps;
  static const int forced_env = getenv("MIVOS_MEMREAD_WGS") ? atoi(getenv("MIVOS_MEMREAD_WGS")) : 0;   // tuning only
  const int forced = g_select_wgs.load(std::memory_order_relaxed) > 0 ? g_select_wgs.load(std::memory_order_relaxed) : forced_env;
  long long n_wg = forced > 0 ? (forced < compute_units() ? forced : compute_units()) : compute_units();
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

// which select kernel runs: the fp32 kernel and the 16-queries-per-wave fp16 kernel cut the queries into tiles of QT = 64,
// the 32-queries-per-wave fp16 kernel (long memories) into tiles of QT2 = 128
static std::atomic<long long> g_q128_min{-1};     // memory positions from which the 128-query plan is used (tests / tuning can move it)
static long long q128_min() {
  long long v = g_q128_min.load(std::memory_order_relaxed);
  if (v < 0) {
    // measured (profiles/r02g_memread_microbench.txt): 1080p, 3 objects: 20 frames (163 k positions) 4.75 ms with 64 queries
    // per workgroup vs 5.1 ms with 128 (smaller candidate regions, more compactions); 100 frames (816 k) 21.9 vs 18.9 ms
    v = getenv("MIVOS_MEMREAD_Q128_MIN") ? atoll(getenv("MIVOS_MEMREAD_Q128_MIN")) : 400000;
    g_q128_min.store(v, std::memory_order_relaxed);
  }
  return v;
}
#include <hip/hip_version.h>
constexpr bool Q256_TOOLCHAIN_VALIDATED = (HIP_VERSION_MAJOR == 7 && HIP_VERSION_MINOR == 2);
static std::atomic<long long> g_q256_min{-1};     // ... and from which the 256-query plan (candidate regions in global scratch) takes over
static long long q256_min() {
  long long v = g_q256_min.load(std::memory_order_relaxed);
  if (v < 0) {
    // measured (profiles/r03h_memread_depth_sweep.txt; ms per launch with 64 / 128 / 256 queries per workgroup): 1080p, 3 objects,
    // 20 frames (163 k positions) 4.69 / 4.94 / 4.70, 30 frames 6.77 / 6.64 / 6.25, 50 frames 10.9 / 10.0 / 9.24, 200 frames
    // 41.8 / 34.5 / 28.3; 480p, 5 objects, 100 frames (162 k positions) 2.16 / 2.47 / 2.51
    v = getenv("MIVOS_MEMREAD_Q256_MIN") ? atoll(getenv("MIVOS_MEMREAD_Q256_MIN")) : 200000;
    // The 256-query kernel (and the other two, for their key prefetch) issues global loads from inline assembly and writes their
    // s_waitcnt vmcnt(N) by hand: correct only as long as the compiler emits no vector-memory instruction of its own inside the tile
    // loop - a property of the code THIS toolchain generated, checked in the ISA (DESIGN / NOTEBOOK) and by the exact index-set tests on
    // ROCm 7.2.  A library built with another hipcc keeps the kernel for callers who ask for it (mivos_memory_read_set_q256_min, the env
    // variable above) but does not select it on its own: deep banks then run on the 128-query kernel, whose requests are counted the same
    // way but whose candidate handling has no loads the compiler could reorder against them.
    if (!Q256_TOOLCHAIN_VALIDATED && !getenv("MIVOS_MEMREAD_Q256_MIN")) {
      v = 0x7fffffffffffffffLL;
      // one line, once per process: a large, otherwise invisible difference in speed on deep banks (config 5: x 1.2)
      fprintf(stderr, "mivos_hip: built with HIP %d.%d - the 256-query memory-read kernel is validated on HIP 7.2 only and is not selected automatically; "
                      "banks >= 200 k positions run on the 128-query kernel (set MIVOS_MEMREAD_Q256_MIN=200000 to use it anyway)\n",
              HIP_VERSION_MAJOR, HIP_VERSION_MINOR);
    }
    g_q256_min.store(v, std::memory_order_relaxed);
  }
  return v;
}
static int select_qt(bool f16, long long n_mem) {
  if (f16 && n_mem >= q256_min()) return QT3;
  return (f16 && n_mem >= q128_min()) ? QT2 : QT;
}

static long long lists_bytes(const Plan &p) { return (long long)p.streams * p.slots * p.qt * p.L * 8; }
// workspace layout: [64-byte header][candidate lists of the largest plan][candidate regions of the 256-query kernel]
static long long max_lists_bytes(int n_obj, long long n_mem, int n_q, int top_k) {
  long long m = 0;
  for (int qt : {QT, QT2, QT3}) {
    const long long b = lists_bytes(make_plan(n_obj, n_mem, n_q, top_k, qt));
    m = b > m ? b : m;
  }
  return (m + 255) / 256 * 256;
}
static long long cand3_bytes() { return (long long)compute_units() * CAND3_PER_WG * 8; }
static long long kmax_bytes(int n_obj) { return ((long long)n_obj * 4 + 255) / 256 * 256; }   // hi-first kernel: max squared key norm per object
static std::atomic<int> g_hifirst{-1};
static int hifirst() {
  int v = g_hifirst.load(std::memory_order_relaxed);
  if (v < 0) {
    v = getenv("MIVOS_MEMREAD_HIFIRST") ? atoi(getenv("MIVOS_MEMREAD_HIFIRST")) : 0;
    g_hifirst.store(v, std::memory_order_relaxed);
  }
  return v;
}

static int check_select_args(const float *keys, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem, int n_q,
                             int top_k, void *workspace, int64_t workspace_bytes) {
  if (!keys || !qk || !workspace) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: null pointer");
  if (top_k < 1 || top_k > MAX_TOPK) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: top_k=%d unsupported (1..%d)", top_k, MAX_TOPK);
  if (n_obj < 1 || n_q < 1 || n_mem < 1 || n_mem >= 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: bad sizes");
  if (n_mem < top_k) return fail(MIVOS_ERR_TOPK_RANGE, "selected index k out of range (top_k=%d > %lld memory positions)", top_k, (long long)n_mem);
  if (((uintptr_t)keys & 15) || ((uintptr_t)qk & 15) || (keys_ostride & 3) || ((uintptr_t)workspace & 15))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: keys/qk/workspace must be 16-byte aligned");
  if (workspace_bytes < mivos_memory_read_workspace_bytes(n_obj, n_mem, n_q, top_k)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: workspace too small");
  return MIVOS_OK;
}

template <bool INDICES, int N>
static void launch_finalize_n(void *workspace, const float *values, int64_t values_ostride, float *out, int64_t out_ostride,
                              int64_t out_pstride, int32_t *idx_out, float *w_out, int n_obj, int n_q, int top_k, hipStream_t st, const ShOut &sh) {
  hipLaunchKernelGGL((memread_finalize_kernel<INDICES, N>), dim3(n_q, n_obj), dim3(64), 0, st, (const int *)workspace, values,
                     (long long)values_ostride, out, (long long)out_ostride, (long long)out_pstride, idx_out, w_out, n_q, top_k, sh);
}

// Which plan the last select launch on a workspace used (host-side mirror of the header it leaves in the workspace, which the
// host cannot read without a synchronisation): the finalize launch sizes its per-lane merge buffer for THAT plan's segment
// count instead of the largest any of the three kernels could have produced (480p, 5 objects: 4 entries per lane instead of
// 15 - the 256-query plan that never runs there cuts a stream into 9 segments).  Registers, not time: the launch is bound by
// its value gather (810 MB of rows per launch at 480p, 5 objects; 82 vs 84 us, profiles/r03h_config3_kernel_stats*.csv).
struct WsPlan { int qt, slots; };
static std::mutex g_ws_mutex;
static std::unordered_map<const void *, WsPlan> g_ws_qt;
// Bounded: callers reallocate their workspace when the bank grows, so stale addresses accumulate in a long-lived process; when
// the table is full it is dropped (finalize then sizes for the largest plan - the documented fallback, 84 vs 82 us).  An entry
// exists only while the LAST select launch on that workspace succeeded: launch_select erases it first and records it after the
// launch was accepted, so a failed select can never leave a plan that did not write the header.
static const size_t WS_PLAN_ENTRIES = 256;
static void remember_select_plan(const void *workspace, int qt, int slots) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (g_ws_qt.size() >= WS_PLAN_ENTRIES && !g_ws_qt.count(workspace)) g_ws_qt.clear();
  g_ws_qt[workspace] = WsPlan{qt, slots};
}
static void forget_select_plan(const void *workspace) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  g_ws_qt.erase(workspace);
}
static WsPlan recall_select_plan(const void *workspace) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  const auto it = g_ws_qt.find(workspace);
  return it == g_ws_qt.end() ? WsPlan{0, 0} : it->second;
}

// The kernel reads the plan from the workspace header; the host only picks how many merged candidates a lane may hold: for the
// plan the last select launch on this workspace used, or - a workspace this process has not seen a select on - for the
// largest of the plans a select call could have used with these sizes.
static int launch_finalize(bool indices, void *workspace, const float *values, int64_t values_ostride, float *out,
                           int64_t out_ostride, int64_t out_pstride, int32_t *idx_out, float *w_out, int n_obj, int64_t n_mem, int n_q, int top_k,
                           hipStream_t st, const ShOut &sh = ShOut{nullptr, nullptr, 0, 0, 0, 1}) {
  const Plan p64 = make_plan(n_obj, n_mem, n_q, top_k, QT);
  int slots;
  const WsPlan seen = recall_select_plan(workspace);
  if (seen.qt) {
    // the slot count the select launch on this workspace actually used (remembered with its plan): mivos_memory_read_set_workgroups may have
    // changed the workgroup count since - a plan recomputed here could be smaller than what the header in the workspace describes
    slots = seen.slots;
  } else {
    const Plan p128 = make_plan(n_obj, n_mem, n_q, top_k, QT2), p256 = make_plan(n_obj, n_mem, n_q, top_k, QT3);
    slots = p64.slots > p128.slots ? p64.slots : p128.slots;
    slots = p256.slots > slots ? p256.slots : slots;
  }
  const int per_lane = cdiv((long long)slots * p64.L, 64);       // slots bounds the segments of any stream
#define MIVOS_FIN(N)                                                                                                              \
  (indices ? launch_finalize_n<true, N>(workspace, values, values_ostride, out, out_ostride, out_pstride, idx_out, w_out, n_obj, n_q, top_k, st, sh) \
           : launch_finalize_n<false, N>(workspace, values, values_ostride, out, out_ostride, out_pstride, idx_out, w_out, n_obj, n_q, top_k, st, sh))
  if (per_lane <= 4) MIVOS_FIN(4);
  else if (per_lane <= 8) MIVOS_FIN(8);
  else MIVOS_FIN(FIN_EPL_MAX);
#undef MIVOS_FIN
  return check_launch("memread_finalize");
}

static int check_finalize_args(const char *what, int n_obj, int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes) {
  if (!workspace || ((uintptr_t)workspace & 15) || top_k < 1 || top_k > MAX_TOPK || n_obj < 1 || n_q < 1 || n_mem < top_k ||
      workspace_bytes < mivos_memory_read_workspace_bytes(n_obj, n_mem, n_q, top_k))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "%s: arguments do not match the select call", what);
  return MIVOS_OK;
}

}  // namespace mivos

using namespace mivos;

extern "C" int64_t mivos_memory_read_workspace_bytes(int n_obj, int64_t n_mem, int n_q, int top_k) {
  if (n_obj < 1 || n_q < 1 || n_mem < 1 || top_k < 1) return 0;
  return HEADER_BYTES + max_lists_bytes(n_obj, n_mem, n_q, top_k) + cand3_bytes() + kmax_bytes(n_obj);
}

extern "C" int mivos_memory_read_set_workgroups(int n_wg) {
  const int prev = g_select_wgs.load(std::memory_order_relaxed);
  if (n_wg >= 0) g_select_wgs.store(n_wg, std::memory_order_relaxed);
  return prev;
}

extern "C" int mivos_memory_read_set_hifirst(int on) {
  const int old = hifirst();
  if (on >= 0) g_hifirst.store(on ? 1 : 0, std::memory_order_relaxed);
  return old;
}

extern "C" int64_t mivos_memory_read_set_q256_min(int64_t n_mem_min) {
  const long long old = q256_min();
  if (n_mem_min >= 0) g_q256_min.store(n_mem_min, std::memory_order_relaxed);
  return old;
}

extern "C" int64_t mivos_memory_read_set_q128_min(int64_t n_mem_min) {
  const long long old = q128_min();
  if (n_mem_min >= 0) g_q128_min.store(n_mem_min, std::memory_order_relaxed);
  return old;
}

extern "C" int mivos_memory_read_plan(int n_obj, int64_t n_mem, int n_q, int top_k, int f16x3, int32_t *plan_out) {
  if (!plan_out || n_obj < 1 || n_q < 1 || n_mem < 1 || top_k < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_plan: bad arguments");
  const Plan p = make_plan(n_obj, n_mem, n_q, top_k, select_qt(f16x3 != 0, n_mem));
  plan_out[0] = p.n_wg; plan_out[1] = p.tiles_per_wg; plan_out[2] = p.slots; plan_out[3] = p.tps; plan_out[4] = p.streams; plan_out[5] = p.L;
  plan_out[6] = p.qt;
  return MIVOS_OK;
}

static int launch_select(bool f16, const float *keys, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem, int n_q, int top_k,
                         void *workspace, int64_t workspace_bytes, void *stream) {
  if (int rc = check_select_args(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes)) return rc;
  const int qt = select_qt(f16, n_mem);
  const Plan pl = make_plan(n_obj, n_mem, n_q, top_k, qt);
  forget_select_plan(workspace);
  SelectArgs a;
  a.keys = keys; a.keys_ostride = keys_ostride; a.qk = qk; a.header = (int *)workspace;
  a.lists = (uint64_t *)((char *)workspace + HEADER_BYTES); a.n_mem = n_mem; a.n_q = n_q;
  a.top_k = top_k; a.n_qtiles = pl.n_qtiles; a.tps = pl.tps; a.total_tiles = pl.total; a.tiles_per_wg = pl.tiles_per_wg;
  a.slots = pl.slots; a.L = pl.L; a.qt = qt;
  a.cand = (uint64_t *)((char *)workspace + HEADER_BYTES + max_lists_bytes(n_obj, n_mem, n_q, top_k));
  a.kmax2 = nullptr;
  static const int sel_xcd = getenv("MIVOS_SELECT_XCD") ? atoi(getenv("MIVOS_SELECT_XCD")) : 1;   // 0: block order (tuning / A-B)
  a.contig = sel_xcd;
  static const int abl = getenv("MIVOS_ABL") ? atoi(getenv("MIVOS_ABL")) : 0;          // profiling only
  static const int dbg = getenv("MIVOS_MEMREAD_DBG") ? atoi(getenv("MIVOS_MEMREAD_DBG")) : 0;   // profiling only: prints cycles per tile
  static unsigned long long *dbg_buf = nullptr;
  a.dbg = nullptr;
  if (dbg) {
    if (!dbg_buf && hipMalloc((void **)&dbg_buf, 64) != hipSuccess) dbg_buf = nullptr;
    a.dbg = dbg_buf;
  }
  // the wave-uniform skip of the append pays once most compares fail everywhere: long memories
  static const int br_min = getenv("MIVOS_MEMREAD_BR_MIN") ? atoi(getenv("MIVOS_MEMREAD_BR_MIN")) : 32768;   // tuning only
  const bool br = n_mem >= br_min;
  const dim3 grid(pl.n_wg), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MIVOS_SEL(ABL, BR, F16) hipLaunchKernelGGL((memread_select_kernel<ABL, BR, F16>), grid, block, 0, st, a)
#define MIVOS_SEL32(ABL, BR) hipLaunchKernelGGL((memread_select32_kernel<ABL, BR>), grid, block, 0, st, a)
  if (qt == QT3) {
    if ((long long)pl.n_wg * CAND3_PER_WG * 8 > cand3_bytes()) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: more workgroups than candidate scratch");
    if (hifirst()) {
      float *kmax2 = (float *)((char *)a.cand + cand3_bytes());
      if (hipMemsetAsync(kmax2, 0, (size_t)n_obj * 4, st) != hipSuccess) return fail(MIVOS_ERR_LAUNCH, "memory_read: hipMemsetAsync");
      const long long blocks = (n_mem * 16 + 255) / 256;
      hipLaunchKernelGGL(key_norm2_max_kernel, dim3((unsigned)blocks, n_obj), dim3(256), 0, st, keys, (long long)keys_ostride, (long long)n_mem, kmax2);
      a.kmax2 = kmax2;
      hipLaunchKernelGGL(memread_select256_kernel<true>, grid, dim3(512), 0, st, a);
    } else {
      hipLaunchKernelGGL(memread_select256_kernel<false>, grid, dim3(512), 0, st, a);
    }
  } else if (qt == QT2) {
    if (abl == 1) MIVOS_SEL32(1, false);
    else if (br) MIVOS_SEL32(0, true);
    else MIVOS_SEL32(0, false);
  } else if (f16) {
    if (abl == 1) MIVOS_SEL(1, false, true);
    else if (br) MIVOS_SEL(0, true, true);
    else MIVOS_SEL(0, false, true);
  } else {
    if (abl == 1) MIVOS_SEL(1, false, false);
    else if (br) MIVOS_SEL(0, true, false);
    else MIVOS_SEL(0, false, false);
  }
#undef MIVOS_SEL
#undef MIVOS_SEL32
  if (dbg && dbg_buf) {
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg_buf, 48, hipMemcpyDeviceToHost);
    fprintf(stderr, "[memread_select %s q%d] n_obj=%d n_mem=%lld n_q=%d: workgroup 0 ran %llu tiles in %llu shader-clock ticks = %.0f per tile (MFMA issue alone: %d cycles); "
            "wave 0: %llu compactions in the loop = %llu ticks, segment prologues %llu, drains + final lists %llu\n",
            f16 ? "f16x3" : "f32", qt, n_obj, (long long)n_mem, n_q, h[1], h[0], h[1] ? (double)h[0] / (double)h[1] : 0.0,
            qt == QT2 ? 24 * 32 : (f16 ? 24 * 17 : 64 * 32), h[3], h[2], h[5], h[4]);
  }
  if (int rc = check_launch("memread_select")) return rc;
  remember_select_plan(workspace, qt, pl.slots);
  return MIVOS_OK;
}

extern "C" int mivos_memory_read_select(const float *keys, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem,
                                        int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream) {
  return launch_select(false, keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, stream);
}

extern "C" int mivos_memory_read_select_f16x3(const void *keys_split, int64_t keys_ostride, const float *qk, int n_obj, int64_t n_mem,
                                              int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream) {
  return launch_select(true, (const float *)keys_split, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, stream);
}

extern "C" int mivos_memory_split_keys(const float *keys, int64_t keys_ostride, void *keys_split, int64_t split_ostride, int n_obj,
                                       int64_t n_rows, void *stream) {
  if (!keys || !keys_split || n_obj < 1 || n_rows < 1 || ((uintptr_t)keys & 15) || ((uintptr_t)keys_split & 15) || (keys_ostride & 3) || (split_ostride & 3))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_split_keys: null / misaligned pointer or bad sizes");
  const long long blocks = (n_rows * 16 + 255) / 256;
  if (blocks > 0x7fffffffLL || n_obj > 65535) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_split_keys: too many rows / objects for one launch");
  hipLaunchKernelGGL(split_keys_kernel, dim3((unsigned)blocks, n_obj), dim3(256), 0, (hipStream_t)stream, keys, (long long)keys_ostride,
                     (float *)keys_split, (long long)split_ostride, (long long)n_rows);
  return check_launch("memory_split_keys");
}

extern "C" int mivos_memory_read_finalize(const float *values, int64_t values_ostride, float *out, int64_t out_ostride,
                                          int64_t out_pstride, int n_obj, int64_t n_mem, int n_q, int top_k, void *workspace,
                                          int64_t workspace_bytes, void *stream) {
  if (!values || !out || ((uintptr_t)values & 15) || ((uintptr_t)out & 15) || (out_pstride & 3) || (out_ostride & 3) || (values_ostride & 3))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read: values/out must be non-null and 16-byte aligned");
  if (int rc = check_finalize_args("memory_read_finalize", n_obj, n_mem, n_q, top_k, workspace, workspace_bytes)) return rc;
  return launch_finalize(false, workspace, values, values_ostride, out, out_ostride, out_pstride, nullptr, nullptr, n_obj, n_mem, n_q,
                         top_k, (hipStream_t)stream);
}

extern "C" int mivos_memory_read_finalize_sh32(const float *values, int64_t values_ostride, void *raw_sh32, void *relu_sh32,
                                               int64_t a_nstride, int64_t a_rstride, int64_t a_pstride, int q_width, int n_obj,
                                               int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream) {
  if (!values || (!raw_sh32 && !relu_sh32) || ((uintptr_t)values & 15) || (values_ostride & 3) || q_width < 1 || n_q % q_width ||
      ((a_nstride | a_rstride | a_pstride) & 31) || ((uintptr_t)raw_sh32 & 127) || ((uintptr_t)relu_sh32 & 127))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_finalize_sh32: bad arguments");
  if (int rc = check_finalize_args("memory_read_finalize_sh32", n_obj, n_mem, n_q, top_k, workspace, workspace_bytes)) return rc;
  const ShOut sh{(float *)raw_sh32, (float *)relu_sh32, a_nstride, a_rstride, a_pstride, q_width};
  return launch_finalize(false, workspace, values, values_ostride, nullptr, 0, 0, nullptr, nullptr, n_obj, n_mem, n_q, top_k, (hipStream_t)stream, sh);
}

extern "C" int mivos_memory_read_finalize_indices(int32_t *idx_out, float *weight_out, int n_obj, int64_t n_mem, int n_q, int top_k,
                                                  void *workspace, int64_t workspace_bytes, void *stream) {
  if (!idx_out || !weight_out) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_finalize_indices: null pointer");
  if (int rc = check_finalize_args("memory_read_finalize_indices", n_obj, n_mem, n_q, top_k, workspace, workspace_bytes)) return rc;
  return launch_finalize(true, workspace, nullptr, 0, nullptr, 0, 0, idx_out, weight_out, n_obj, n_mem, n_q, top_k, (hipStream_t)stream);
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
  if (int rc = mivos_memory_read_select(keys, keys_ostride, qk, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, stream)) return rc;
  return mivos_memory_read_finalize_indices(idx_out, weight_out, n_obj, n_mem, n_q, top_k, workspace, workspace_bytes, stream);
}
