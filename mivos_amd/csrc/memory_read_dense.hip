// Full-softmax space-time memory read for gfx950: PropagationNetwork(top_k=None), the reference's "no top-k" configuration
// (model/propagation/prop_net.py:99-102: affinity = F.softmax(affinity, dim=1); :104-108 mem = mv @ affinity).  The [T*H*W x H*W]
// affinity is never materialised: one pass over the memory with a running (max, denominator, numerator) per query - the
// same recurrence as csrc/attention.hip, with the 512-channel value readout on the matrix cores.
//
// memread_dense_kernel: a workgroup = 4 waves x 16 queries against one segment of one object's memory, tiles of 16 positions.
//   scores  S[16 positions][16 queries]: 32 x v_mfma_f32_16x16x4_f32, A = key rows from LDS (the fragment / channel permutation of
//           memread_select_kernel's exact-fp32 variant), B = the wave's queries, divided by sqrt(128) like prop_net.py:86;
//           lane (q, g) then holds the scores of positions 4g .. 4g+3 for query q;
//   softmax running maximum per query (two cross-lane exchanges), p = exp(s - max), denominator; the 32 x 4 output accumulators are
//           rescaled only when some query's maximum moved (wave-uniform test: after the first tiles it almost never does);
//   readout O[16 queries][512] += P[16 x 16] V[16 x 512]: 4 x 32 x v_mfma_f32_16x16x4_f32 - the score registers ARE the A
//           operand (lane (q, g) supplies position 4g + r in step r, so the B operand reads value row 4g + r: any bijection
//           between k-slots and positions works as long as both operands use the same one), B = one float per lane from the
//           LDS value tile (516-float pitch: the four row groups of a step hit disjoint banks).
//   Exact fp32 throughout (the matrix pipe's fp32 rate, 157 TFLOP/s, bounds it: 5 x the affinity FLOPs).
// memread_dense_merge_kernel: one wave per (object, query) combines the segments' (max, denominator, numerator) and writes the
//   fp32 row and / or the SH32 activations of the decoder (raw and relu), like memread_finalize_kernel.
#include <math.h>

#include "conv_common.h"

namespace mivos {

constexpr int DCK = 128, DCV = 512;
constexpr int DQW = 16, DQT = 64, DKT = 16;
constexpr int DKLD = 132, DVLD = 516;
constexpr int DREC = DCV + 2;                        // floats per (object, segment, query) record: max, denominator, numerator[512]

typedef __attribute__((ext_vector_type(4))) float d4_t;

struct DenseArgs {
  const float *keys, *values, *qk;
  long long keys_ostride, values_ostride;
  float *rec;                                        // [n_obj][n_seg][n_q][DREC]
  int n_mem, n_q, n_qtiles, n_seg, tiles_per_seg;
};

__global__ __launch_bounds__(256, 2) void memread_dense_kernel(const DenseArgs a) {
  __shared__ __attribute__((aligned(16))) float ktile[DKT * DKLD];
  __shared__ __attribute__((aligned(16))) float vtile[DKT * DVLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jq = lane & 15, g = lane >> 4;
  const int coff = 64 * (g & 1) + 32 * (g >> 1);    // channel of (step u, k-slot g, element s): coff + 4u + s (memory_read.hip)
  int b = blockIdx.x;
  const int seg = b % a.n_seg; b /= a.n_seg;
  const int qtile = b % a.n_qtiles;
  const int obj = b / a.n_qtiles;
  const int t0 = seg * a.tiles_per_seg;
  const int n_tiles_all = (a.n_mem + DKT - 1) / DKT;
  const int t1 = t0 + a.tiles_per_seg < n_tiles_all ? t0 + a.tiles_per_seg : n_tiles_all;
  const float *kbase = a.keys + (long long)obj * a.keys_ostride;
  const float *vbase = a.values + (long long)obj * a.values_ostride;

  d4_t qreg[8];
  {
    const int q = qtile * DQT + wave * DQW + jq;
    const float *qrow = a.qk + (long long)(q < a.n_q ? q : a.n_q - 1) * DCK;
    const float d = sqrtf((float)DCK);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      d4_t v = *reinterpret_cast<const d4_t *>(qrow + coff + 4 * u);
      v.x /= d; v.y /= d; v.z /= d; v.w /= d;
      qreg[u] = v;
    }
  }
  d4_t O[32];
#pragma unroll
  for (int nb = 0; nb < 32; ++nb) O[nb] = d4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int t = t0; t < t1; ++t) {
    const int row0 = t * DKT;
    __syncthreads();                                 // the previous tile's readers are done
    {                                                // stage the key tile (16 x 128) and the value tile (16 x 512); rows past the end: zeros
      const int r = tid >> 4, c = tid & 15;          // 16 threads per row
      const bool ok = row0 + r < a.n_mem;
      const float *kr = kbase + (long long)(ok ? row0 + r : 0) * DCK;
      const float *vr = vbase + (long long)(ok ? row0 + r : 0) * DCV;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        d4_t v = *reinterpret_cast<const d4_t *>(kr + 4 * (c + 16 * j));
        if (!ok) v = d4_t{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<d4_t *>(&ktile[r * DKLD + 4 * (c + 16 * j)]) = v;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        d4_t v = *reinterpret_cast<const d4_t *>(vr + 4 * (c + 16 * j));
        if (!ok) v = d4_t{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<d4_t *>(&vtile[r * DVLD + 4 * (c + 16 * j)]) = v;
      }
    }
    __syncthreads();
    // scores of the 16 positions against this wave's 16 queries
    d4_t s = {0.f, 0.f, 0.f, 0.f};
    {
      const float *arow = &ktile[jq * DKLD + coff];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const d4_t fa = *reinterpret_cast<const d4_t *>(arow + 4 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], qreg[u][e], s, 0, 0, 0);
      }
    }
    // lane (q = jq, g): s[r] = score of position row0 + 4g + r
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + 4 * g + r >= a.n_mem) s[r] = -INFINITY;
    float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);            // finite: every tile of a segment holds at least one real position
    const float scale = expf(m_run - m_new);         // exp(-inf) = 0 on the first tile
    d4_t p;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = expf(s[r] - m_new);
    float ps = (p[0] + p[1]) + (p[2] + p[3]);
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * scale + ps;
    m_run = m_new;
    if (__ballot(scale != 1.f)) {                    // some query's maximum moved: rescale its numerators
      // accumulator layout: lane (c = lane & 15, qg = lane >> 4), register r = query 4 qg + r; the scale of query j lives in lanes with jq == j
      float sc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[r] = __shfl(scale, 4 * g + r);
#pragma unroll
      for (int nb = 0; nb < 32; ++nb) { O[nb][0] *= sc[0]; O[nb][1] *= sc[1]; O[nb][2] *= sc[2]; O[nb][3] *= sc[3]; }
    }
    // O[q][c] += sum_r sum_g p(q, 4g + r) * V[4g + r][c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *vrow = &vtile[(4 * g + r) * DVLD + jq];
#pragma unroll
      for (int nb = 0; nb < 32; ++nb) O[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], vrow[16 * nb], O[nb], 0, 0, 0);
    }
  }
  // record: query of lane (jq, g = 0) owns (max, denominator); numerators: lane (c, qg) register r -> query 4 qg + r, channel 16 nb + c
  const int qbase = qtile * DQT + wave * DQW;
  float *rbase = a.rec + (((long long)obj * a.n_seg + seg) * a.n_q) * DREC;
  if (g == 0 && qbase + jq < a.n_q) {
    rbase[(long long)(qbase + jq) * DREC] = m_run;
    rbase[(long long)(qbase + jq) * DREC + 1] = l_run;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = qbase + 4 * g + r;
    if (q < a.n_q) {
      float *dst = rbase + (long long)q * DREC + 2 + jq;
#pragma unroll
      for (int nb = 0; nb < 32; ++nb) dst[16 * nb] = O[nb][r];
    }
  }
}

struct DenseOut {
  float *out;
  long long out_ostride, out_pstride;
  float *raw, *relu;                                 // SH32 activations (zero-bordered buffers of the LDS-DMA convolutions) or NULL
  long long ns, rs, ps;
  int q_width;
};

__global__ __launch_bounds__(64) void memread_dense_merge_kernel(const float *__restrict__ rec, int n_seg, int n_q, DenseOut o) {
  const int q = blockIdx.x, obj = blockIdx.y, lane = threadIdx.x;
  const float *base = rec + ((long long)obj * n_seg * n_q + q) * DREC;
  const long long sstride = (long long)n_q * DREC;
  float M = -INFINITY;
  for (int s = 0; s < n_seg; ++s) M = fmaxf(M, base[s * sstride]);
  float den = 0.f;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  for (int s = 0; s < n_seg; ++s) {                  // segments in memory order
    const float *r = base + s * sstride;
    const float w = expf(r[0] - M);
    den += w * r[1];
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(r + 2 + 4 * lane), v1 = *reinterpret_cast<const f32x4 *>(r + 2 + 256 + 4 * lane);
    a0.x += w * v0.x; a0.y += w * v0.y; a0.z += w * v0.z; a0.w += w * v0.w;
    a1.x += w * v1.x; a1.y += w * v1.y; a1.z += w * v1.z; a1.w += w * v1.w;
  }
  a0.x /= den; a0.y /= den; a0.z /= den; a0.w /= den;
  a1.x /= den; a1.y /= den; a1.z /= den; a1.w /= den;
  if (o.out) {
    float *dst = o.out + (long long)obj * o.out_ostride + (long long)q * o.out_pstride + 4 * lane;
    *reinterpret_cast<f32x4 *>(dst) = a0;
    *reinterpret_cast<f32x4 *>(dst + 256) = a1;
  }
  if (o.raw || o.relu) {
    const int qy = q / o.q_width, qx = q - qy * o.q_width;
    const long long pix = (long long)obj * o.ns + (long long)qy * o.rs + (long long)qx * o.ps;
    if (o.raw) { store_sh32x4(o.raw, pix, 4 * lane, a0); store_sh32x4(o.raw, pix, 256 + 4 * lane, a1); }
    if (o.relu) {
      store_sh32x4(o.relu, pix, 4 * lane, f32x4{fmaxf(a0.x, 0.f), fmaxf(a0.y, 0.f), fmaxf(a0.z, 0.f), fmaxf(a0.w, 0.f)});
      store_sh32x4(o.relu, pix, 256 + 4 * lane, f32x4{fmaxf(a1.x, 0.f), fmaxf(a1.y, 0.f), fmaxf(a1.z, 0.f), fmaxf(a1.w, 0.f)});
    }
  }
}

static int dense_segments(int n_obj, long long n_mem, int n_q) {
  const int streams = n_obj * cdiv(n_q, DQT);
  const int tiles = cdiv(n_mem, DKT);
  int seg = cdiv(512, streams);                      // two workgroups per CU on 256 CUs
  if (seg > tiles) seg = tiles;
  if (seg > 64) seg = 64;
  return seg < 1 ? 1 : seg;
}

}  // namespace mivos

using namespace mivos;

extern "C" int64_t mivos_memory_read_dense_workspace_bytes(int n_obj, int64_t n_mem, int n_q) {
  if (n_obj < 1 || n_mem < 1 || n_q < 1) return 0;
  return (int64_t)n_obj * dense_segments(n_obj, n_mem, n_q) * n_q * DREC * 4;
}

extern "C" int mivos_memory_read_dense(const float *keys, int64_t keys_ostride, const float *values, int64_t values_ostride, const float *qk,
                                       float *out, int64_t out_ostride, int64_t out_pstride, void *raw_sh32, void *relu_sh32, int64_t a_nstride,
                                       int64_t a_rstride, int64_t a_pstride, int q_width, int n_obj, int64_t n_mem, int n_q, void *workspace,
                                       int64_t workspace_bytes, void *stream) {
  if (!keys || !values || !qk || !workspace || (!out && !raw_sh32 && !relu_sh32) || n_obj < 1 || n_q < 1 || n_mem < 1 || n_mem >= 0x7fffffffLL ||
      n_obj > 65535 || ((uintptr_t)keys & 15) || ((uintptr_t)values & 15) || ((uintptr_t)qk & 15) || ((uintptr_t)out & 15) || ((uintptr_t)workspace & 15) ||
      (keys_ostride & 3) || (values_ostride & 3) || (out_ostride & 3) || (out_pstride & 3))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_dense: null / misaligned pointer or bad sizes");
  if ((raw_sh32 || relu_sh32) && (q_width < 1 || n_q % q_width || ((a_nstride | a_rstride | a_pstride) & 31) || ((uintptr_t)raw_sh32 & 127) || ((uintptr_t)relu_sh32 & 127)))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_dense: bad SH32 output arguments");
  if (workspace_bytes < mivos_memory_read_dense_workspace_bytes(n_obj, n_mem, n_q)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_dense: workspace too small");
  DenseArgs a;
  a.keys = keys; a.values = values; a.qk = qk; a.keys_ostride = keys_ostride; a.values_ostride = values_ostride; a.rec = (float *)workspace;
  a.n_mem = (int)n_mem; a.n_q = n_q; a.n_qtiles = cdiv(n_q, DQT); a.n_seg = dense_segments(n_obj, n_mem, n_q);
  a.tiles_per_seg = cdiv(cdiv(n_mem, DKT), a.n_seg);
  a.n_seg = cdiv(cdiv(n_mem, DKT), a.tiles_per_seg);                 // no empty segment
  const long long wgs = (long long)n_obj * a.n_qtiles * a.n_seg;
  if (wgs > 0x7fffffffLL) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_dense: too many workgroups");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(memread_dense_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a);
  if (int rc = check_launch("memread_dense")) return rc;
  DenseOut o{out, out_ostride, out_pstride, (float *)raw_sh32, (float *)relu_sh32, a_nstride, a_rstride, a_pstride, q_width < 1 ? 1 : q_width};
  hipLaunchKernelGGL(memread_dense_merge_kernel, dim3(n_q, n_obj), dim3(64), 0, st, (const float *)workspace, a.n_seg, n_q, o);
  return check_launch("memread_dense_merge");
}

// ---- any k: top-k softmax read for k beyond the streaming kernels' candidate lists (k > 64) -------------------------------
// PropagationNetwork accepts every top_k (prop_net.py:133); the streaming select kernels of memory_read.hip hold k <= 64.  Larger k
// (ablations: the reference's default is 50, DAVIS uses 20) take the reference's own route in two launches per query chunk:
//   memread_scores_kernel    S[object][query][position] = the affinity (exact fp32 MFMA) of a chunk of queries -> global scratch;
//   memread_topk_any_kernel  one workgroup per (object, query): exact k-th largest of the row by a 4-pass radix select, the k
//                            survivors (ties: lowest positions first) into LDS, softmax over them (prop_net.py:54-59), readout.
namespace mivos {

constexpr int ANY_MAX_K = 1024;

struct ScoreArgs {
  const float *keys, *qk;
  long long keys_ostride;
  float *S;                      // [n_obj][qc][pitch]
  int n_mem, n_q, q0, qc, pitch, tiles_per_wg;
};

__global__ __launch_bounds__(256) void memread_scores_kernel(const ScoreArgs a) {
  __shared__ __attribute__((aligned(16))) float ktile[DKT * DKLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jq = lane & 15, g = lane >> 4;
  const int coff = 64 * (g & 1) + 32 * (g >> 1);
  const int qtile = blockIdx.y, obj = blockIdx.z;
  const int n_tiles = (a.n_mem + DKT - 1) / DKT;
  const int t0 = blockIdx.x * a.tiles_per_wg, t1 = t0 + a.tiles_per_wg < n_tiles ? t0 + a.tiles_per_wg : n_tiles;
  const float *kbase = a.keys + (long long)obj * a.keys_ostride;
  const int ql = qtile * DQT + wave * DQW + jq;                  // query within the chunk
  d4_t qreg[8];
  {
    const int q = a.q0 + ql;
    const float *qrow = a.qk + (long long)(q < a.n_q ? q : a.n_q - 1) * DCK;
    const float d = sqrtf((float)DCK);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      d4_t v = *reinterpret_cast<const d4_t *>(qrow + coff + 4 * u);
      v.x /= d; v.y /= d; v.z /= d; v.w /= d;
      qreg[u] = v;
    }
  }
  for (int t = t0; t < t1; ++t) {
    const int row0 = t * DKT;
    __syncthreads();
    {
      const int r = tid >> 4, c = tid & 15;
      const bool ok = row0 + r < a.n_mem;
      const float *kr = kbase + (long long)(ok ? row0 + r : 0) * DCK;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        d4_t v = *reinterpret_cast<const d4_t *>(kr + 4 * (c + 16 * j));
        if (!ok) v = d4_t{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<d4_t *>(&ktile[r * DKLD + 4 * (c + 16 * j)]) = v;
      }
    }
    __syncthreads();
    d4_t s = {0.f, 0.f, 0.f, 0.f};
    const float *arow = &ktile[jq * DKLD + coff];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const d4_t fa = *reinterpret_cast<const d4_t *>(arow + 4 * u);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], qreg[u][e], s, 0, 0, 0);
    }
    if (ql < a.qc && a.q0 + ql < a.n_q)                          // lane (q, g): positions row0 + 4g .. + 3 (pitch % 16 == 0: aligned)
      *reinterpret_cast<d4_t *>(a.S + ((long long)obj * a.qc + ql) * a.pitch + row0 + 4 * g) = s;
  }
}

__device__ __forceinline__ uint32_t f2ord_any(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct AnyArgs {
  const float *S, *values;
  long long values_ostride;
  float *out;
  long long out_ostride, out_pstride;
  int32_t *idx_out;              // optional [n_obj][n_q][k] (best first is NOT guaranteed: list order) + weights
  float *w_out;
  int n_mem, n_q, q0, qc, pitch, top_k;
};

__global__ __launch_bounds__(256) void memread_topk_any_kernel(const AnyArgs a) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need;
  __shared__ unsigned cnt_gt[256], cnt_eq[256];
  __shared__ float l_score[ANY_MAX_K];
  __shared__ int l_idx[ANY_MAX_K];
  __shared__ float red[256];
  const int ql = blockIdx.x, obj = blockIdx.y, tid = threadIdx.x;
  const int q = a.q0 + ql;
  const float *row = a.S + ((long long)obj * a.qc + ql) * a.pitch;
  const int k = a.top_k;
  if (tid == 0) { s_prefix = 0u; s_need = (unsigned)k; }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {                          // MSB-first radix select of the k-th largest score
    const int shift = 24 - 8 * pass;
    hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix, himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < a.n_mem; i += 256) {
      const uint32_t o = f2ord_any(row[i]);
      if ((o & himask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, d = 255u;
      for (;; --d) {
        if (hist[d] >= need) break;
        need -= hist[d];
        if (d == 0u) break;
      }
      s_prefix = prefix | (d << shift);
      s_need = need;
    }
    __syncthreads();
  }
  const uint32_t tau = s_prefix;
  // survivors: every score above the k-th, then the lowest positions among the ties; thread t owns a contiguous run of positions
  const int chunk = (a.n_mem + 255) / 256, i0 = tid * chunk, i1 = i0 + chunk < a.n_mem ? i0 + chunk : a.n_mem;
  unsigned ng = 0u, ne = 0u;
  for (int i = i0; i < i1; ++i) {
    const uint32_t o = f2ord_any(row[i]);
    ng += o > tau ? 1u : 0u;
    ne += o == tau ? 1u : 0u;
  }
  cnt_gt[tid] = ng; cnt_eq[tid] = ne;
  __syncthreads();
  if (tid == 0) {                                                 // exclusive scans (256 entries)
    unsigned sg = 0u, se = 0u;
    for (int t = 0; t < 256; ++t) { const unsigned g0 = cnt_gt[t], e0 = cnt_eq[t]; cnt_gt[t] = sg; cnt_eq[t] = se; sg += g0; se += e0; }
    s_need = sg;                                                  // total number of scores above the threshold (< k)
  }
  __syncthreads();
  {
    const unsigned total_gt = s_need;
    unsigned pg = cnt_gt[tid], pe = cnt_eq[tid];
    const unsigned eq_room = (unsigned)k - total_gt;             // ties that still fit
    for (int i = i0; i < i1; ++i) {
      const float v = row[i];
      const uint32_t o = f2ord_any(v);
      if (o > tau) { l_score[pg] = v; l_idx[pg] = i; ++pg; }
      else if (o == tau) { if (pe < eq_room) { l_score[total_gt + pe] = v; l_idx[total_gt + pe] = i; } ++pe; }
    }
  }
  __syncthreads();
  // softmax over the k survivors (max = the best score, prop_net.py:55)
  float m = -INFINITY;
  for (int j = tid; j < k; j += 256) m = fmaxf(m, l_score[j]);
  red[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
  m = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < k; j += 256) { const float e = expf(l_score[j] - m); l_score[j] = e; sum += e; }
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const float denom = red[0];
  if (a.idx_out)
    for (int j = tid; j < k; j += 256) {
      a.idx_out[((long long)obj * a.n_q + q) * k + j] = l_idx[j];
      a.w_out[((long long)obj * a.n_q + q) * k + j] = l_score[j] / denom;
    }
  if (a.out) {                                                    // readout: thread t owns channels 2t, 2t + 1
    const float *vb = a.values + (long long)obj * a.values_ostride + 2 * tid;
    float o0 = 0.f, o1 = 0.f;
    for (int j = 0; j < k; ++j) {
      const float w = l_score[j] / denom;
      const float *vr = vb + (long long)l_idx[j] * DCV;
      o0 = fmaf(w, vr[0], o0);
      o1 = fmaf(w, vr[1], o1);
    }
    float *dst = a.out + (long long)obj * a.out_ostride + (long long)q * a.out_pstride + 2 * tid;
    dst[0] = o0; dst[1] = o1;
  }
}

static int any_pitch(long long n_mem) { return (int)((n_mem + 15) / 16 * 16); }
static int any_chunk(int n_obj, long long n_mem, int n_q) {
  long long qc = (1ll << 30) / ((long long)any_pitch(n_mem) * 4 * n_obj);      // <= 1 GiB of scores at a time
  qc = qc / DQT * DQT;
  if (qc < DQT) qc = DQT;
  const long long all = (long long)cdiv(n_q, DQT) * DQT;
  return (int)(qc < all ? qc : all);
}

}  // namespace mivos

extern "C" int64_t mivos_memory_read_topk_any_workspace_bytes(int n_obj, int64_t n_mem, int n_q) {
  if (n_obj < 1 || n_mem < 1 || n_q < 1) return 0;
  return (int64_t)n_obj * any_chunk(n_obj, n_mem, n_q) * any_pitch(n_mem) * 4;
}

extern "C" int mivos_memory_read_topk_any(const float *keys, int64_t keys_ostride, const float *values, int64_t values_ostride, const float *qk,
                                          float *out, int64_t out_ostride, int64_t out_pstride, int32_t *idx_out, float *weight_out, int n_obj,
                                          int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream) {
  if (!keys || !qk || !workspace || (!out && !idx_out) || (out && !values) || (idx_out && !weight_out) || n_obj < 1 || n_q < 1 || n_mem < 1 ||
      n_mem >= 0x7fffffffLL || n_obj > 65535 || ((uintptr_t)keys & 15) || ((uintptr_t)qk & 15) || ((uintptr_t)workspace & 15) || (keys_ostride & 3) ||
      (out_ostride & 1) || (out_pstride & 1) || (values_ostride & 1))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_topk_any: null / misaligned pointer or bad sizes");
  if (top_k < 1 || top_k > ANY_MAX_K) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_topk_any: top_k=%d unsupported (1..%d)", top_k, ANY_MAX_K);
  if (n_mem < top_k) return fail(MIVOS_ERR_TOPK_RANGE, "selected index k out of range (top_k=%d > %lld memory positions)", top_k, (long long)n_mem);
  if (workspace_bytes < mivos_memory_read_topk_any_workspace_bytes(n_obj, n_mem, n_q)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "memory_read_topk_any: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int qc = any_chunk(n_obj, n_mem, n_q), pitch = any_pitch(n_mem);
  const int n_tiles = cdiv(n_mem, DKT);
  for (int q0 = 0; q0 < n_q; q0 += qc) {
    const int nq = n_q - q0 < qc ? n_q - q0 : qc;
    ScoreArgs sa;
    sa.keys = keys; sa.qk = qk; sa.keys_ostride = keys_ostride; sa.S = (float *)workspace;
    sa.n_mem = (int)n_mem; sa.n_q = n_q; sa.q0 = q0; sa.qc = qc; sa.pitch = pitch;
    const int qtiles = cdiv(nq, DQT);
    int splits = cdiv(1024, qtiles * n_obj);                     // enough workgroups for a few rounds of the chip
    if (splits > n_tiles) splits = n_tiles;
    sa.tiles_per_wg = cdiv(n_tiles, splits);
    splits = cdiv(n_tiles, sa.tiles_per_wg);
    hipLaunchKernelGGL(memread_scores_kernel, dim3(splits, qtiles, n_obj), dim3(256), 0, st, sa);
    if (int rc = check_launch("memread_scores")) return rc;
    AnyArgs aa;
    aa.S = (const float *)workspace; aa.values = values; aa.values_ostride = values_ostride; aa.out = out; aa.out_ostride = out_ostride;
    aa.out_pstride = out_pstride; aa.idx_out = idx_out; aa.w_out = weight_out; aa.n_mem = (int)n_mem; aa.n_q = n_q; aa.q0 = q0; aa.qc = qc;
    aa.pitch = pitch; aa.top_k = top_k;
    hipLaunchKernelGGL(memread_topk_any_kernel, dim3(nq, n_obj), dim3(256), 0, st, aa);
    if (int rc = check_launch("memread_topk_any")) return rc;
  }
  return MIVOS_OK;
}
