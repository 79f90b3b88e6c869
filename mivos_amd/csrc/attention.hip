// Difference-aware attention alignment (T = 1): for every query position q
//     out[0][q] = sum_m pos16[m] * softmax_m(mk[m] . qk[q] / sqrt(128)),   out[1][q] likewise with neg16
// i.e. PropagationNetwork.get_attention's  `pos/neg [1 x HW] @ W[HW x HW]`  (prop_net.py:187-196) with W
// (AttentionMemory.forward, prop_net.py:115-129) folded in flash-style: score tiles come from
// v_mfma_f32_32x32x2_f32, every lane keeps a running (max, denominator, pos-numerator, neg-numerator)
// for its query and the [HW x HW] softmax matrix is never written.
// Workgroup = 4 waves on the SAME 32 queries; wave w takes memory tiles w, w+4, ... (no LDS staging:
// a tile is used by one wave only; the 0.8 MB key map stays L2-resident), partials meet in LDS.
#include "common.h"

namespace mivos {

constexpr int ACK = 128;

__global__ __launch_bounds__(256) void attention_align_kernel(const float *__restrict__ mk, const float *__restrict__ qk,
                                                             const float *__restrict__ pos16, const float *__restrict__ neg16,
                                                             float *__restrict__ out, int n_pos) {
  __shared__ float part[4][32][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int obj = blockIdx.y;
  const int q = blockIdx.x * 32 + j;
  const float *kbase = mk + (long long)obj * n_pos * ACK;
  const float *pb = pos16 + (long long)obj * n_pos, *nb = neg16 + (long long)obj * n_pos;

  f32x4 qreg[16];
  {
    const float *qrow = qk + (long long)(q < n_pos ? q : n_pos - 1) * ACK + 4 * h;
    const float d = sqrtf((float)ACK);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + 8 * u);
      v.x /= d; v.y /= d; v.z /= d; v.w /= d;
      qreg[u] = v;
    }
  }
  float mx = -INFINITY, den = 0.f, np = 0.f, nn = 0.f;
  const int n_tiles = (n_pos + 31) / 32;
  for (int t = wave; t < n_tiles; t += 4) {
    const int mrow = t * 32 + j;
    const float *arow = kbase + (long long)(mrow < n_pos ? mrow : n_pos - 1) * ACK + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + 8 * u);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qreg[u][s], acc, 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t * 32 + mfma32_row(r, lane);
      if (m < n_pos) tmax = fmaxf(tmax, acc[r]);
    }
    const float nmx = fmaxf(mx, tmax);
    if (nmx > -INFINITY) {
      const float sc = expf(mx - nmx);  // mx = -inf -> 0
      den *= sc; np *= sc; nn *= sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = t * 32 + mfma32_row(r, lane);
        if (m < n_pos) {
          const float e = expf(acc[r] - nmx);
          den += e;
          np = fmaf(pb[m], e, np);
          nn = fmaf(nb[m], e, nn);
        }
      }
      mx = nmx;
    }
  }
  // merge the two half-wave partials of each query, then the 4 waves
  {
    const float omx = __shfl_xor(mx, 32, 64), oden = __shfl_xor(den, 32, 64);
    const float onp = __shfl_xor(np, 32, 64), onn = __shfl_xor(nn, 32, 64);
    const float M = fmaxf(mx, omx);
    const float a = mx > -INFINITY ? expf(mx - M) : 0.f, b = omx > -INFINITY ? expf(omx - M) : 0.f;
    den = den * a + oden * b; np = np * a + onp * b; nn = nn * a + onn * b; mx = M;
  }
  if (h == 0) { part[wave][j][0] = mx; part[wave][j][1] = den; part[wave][j][2] = np; part[wave][j][3] = nn; }
  __syncthreads();
  if (wave == 0 && h == 0 && q < n_pos) {
    float M = -INFINITY;
    for (int w = 0; w < 4; ++w) M = fmaxf(M, part[w][j][0]);
    float D = 0.f, P = 0.f, Ng = 0.f;
    for (int w = 0; w < 4; ++w) {
      const float m_w = part[w][j][0];
      const float a = m_w > -INFINITY ? expf(m_w - M) : 0.f;
      D += part[w][j][1] * a; P += part[w][j][2] * a; Ng += part[w][j][3] * a;
    }
    out[((long long)obj * 2 + 0) * n_pos + q] = P / D;
    out[((long long)obj * 2 + 1) * n_pos + q] = Ng / D;
  }
}

// ---- dense W (PropagationNetwork.get_W / AttentionMemory.forward, prop_net.py:115-129, 183-185) ----------------
// W[obj][m][q] = softmax over m of mk[m].qk[q]/sqrt(128).  Not on the InferenceCore path (get_attention never
// materialises W) but part of the module's public surface.  Two launches: a wave per 32x32 score tile writes the raw
// affinities (fp32 MFMA, operands straight from L2), then one thread per query column normalises it in place
// (max, sum of exp, divide: torch.softmax's formulation; consecutive threads = consecutive q => coalesced rows).
__global__ __launch_bounds__(64) void affinity_tile_kernel(const float *__restrict__ mk, const float *__restrict__ qk,
                                                          long long qk_ostride, float *__restrict__ w, int n_mem, int n_q) {
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  const int obj = blockIdx.z;
  const int q0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int q = q0 + j, mrow = m0 + j;
  const float *qrow = qk + (long long)obj * qk_ostride + (long long)(q < n_q ? q : n_q - 1) * ACK + 4 * h;
  const float *arow = mk + ((long long)obj * n_mem + (mrow < n_mem ? mrow : n_mem - 1)) * ACK + 4 * h;
  const float d = sqrtf((float)ACK);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + 8 * u);
    f32x4 b = *reinterpret_cast<const f32x4 *>(qrow + 8 * u);
    b.x /= d; b.y /= d; b.z /= d; b.w /= d;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  }
  if (q < n_q) {
    float *wb = w + (long long)obj * n_mem * n_q + q;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mfma32_row(r, lane);
      if (m < n_mem) wb[(long long)m * n_q] = acc[r];
    }
  }
}

__global__ void column_softmax_kernel(float *__restrict__ w, int n_mem, int n_q) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_q) return;
  float *col = w + (long long)blockIdx.y * n_mem * n_q + q;
  float mx = -INFINITY;
  for (int m = 0; m < n_mem; ++m) mx = fmaxf(mx, col[(long long)m * n_q]);
  float sum = 0.f;
  for (int m = 0; m < n_mem; ++m) sum += expf(col[(long long)m * n_q] - mx);
  for (int m = 0; m < n_mem; ++m) col[(long long)m * n_q] = expf(col[(long long)m * n_q] - mx) / sum;
}

}  // namespace mivos

using namespace mivos;

extern "C" int mivos_attention_weights(const float *mk, const float *qk, int64_t qk_ostride, float *w, int n_obj, int n_mem,
                                       int n_q, void *stream) {
  if (!mk || !qk || !w || n_obj < 1 || n_mem < 1 || n_q < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "attention_weights: bad arguments");
  if (((uintptr_t)mk & 15) || ((uintptr_t)qk & 15) || (qk_ostride & 3)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "attention_weights: mk/qk must be 16-byte aligned");
  hipLaunchKernelGGL(affinity_tile_kernel, dim3(cdiv(n_q, 32), cdiv(n_mem, 32), n_obj), dim3(64), 0, (hipStream_t)stream, mk, qk,
                     (long long)qk_ostride, w, n_mem, n_q);
  hipLaunchKernelGGL(column_softmax_kernel, dim3(cdiv(n_q, 64), n_obj), dim3(64), 0, (hipStream_t)stream, w, n_mem, n_q);
  return check_launch("attention_weights");
}

extern "C" int mivos_attention_align(const float *mk, const float *qk, const float *pos16, const float *neg16, float *out,
                                     int n_obj, int n_pos, void *stream) {
  if (!mk || !qk || !pos16 || !neg16 || !out || n_obj < 1 || n_pos < 1) return fail(MIVOS_ERR_INVALID_ARGUMENT, "attention_align: bad arguments");
  if (((uintptr_t)mk & 15) || ((uintptr_t)qk & 15)) return fail(MIVOS_ERR_INVALID_ARGUMENT, "attention_align: mk/qk must be 16-byte aligned");
  hipLaunchKernelGGL(attention_align_kernel, dim3(cdiv(n_pos, 32), n_obj), dim3(256), 0, (hipStream_t)stream, mk, qk, pos16, neg16, out, n_pos);
  return check_launch("attention_align");
}
