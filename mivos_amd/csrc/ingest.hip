// Clip ingest on the GPU: the step right before InferenceCore in the reference's evaluation scripts
// (dataset/davis_test_dataset.py:66-110, dataset/yv_test_dataset.py:54-119): decoded uint8 frames -> normalised float
// planes (torchvision ToTensor + Normalize, dataset/range_transform.py:5-8), the YouTube-VOS loader's bicubic resize to
// 480p and its nearest-neighbour resize of the one-hot masks, written straight into the zero-padded layout InferenceCore
// keeps (pad_divide_by, util/tensor_util.py:62-80) so that no host-side float tensor is ever built.  All HBM-bound.
#include "common.h"

namespace mivos {

static inline int ingest_grid(int64_t n) {
  const int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// frames [T][H][W][3] uint8 (HWC, as decoded) -> out[t][c] plane at out + t*out_tstride + c*out_cstride, pixel (y, x) at
// (y + pad_top)*out_rstride + x + pad_left.  v = (float(u8) / 255 - mean[c]) / std[c] with true divisions (bit-identical to
// ToTensor + Normalize).  The padding itself is written by the caller (a zero-initialised tensor).
__global__ void ingest_u8_kernel(const uint8_t *__restrict__ frames, float *__restrict__ out, int T, int H, int W,
                                 int64_t out_tstride, int64_t out_cstride, int64_t out_rstride, int pad_top, int pad_left,
                                 float m0, float m1, float m2, float s0, float s1, float s2) {
  const int64_t total = (int64_t)T * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    int64_t r = i / W;
    const int y = (int)(r % H);
    const int t = (int)(r / H);
    const uint8_t *px = frames + i * 3;
    float *o = out + (int64_t)t * out_tstride + (int64_t)(y + pad_top) * out_rstride + x + pad_left;
    o[0] = ((float)px[0] / 255.0f - m0) / s0;
    o[out_cstride] = ((float)px[1] / 255.0f - m1) / s1;
    o[2 * out_cstride] = ((float)px[2] / 255.0f - m2) / s2;
  }
}

// torch upsample_bicubic2d (align_corners=False, A = -0.75): planes [P][h][w] -> [P][H][W] written at
// out + p*out_pstride + (Y + pad_top)*out_rstride + X + pad_left
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ void resize_bicubic_kernel(const float *__restrict__ x, float *__restrict__ out, int P, int h, int w, int H, int W,
                                      int64_t out_pstride, int64_t out_rstride, int pad_top, int pad_left) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int64_t total = (int64_t)P * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    int64_t r = i / W;
    const int Y = (int)(r % H);
    const int p = (int)(r / H);
    const float fy = sy * ((float)Y + 0.5f) - 0.5f, fx = sx * ((float)X + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float cy[4], cx[4];
    cubic_coeffs(fy - (float)iy, cy);
    cubic_coeffs(fx - (float)ix, cx);
    const float *src = x + (int64_t)p * h * w;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = iy - 1 + a;
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = ix - 1 + b;
        xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
        row += src[(int64_t)yy * w + xx] * cx[b];
      }
      acc += row * cy[a];
    }
    out[(int64_t)p * out_pstride + (int64_t)(Y + pad_top) * out_rstride + X + pad_left] = acc;
  }
}

// label map [h][w] uint8 (palette indices) -> one-hot float planes: plane 0 = background (none of the labels), plane 1 + k =
// labels[k], resized with torch's 'nearest' rule (src = floor(dst * in / out)), at out + plane*out_pstride +
// (Y + pad_top)*out_rstride + X + pad_left
__global__ void onehot_nearest_kernel(const uint8_t *__restrict__ lab, const uint8_t *__restrict__ labels, int n_labels,
                                      float *__restrict__ out, int h, int w, int H, int W, int64_t out_pstride,
                                      int64_t out_rstride, int pad_top, int pad_left) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int64_t total = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W), Y = (int)(i / W);
    int yy = (int)floorf((float)Y * sy), xx = (int)floorf((float)X * sx);
    yy = yy > h - 1 ? h - 1 : yy;
    xx = xx > w - 1 ? w - 1 : xx;
    const uint8_t v = lab[(int64_t)yy * w + xx];
    float *o = out + (int64_t)(Y + pad_top) * out_rstride + X + pad_left;
    bool any = false;
    for (int k = 0; k < n_labels; ++k) {
      const bool hit = v == labels[k];
      any |= hit;
      o[(int64_t)(k + 1) * out_pstride] = hit ? 1.f : 0.f;
    }
    o[0] = any ? 0.f : 1.f;
  }
}

}  // namespace mivos

using namespace mivos;

extern "C" int mivos_ingest_u8(const uint8_t *frames, float *out, int T, int H, int W, int64_t out_tstride, int64_t out_cstride,
                               int64_t out_rstride, int pad_top, int pad_left, const float *mean3, const float *std3, void *stream) {
  if (!frames || !out || !mean3 || !std3 || T < 1 || H < 1 || W < 1 || pad_top < 0 || pad_left < 0)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "ingest_u8: bad arguments");
  hipLaunchKernelGGL(ingest_u8_kernel, dim3(ingest_grid((int64_t)T * H * W)), dim3(256), 0, (hipStream_t)stream, frames, out, T, H, W,
                     out_tstride, out_cstride, out_rstride, pad_top, pad_left, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return check_launch("ingest_u8");
}

extern "C" int mivos_resize_bicubic(const float *x, float *out, int planes, int h, int w, int H, int W, int64_t out_pstride,
                                    int64_t out_rstride, int pad_top, int pad_left, void *stream) {
  if (!x || !out || planes < 1 || h < 1 || w < 1 || H < 1 || W < 1 || pad_top < 0 || pad_left < 0)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "resize_bicubic: bad arguments");
  hipLaunchKernelGGL(resize_bicubic_kernel, dim3(ingest_grid((int64_t)planes * H * W)), dim3(256), 0, (hipStream_t)stream, x, out, planes, h, w, H,
                     W, out_pstride, out_rstride, pad_top, pad_left);
  return check_launch("resize_bicubic");
}

extern "C" int mivos_onehot_nearest(const uint8_t *label_map, const uint8_t *labels, int n_labels, float *out, int h, int w, int H,
                                    int W, int64_t out_pstride, int64_t out_rstride, int pad_top, int pad_left, void *stream) {
  if (!label_map || !labels || !out || n_labels < 1 || h < 1 || w < 1 || H < 1 || W < 1)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "onehot_nearest: bad arguments");
  hipLaunchKernelGGL(onehot_nearest_kernel, dim3(ingest_grid((int64_t)H * W)), dim3(256), 0, (hipStream_t)stream, label_map, labels, n_labels, out,
                     h, w, H, W, out_pstride, out_rstride, pad_top, pad_left);
  return check_launch("onehot_nearest");
}
