// FusionNet.forward (model/fusion_net.py:32-50) as ONE C-ABI call: a host-side composition of the library's own
// launches - conv1 (16 -> 32, ReLU), two residual blocks (32 -> 32 twice each, ReLU after the add), the 32 -> 1 head as
// its 1x1 projection to the nine tap products + mivos_tap_sum9 - in exactly the order, with exactly the kernels, that
// mivos_amd/model/fusion_net.py::FusionNet.run issues them one by one (bit-identical results).  No new device code.
#include <string.h>

#include "common.h"

using namespace mivos;

namespace {

// one precision-1 convolution on dense NHWC tensors
int conv(const mivos_fusion_net_desc &d, const mivos_fusion_layer &L, const float *x, int cin, float *y, int cout, int k,
         const float *res, int relu_out, void *stream) {
  mivos_conv_desc c = {};
  c.x = x; c.w = (const float *)L.w16; c.scale = L.scale16; c.bias = L.bias; c.res = res; c.y = y; c.y2 = nullptr;
  c.N = d.batch; c.H = d.height; c.W = d.width; c.Cin = cin; c.Cout = cout; c.KH = k; c.KW = k; c.stride = 1; c.pad = k / 2;
  c.Ho = d.height; c.Wo = d.width; c.split = cout; c.relu_in = 0; c.relu_out = relu_out; c.precision = 1;
  const int64_t px = (int64_t)d.height * d.width;
  c.x_nstride = px * cin; c.x_pstride = cin;
  c.y_nstride = px * cout; c.y_pstride = cout;
  c.res_nstride = px * cout; c.res_pstride = cout;
  c.workspace = d.workspace; c.workspace_bytes = d.workspace_bytes;
  c.dilation = 1;
  return mivos_conv2d_fused(&c, stream);
}

}  // namespace

extern "C" int64_t mivos_fusion_net_scratch_floats(int batch, int height, int width) {
  if (batch < 1 || height < 1 || width < 1) return 0;
  return 3ll * batch * height * width * 32;
}

extern "C" int mivos_fusion_net_forward(const mivos_fusion_net_desc *dp, void *stream) {
  if (!dp) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: null descriptor");
  const mivos_fusion_net_desc &d = *dp;
  if (!d.x16 || !d.logits || !d.scratch || d.batch < 1 || d.height < 1 || d.width < 1)
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: null pointer or bad sizes");
  if (d.scratch_floats < mivos_fusion_net_scratch_floats(d.batch, d.height, d.width))
    return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: scratch too small (mivos_fusion_net_scratch_floats)");
  for (int i = 0; i < 6; ++i)
    if (!d.layer[i].w16 || !d.layer[i].scale16) return fail(MIVOS_ERR_INVALID_ARGUMENT, "fusion_net_forward: layer %d is not packed", i);
  const int64_t plane = (int64_t)d.batch * d.height * d.width * 32;
  float *A = d.scratch, *B = A + plane, *C = B + plane;
  if (int rc = conv(d, d.layer[0], d.x16, 16, A, 32, 3, nullptr, 1, stream)) return rc;   // x = relu(conv1(cat))    fusion_net.py:39-40
  if (int rc = conv(d, d.layer[1], A, 32, B, 32, 3, nullptr, 1, stream)) return rc;       // r = relu(conv2[0](x))
  if (int rc = conv(d, d.layer[2], B, 32, C, 32, 3, A, 1, stream)) return rc;             // x = relu(x + conv2[2](r))  :42-43
  if (int rc = conv(d, d.layer[3], C, 32, B, 32, 3, nullptr, 1, stream)) return rc;       // r = relu(conv3[0](x))
  if (int rc = conv(d, d.layer[4], B, 32, A, 32, 3, C, 1, stream)) return rc;             // x = relu(x + conv3[2](r))  :45-46
  if (int rc = conv(d, d.layer[5], A, 32, B, 16, 1, nullptr, 0, stream)) return rc;       // nine tap products of final_conv :49
  return mivos_tap_sum9(B, d.final_bias, d.logits, d.batch, d.height, d.width, stream);
}
