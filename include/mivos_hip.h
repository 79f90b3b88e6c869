/*
 * mivos_hip.h — C ABI of libmivos_hip.so, the MI355X (gfx950 / CDNA4) compute library under the
 * MiVOS propagation + difference-aware-fusion drop-in (package mivos_amd).
 *
 * The reference (hkchengrex/MiVOS) has NO native/FFI layer on this path: every FLOP runs inside
 * stock PyTorch ops called from Python classes.  Each entry point below therefore cites the
 * reference *Python* code it replaces (paths relative to the reference root) — that is the
 * interface a maintainer would bind instead of the torch ops (see INTEGRATION.md for the ctypes
 * stub).  Conventions:
 *   - every function returns 0 on success, a negative mivos_status on failure; the message of the
 *     last failure on the calling thread is available from mivos_last_error();
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch tensors in the Python host);
 *     nothing is allocated or freed inside the library; workspaces are passed in explicitly;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous;
 *   - activations are fp32 "NHWC": element (n, y, x, c) lives at
 *         base + n*nstride + (y*W + x)*pstride + c        (strides in ELEMENTS)
 *     so channel slices / concatenations / memory-bank slots are expressed with strides, not copies;
 *   - single-channel maps (masks, probabilities, logits) are planar [planes][H*W].
 * Re-entrancy: calls are safe from any thread / on any stream; no result depends on library state.  What the library keeps
 * between calls, all of it process-wide and none of it data: the thread-local error string; tuning knobs read once from the
 * environment (MIVOS_PP_*, MIVOS_MEMREAD_*: launch geometry only) and the three mivos_memory_read_set_* thresholds; a per-device
 * record of which kernels had their dynamic-LDS attribute raised; and a small mutex-guarded map (<= 256 entries) from a memory-read
 * workspace pointer to the query-tile size its last select launch used, which mivos_memory_read_finalize* checks against the plan
 * that launch left in the workspace header.
 */
#ifndef MIVOS_HIP_H_
#define MIVOS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIVOS_ABI_VERSION 1

typedef enum {
  MIVOS_OK = 0,
  MIVOS_ERR_INVALID_ARGUMENT = -1, /* bad shape / alignment / unsupported configuration      */
  MIVOS_ERR_LAUNCH = -2,           /* hipLaunch / runtime error (message has hipGetErrorString) */
  MIVOS_ERR_DEVICE = -3,           /* not a gfx950 device                                     */
  MIVOS_ERR_TOPK_RANGE = -4        /* top_k larger than the memory (reference: torch.topk raises
                                      "selected index k out of range", prop_net.py:54)           */
} mivos_status;

int mivos_version(void);
const char *mivos_last_error(void);
/* 0 iff `device` is a gfx950 (MI355X) GPU. */
int mivos_device_check(int device);

/* --------------------------------------------------------------------------------------------
 * Fused convolution (implicit GEMM on fp32 MFMA, exact-f32 accumulate).
 * Replaces nn.Conv2d (+ eval BatchNorm2d folded into scale/bias, + ReLU, + residual add) in
 *   model/propagation/mod_resnet.py:92-112 (Bottleneck), torchvision resnet50 (modules.py:70),
 *   modules.py:28-35 (ResBlock), :100-104 (UpsampleBlock), :113-114 (KeyValue),
 *   prop_net.py:23-31 (Decoder), model/fusion_net.py:32-50 (FusionNet).
 *   y = act_out( conv(act_in(x), w) * scale + bias + res )
 * w is OHWI: [Cout][KH][KW][Cin] (K = KH*KW*Cin contiguous).  Cin must be a power of two >= 4
 * (pad the channel dimension with zero weights otherwise).  Output channels [0, split) go to y,
 * channels [split, Cout) go to y2 (split == Cout or y2 == NULL: single destination) — this is how
 * KeyValue's two projections run as one GEMM and land in two memory-bank tensors.
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  const float *x;     /* input,  N x H x W x Cin  (strided NHWC)                                 */
  const float *w;     /* weights, OHWI                                                            */
  const float *scale; /* per-Cout multiplier (folded BN gamma/sqrt(var+eps)) or NULL = 1          */
  const float *bias;  /* per-Cout additive term or NULL = 0                                       */
  const float *res;   /* residual, N x Ho x Wo x Cout (strided NHWC) or NULL                       */
  float *y;           /* output channels [0, split)                                               */
  float *y2;          /* output channels [split, Cout) or NULL                                    */
  int32_t N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int32_t split;      /* see above; set to Cout when y2 is NULL                                   */
  int32_t relu_in;    /* apply max(0,.) to x while loading (pre-activation ResBlock)              */
  int32_t relu_out;   /* apply max(0,.) before the store                                          */
  int32_t precision;  /* 0: exact fp32 MFMA, w = fp32 OHWI.
                         1: error-compensated fp16 MFMA ("f16x3"): x is split on the fly into fp16 hi + lo,
                            w points to weights packed by mivos_pack_weights_f16x3 (hi/lo fp16, scaled by a
                            power of two 2^s; the caller folds 2^-s into `scale`); acc is fp32 and
                            sums hi*hi + hi*lo + lo*hi, i.e. ~2^-22 relative product error (fp32 class).
                            Cout == 1 layers always take the fp32 dot-product kernel (w = fp32 OHWI).
                         2: as 1, but x is already split (SH32 layout, see mivos_pack_activation_sh32), w is
                            packed by mivos_pack_weights_f16x3_dma and both operands are staged by LDS-DMA;
                            Cin % 32 == 0, relu_in must be 0.                                            */
  int64_t x_nstride, x_pstride;
  int64_t y_nstride, y_pstride;
  int64_t y2_nstride, y2_pstride;
  int64_t res_nstride, res_pstride; /* res_nstride == 0 broadcasts one residual over the batch   */
  void *workspace;         /* optional device scratch for split-K (small-M layers: K is cut into slices whose
                              fp32 partial tiles are summed in a fixed order by a second kernel: deterministic);
                              NULL or too small = no split-K; not to be shared by calls running concurrently
                              on two streams                                                               */
  int64_t workspace_bytes;
  /* Row strides in floats; 0 = dense rows (W resp. Wo pixels apart).  A non-dense row stride is how a tensor
   * stored with a border of zero pixels is addressed: x / y / res point at interior pixel (0, 0) of image 0. */
  int64_t x_rstride, y_rstride, res_rstride;
  int32_t x_border;    /* zero pixels guaranteed around every input image; precision 2 requires >= pad       */
  int32_t x_format;    /* 0: fp32, 1: SH32 (must be 1 for precision 2, 0 otherwise)                             */
  int32_t y_format;    /* 0: fp32, 1: SH32 (channels [0, split) only; needs the 16-byte vectorised epilogue)   */
  int32_t res_format;  /* 0: fp32, 1: SH32                                                                      */
  int32_t dilation;    /* spacing of the kernel taps (0 or 1: dense).  DeepLab's atrous convolutions (model/s2m/_deeplab.py:
                          110-118, s2m_resnet.py:17-20) use 2 / 6 / 12 / 18 with pad == dilation; precision 0 / 1 only.       */
  int32_t chip_share;  /* how many independent launch streams the caller keeps busy on this GPU (0 or 1: this launch has the
                          chip to itself).  A hint that must NEVER change results: the split-K slice count (= the fp32 summation
                          order) is a function of the layer shape alone since round 6 (round 5's share-dependent rule made a
                          clip's masks depend on the number of clips in flight); with chip_share > 1 a precision-2 layer that
                          splits K only runs its slices folded inside one workgroup per tile instead of as separate workgroups
                          + a reduce pass - the same additions in the same order, bit-identical (mivos_conv2d_set_fold_mode).   */
  uint32_t *status;    /* optional device word (4-byte aligned, zeroed by the caller; NULL: off).  Precision 1 / 2 only: the
                          epilogue ORs bit 0 into it when an output value leaves the fp16 range (|y| > 65504) - such a value
                          becomes inf in the hi half of the next layer's operand split, and the ReLUs / clamps downstream would
                          hide the NaNs that follow.  One atomic per offending wavefront; nothing is written otherwise.         */
} mivos_conv_desc;

int mivos_conv2d_fused(const mivos_conv_desc *d, void *stream);
/* How a precision-2 layer that splits K runs its slices (results are bit-identical in every mode: same slices, same fp32 additions in the same order):
 * 0 = always as gridDim.y workgroups + a reduce pass; 1 (default) = folded one after the other inside one workgroup per tile when the descriptor says
 * other streams share the chip (chip_share > 1) and the tile grid has >= 64 workgroups; 2 = always folded (tests, tuning).  Process-wide; returns the
 * previous mode, an out-of-range argument only queries.  Environment variable MIVOS_PP_FOLD sets the initial value. */
int mivos_conv2d_set_fold_mode(int mode);
/* Pack OHWI fp32 weights [Cout][KH][KW][Cin] for precision 1.  out[Cout][Kpad/4][8 halves]: per 4
 * consecutive K positions the 4 fp16 "hi" parts of w*mult followed by the 4 fp16 "lo" parts
 * (w*mult - hi); Kpad = KH*KW*Cin rounded up to a multiple of 64 (zero filled).  For Cin % 32 == 0 the
 * K axis is stored "taps inner" (k' = (c/32)*(KH*KW*32) + tap*32 + c%32), the order in which the f16x3
 * kernels walk K (L1/L2 reuse across the taps).  mult must be a power of two.  out: Cout*Kpad*4 bytes. */
int mivos_pack_weights_f16x3(const float *w, void *out, int Cout, int KH, int KW, int Cin, float mult, void *stream);

/* Pre-split operands for precision 2 (LDS-DMA fed f16x3 GEMM, conv_f16x3_dma.hip).
 * SH32 activation layout: N x H x W x C, C % 32 == 0; per pixel and group of 32 channels one 128-byte line =
 * 32 fp16 hi parts then 32 fp16 lo parts (x ~= hi + lo); strides are the fp32 ones (pixel stride C floats).
 * Precision 2 reads its input with im2col offsets and NO padding masks: every image must be stored with a
 * border of >= pad zero pixels (x_border), addressed through x_rstride / x_nstride; the (bordered) tensor and the
 * packed weights must each be smaller than 2 GB (32-bit buffer offsets), pointers 128-byte aligned.
 * mivos_pack_activation_sh32 converts a strided fp32 NHWC tensor into a strided SH32 tensor (e.g. the interior of a
 * zero-bordered buffer), optionally through ReLU (the consumer cannot apply relu_in on DMA-staged data).
 * mivos_pack_weights_f16x3_dma: OHWI fp32 -> 128 zero bytes + [K step][Cout][128 B] hi|lo lines, chunk-swizzled
 * for the LDS image; out must hold mivos_pack_weights_f16x3_dma_bytes(). */
int mivos_pack_activation_sh32(const float *x, int64_t x_nstride, int64_t x_rstride, int64_t x_pstride, void *y, int64_t y_nstride,
                               int64_t y_rstride, int64_t y_pstride, int N, int H, int W, int C, int relu, void *stream);
/* The inverse (x = hi + lo) for consumers outside the convolution path (reference-layout API, tests). */
int mivos_unpack_activation_sh32(const void *x, int64_t x_nstride, int64_t x_rstride, int64_t x_pstride, float *y, int64_t y_nstride,
                                 int64_t y_rstride, int64_t y_pstride, int N, int H, int W, int C, void *stream);
int64_t mivos_pack_weights_f16x3_dma_bytes(int Cout, int KH, int KW, int Cin);
int mivos_pack_weights_f16x3_dma(const float *w, void *out, int Cout, int KH, int KW, int Cin, float mult, void *stream);

/* Which kernel instantiation mivos_conv2d_fused picks for M = N*Ho*Wo output pixels and Cout channels
 * (0: 128x128 tile, 1: 64x64, 2: 128x32, 3: 128x64, 4: Cout==1 dot product) — for profilers/benchmarks. */
int mivos_conv2d_variant(int M, int Cout);
/* Same for precision 1 (f16x3): additionally 5: 256x256 tile / 8 waves pipelined, 6: 128x256 / 8 waves,
 * 7: 128x128 / 8 waves. */
int mivos_conv2d_variant_f16x3(int M, int Cout);
/* Same for precision 2 (LDS-DMA ping-pong kernels): 20: 128x128 tile, 21: 128x256, 22: 128x64, 23: 256x256;
 * ksteps = KH*KW*Cin/32. */
int mivos_conv2d_variant_pp(int M, int Cout, int ksteps);

/* MaxPool2d(3, stride 2, pad 1) on NHWC (mod_resnet.py:121 / torchvision stem). C % 4 == 0. */
int mivos_maxpool3x3s2(const float *x, float *y, int N, int H, int W, int C, void *stream);

/* x = skip + bilinear_up2(up), align_corners=False (modules.py:102).  skip is broadcast over the
 * batch when skip_nstride == 0.  up: N x h x w x C, skip/out: N x 2h x 2w x C, all dense NHWC. */
int mivos_upsample2x_add(const float *skip, int64_t skip_nstride, const float *up, float *out,
                         int N, int h, int w, int C, void *stream);

/* The same with up to three outputs (any may be NULL) that feed the next layers without a conversion pass: `out` dense
 * fp32 NHWC, `raw_sh32` the sum in SH32 and `relu_sh32` = SH32 of max(sum, 0) (pre-activation ResBlocks, modules.py:28-35:
 * a DMA-staged operand cannot be modified on load, so its producer applies the ReLU).  The SH32 outputs point at interior
 * pixel (0, 0) of image 0 of zero-bordered buffers with the given strides (floats). */
int mivos_upsample2x_add_multi(const float *skip, int64_t skip_nstride, const float *up, float *out, void *raw_sh32,
                               void *relu_sh32, int64_t a_nstride, int64_t a_rstride, int64_t a_pstride, int N, int h,
                               int w, int C, void *stream);
/* mivos_maxpool3x3s2 writing SH32 into a zero-bordered buffer (C % 32 == 0). */
int mivos_maxpool3x3s2_sh32(const float *x, void *y_sh32, int64_t y_nstride, int64_t y_rstride, int64_t y_pstride, int N,
                            int H, int W, int C, void *stream);
/* Second half of a 3x3 / pad 1 convolution with ONE output channel (decoder `pred`, prop_net.py:22,29; FusionNet
 * `final_conv`, fusion_net.py:30,49): t [N][H][W][16] holds per input pixel the nine tap products t[p][k] = w[k] . x[p]
 * (a 1x1 projection computed by mivos_conv2d_fused, which reads x exactly once instead of nine times);
 * out[n][y][x] = bias[0] + sum_k t[n][y + k/3 - 1][x + k%3 - 1][k], zero outside the image.  out: planar [N][H*W]. */
int mivos_tap_sum9(const float *t, const float *bias, float *out, int N, int H, int W, void *stream);

/* --------------------------------------------------------------------------------------------
 * FusionNet.forward (model/fusion_net.py:32-50; called per object and frame from inference_core.py:202-217) as one call:
 * x16 [batch][H][W][16] = the channel concatenation of fusion_net.py:38 (im 3, seg1, seg2, attn 2, time 2, 7 zero channels;
 * mivos_interleave_planes builds it) -> logits [batch][H][W] (the caller applies the sigmoid, inference_core.py:214).
 * layer[0..4] = conv1[0], conv2[0], conv2[2], conv3[0], conv3[2] (3x3) as the precision-1 operands of mivos_conv2d_fused
 * (w16 from mivos_pack_weights_f16x3, scale16 = 2^-s per output channel, bias or NULL); final_w = final_conv.weight as
 * fp32 OHWI [1][3][3][32], final_bias = its bias (one float) or NULL.
 * Launches on `stream`: conv1 (mivos_fusion_conv1_planes, or the library's direct 3x3 kernel on x16), mivos_fusion_resblock twice,
 * mivos_fusion_head.
 * scratch: mivos_fusion_net_scratch_floats() floats of device memory; workspace: unused split-K scratch of conv1 (may be NULL).
 *
 * mivos_fusion_resblock - one residual block of the network in ONE launch (fusion_net.py:42-43 / :45-46):
 *     y = relu(x + conv_b(relu(conv_a(x))))      x, y: dense fp32 [batch][H][W][32], x != y
 *   the intermediate never leaves LDS; arithmetic = the f16x3 convolutions of mivos_conv2d_fused (same products, same order).
 * mivos_fusion_head - final_conv (3x3, pad 1, 32 -> 1; fusion_net.py:49) in exact fp32: x [batch][H][W][32] -> logits [batch][H*W].
 * -------------------------------------------------------------------------------------------- */
struct mivos_interleave_desc_s;
typedef struct { const void *w16; const float *scale16; const float *bias; } mivos_fusion_layer;
typedef struct {
  mivos_fusion_layer layer[5];
  const float *final_w;        /* final_conv.weight, fp32 OHWI [1][3][3][32] */
  const float *final_bias;     /* final_conv.bias (one float) or NULL */
  const float *x16;            /* the interleaved input, or NULL when `planes` is given */
  const struct mivos_interleave_desc_s *planes;   /* the nine planar inputs (channel c of sample n = plane[c] + n*nstride[c], NULL
                                                     plane = the constant cval[c]): conv1 gathers them itself, no x16 tensor */
  float *logits;
  float *scratch;
  int64_t scratch_floats;
  int32_t batch, height, width;
  void *workspace;
  int64_t workspace_bytes;
} mivos_fusion_net_desc;
int64_t mivos_fusion_net_scratch_floats(int batch, int height, int width);
int mivos_fusion_net_forward(const mivos_fusion_net_desc *d, void *stream);
int mivos_fusion_resblock(const float *x, float *y, const mivos_fusion_layer *conv_a, const mivos_fusion_layer *conv_b, int batch,
                          int height, int width, void *stream);
int mivos_fusion_head(const float *x, const float *w_ohwi, const float *bias, float *logits, int batch, int height, int width,
                      void *stream);
/* conv1 of the network (9 -> 32, 3x3, ReLU; fusion_net.py:38-40) straight from the planar inputs: y [batch][H][W][32]. */
int mivos_fusion_conv1_planes(const struct mivos_interleave_desc_s *planes, const mivos_fusion_layer *conv1, float *y, int batch,
                              int height, int width, void *stream);

/* Full-softmax read: PropagationNetwork(top_k=None), the reference's "no top-k" configuration (prop_net.py:99-102: softmax over
 * ALL memory positions; :104-108 mem = mv @ affinity).  One pass with a running (max, denominator, numerator) per query, both
 * products on exact fp32 MFMA; same operands as mivos_memory_read_topk; the result goes to `out` rows (may be NULL) and / or to the
 * SH32 activation buffers raw_sh32 / relu_sh32 (x and relu(x), addressed like mivos_memory_read_finalize_sh32; may be NULL).
 * workspace: mivos_memory_read_dense_workspace_bytes() bytes (per-segment partial results). */
int64_t mivos_memory_read_dense_workspace_bytes(int n_obj, int64_t n_mem, int n_q);
int mivos_memory_read_dense(const float *keys, int64_t keys_ostride, const float *values, int64_t values_ostride, const float *qk,
                            float *out, int64_t out_ostride, int64_t out_pstride, void *raw_sh32, void *relu_sh32, int64_t a_nstride,
                            int64_t a_rstride, int64_t a_pstride, int q_width, int n_obj, int64_t n_mem, int n_q, void *workspace,
                            int64_t workspace_bytes, void *stream);

/* Top-k read for ANY k up to 1024 (prop_net.py:133 accepts every top_k; the streaming kernels above hold k <= 64): per chunk of
 * queries the affinity rows go to `workspace` (exact fp32 MFMA), then one workgroup per (object, query) finds the exact k-th largest
 * by a radix select, takes the k survivors (ties: lowest memory positions), softmax (prop_net.py:54-59) and readout.  out rows like
 * mivos_memory_read_topk (may be NULL); idx_out / weight_out [n_obj][n_q][k] optional (survivors in list order, not ranked). */
int64_t mivos_memory_read_topk_any_workspace_bytes(int n_obj, int64_t n_mem, int n_q);
int mivos_memory_read_topk_any(const float *keys, int64_t keys_ostride, const float *values, int64_t values_ostride, const float *qk,
                               float *out, int64_t out_ostride, int64_t out_pstride, int32_t *idx_out, float *weight_out, int n_obj,
                               int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream);

/* --------------------------------------------------------------------------------------------
 * FusionNet training step (model/fusion_model.py:54-131 FusionModel.do_pass; model/losses.py:21-76; train.py:96-124).
 * The forward pass and the data gradients are mivos_conv2d_fused / mivos_fusion_* launches (dgrad of a 3x3 convolution =
 * the 3x3 convolution with transposed, 180-degree-rotated weights); these entry points are the rest of the step:
 *   mivos_fusion_wgrad3x3   dw[n][tap][c] = sum_pixels g[p][n] * x[p + tap][c] (OHWI, like the weights), db[n] = sum_pixels g[p][n]
 *                           for a 3x3 / pad 1 / stride 1 convolution; x [N][H][W][cx] (cx 16 or 32), g [N][H][W][cg] (cg 32 or 1),
 *                           dense fp32; exact fp32 MFMA, fixed summation order (deterministic); scratch:
 *                           mivos_fusion_wgrad_scratch_floats() floats
 *   mivos_fusion_loss       z1, z2 [B][P] (FusionNet logits of object 1 / 2), selector [B][2], cls_gt [B][P] int32 ->
 *                           logits [B][3][P], mask [B][3][P] (aggregate_wbg_channel of sigmoid(z) * selector, aggregate.py:39-53),
 *                           loss [B][P] = per-pixel cross-entropy (3 classes if selector[b][1] > 0.5, else classes {0, 1})
 *   mivos_fusion_kth_loss   BootstrappedCE's top-k: out4[b] = {k[b]-th largest of loss[b][:], #(loss > it), sum(loss > it), #(loss == it)}
 *   mivos_fusion_loss_grad  d total_loss / d z1, d z2 for per-sample pixel weights wsel[b] = {tau, w(loss > tau), w(loss == tau)}
 *   mivos_mul_positive      g[i] = y[i] > 0 ? g[i] : 0 (ReLU backward), n % 4 == 0
 *   mivos_adam_step         torch.optim.Adam (betas, eps, L2 weight_decay, bias correction for `step` >= 1) on flat fp32 vectors
 * -------------------------------------------------------------------------------------------- */
int64_t mivos_fusion_wgrad_scratch_floats(void);
int mivos_fusion_wgrad3x3(const float *x, int cx, const float *g, int cg, float *dw_ohwi, float *db, float *scratch,
                          int64_t scratch_floats, int N, int H, int W, void *stream);
int mivos_fusion_loss(const float *z1, const float *z2, const float *selector, const int32_t *cls_gt, float *logits, float *mask,
                      float *loss, int B, int64_t P, void *stream);
int mivos_fusion_kth_loss(const float *loss, const int32_t *k, float *out4, int B, int64_t P, void *stream);
int mivos_fusion_loss_grad(const float *z1, const float *z2, const float *selector, const int32_t *cls_gt, const float *loss,
                           const float *wsel, float *dz1, float *dz2, int B, int64_t P, void *stream);
int mivos_mul_positive(float *g, const float *y, int64_t n, void *stream);
int mivos_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int step, void *stream);

/* --------------------------------------------------------------------------------------------
 * Space-time memory read: affinity (MFMA) -> streaming per-query top-k -> softmax over the k
 * survivors -> sparse value readout.  Replaces EvalMemoryReader.forward + softmax_w_g_top
 * (prop_net.py:47-63, 81-108) without materialising the [THW x HW] affinity.
 *   keys   [n_obj][n_mem][CK=128]  (n_mem = T*H*W memory positions, row = one position)
 *   values [n_obj][n_mem][CV=512]
 *   qk     [n_q][128]              (shared by all objects, NOT pre-scaled; the kernel applies
 *                                   1/sqrt(128) exactly like prop_net.py:86)
 *   out    [n_obj][n_q] rows of 512 floats at out + o*out_ostride + q*out_pstride
 * workspace: mivos_memory_read_workspace_bytes() bytes of device scratch (the per-segment candidate lists).
 * Returns MIVOS_ERR_TOPK_RANGE when top_k > n_mem (reference raises).  top_k <= 64 (larger k: mivos_memory_read_topk_any;
 * no top-k at all: mivos_memory_read_dense).
 * The read is two launches, also callable one by one (profilers time the affinity/selection kernel alone):
 *   mivos_memory_read_select   - persistent affinity + streaming top-k kernel, candidate lists -> workspace
 *   mivos_memory_read_finalize - exact merge, softmax over the k survivors, value gather -> out
 * mivos_memory_read_topk = select + finalize with the same arguments.
 * -------------------------------------------------------------------------------------------- */
int64_t mivos_memory_read_workspace_bytes(int n_obj, int64_t n_mem, int n_q, int top_k);
int mivos_memory_read_topk(const float *keys, int64_t keys_ostride, const float *values,
                           int64_t values_ostride, const float *qk, float *out, int64_t out_ostride,
                           int64_t out_pstride, int n_obj, int64_t n_mem, int n_q, int top_k,
                           void *workspace, int64_t workspace_bytes, void *stream);
/* How a select launch cuts the work (host logic, no GPU needed; tests / profilers): plan_out[7] = {workgroups, tiles of
 * 32 memory positions per workgroup, candidate lists per stream, tiles per stream, streams (= n_obj * ceil(n_q / queries per
 * workgroup)), entries per list, queries per workgroup}.  f16x3 = 0: mivos_memory_read_select (64 queries per workgroup);
 * 1: mivos_memory_read_select_f16x3 (64; 128 / 256 for long memories, see the two setters below).  Workgroup w takes tiles
 * [w * tiles_per_wg, (w + 1) * tiles_per_wg) of the concatenated streams.  The select kernels leave the plan they used in the
 * first 64 bytes of the workspace, where the finalize kernels read it. */
int mivos_memory_read_plan(int n_obj, int64_t n_mem, int n_q, int top_k, int f16x3, int32_t *plan_out);
/* Tests / tuning: the bank depth (memory positions) from which mivos_memory_read_select_f16x3 uses its 128-queries-per-workgroup
 * kernel (default 400000, or the environment variable MIVOS_MEMREAD_Q128_MIN); returns the previous value, negative = only query. */
int64_t mivos_memory_read_set_q128_min(int64_t n_mem_min);
/* ... and the depth from which it uses the 256-queries-per-workgroup kernel (8 waves, candidate regions in global scratch inside the
 * workspace; default 200000 = it takes precedence over the 128-query kernel, environment variable MIVOS_MEMREAD_Q256_MIN). */
int64_t mivos_memory_read_set_q256_min(int64_t n_mem_min);
/* Experimental (round 3, off by default; environment variable MIVOS_MEMREAD_HIFIRST): the 256-query kernel multiplies a key tile
 * with the hi halves of the split operands first and completes it with the two lo products only where a bound of the dropped terms
 * (largest key norm of the object x the query's norm x 1.25 x 2^-10) cannot rule a candidate out.  Same selection, a third of the
 * matrix work for almost every tile.  Returns the previous setting; negative = only query. */
int mivos_memory_read_set_hifirst(int on);
/* Launch geometry of the persistent select kernels: at most n_wg workgroups (0 = one per CU, the default).  A select workgroup holds
 * 155 KB of LDS, so with one per CU nothing else runs while the launch lasts; a caller that keeps several launch streams busy on the GPU
 * (several clips in flight) sets CUs / streams and the launches of the other streams run beside it (+3.4 % frames/s with two 480p
 * sessions in flight, profiles/r05e_select_workgroups_ab.txt).  The candidate lists follow the work partition; results are identical.
 * Process-wide and not synchronised with launches: a select launch reads it once, and the finalize launch on the same workspace uses the
 * partition that select recorded (not the current value), so changing it between the two is harmless; it must not be changed from another
 * thread WHILE a select call is computing its plan.  Returns the previous value; negative = only query. */
int mivos_memory_read_set_workgroups(int n_wg);
int mivos_memory_read_select(const float *keys, int64_t keys_ostride, const float *qk, int n_obj,
                             int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes,
                             void *stream);
int mivos_memory_read_finalize(const float *values, int64_t values_ostride, float *out,
                               int64_t out_ostride, int64_t out_pstride, int n_obj, int64_t n_mem,
                               int n_q, int top_k, void *workspace, int64_t workspace_bytes, void *stream);
/* mivos_memory_read_finalize writing the readout pre-split for the LDS-DMA convolutions instead of as fp32 rows: SH32 of
 * the 512 channels (`raw_sh32`) and of their ReLU (`relu_sh32`; either may be NULL) at image o, pixel (q / q_width,
 * q % q_width) of zero-bordered buffers with the given strides (floats) - the decoder's first ResBlock (prop_net.py:24,
 * modules.py:28-35) reads relu(x) and x, so no conversion pass is needed between the read and the decoder. */
int mivos_memory_read_finalize_sh32(const float *values, int64_t values_ostride, void *raw_sh32, void *relu_sh32,
                                    int64_t a_nstride, int64_t a_rstride, int64_t a_pstride, int q_width, int n_obj,
                                    int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes,
                                    void *stream);
/* Debug/test export: same selection, but writes the k selected memory indices (ascending score
 * rank, best first) and their normalised softmax weights instead of the readout. */
int mivos_memory_read_topk_indices(const float *keys, int64_t keys_ostride, const float *qk,
                                   int32_t *idx_out, float *weight_out, int n_obj, int64_t n_mem,
                                   int n_q, int top_k, void *workspace, int64_t workspace_bytes,
                                   void *stream);
/* ... and its staged form, after either select call. */
int mivos_memory_read_finalize_indices(int32_t *idx_out, float *weight_out, int n_obj, int64_t n_mem, int n_q, int top_k,
                                       void *workspace, int64_t workspace_bytes, void *stream);
/* The engine's default precision ("f16x3", like the convolutions): the affinity of prop_net.py:85-88 on the fp16 matrix
 * pipe with error compensation - keys and queries are split x = hi + lo (two fp16), each 32-channel step is
 * lo*hi + hi*lo + hi*hi accumulated in fp32 (the same 22-bit operands / fp32 sums as every convolution of the engine).
 * Keys are split ONCE, when a frame is memorised: mivos_memory_split_keys turns fp32 rows [n_obj][n_rows][128] (object stride
 * keys_ostride floats) into the streamed layout (again 512 bytes per row, object stride split_ostride in 4-byte units; block
 * b = 0..3 of 64 halves holds, for ks = 0..3, hi[8] | lo[8] of channels 32 ks + 8 b + e), and mivos_memory_read_select_f16x3
 * is mivos_memory_read_select on those rows (qk stays fp32, same workspace, followed by any of the finalize calls).
 * |key| and |qk| / sqrt(128) must stay below the fp16 range (65504). */
int mivos_memory_split_keys(const float *keys, int64_t keys_ostride, void *keys_split, int64_t split_ostride, int n_obj,
                            int64_t n_rows, void *stream);
int mivos_memory_read_select_f16x3(const void *keys_split, int64_t keys_ostride, const float *qk, int n_obj,
                                   int64_t n_mem, int n_q, int top_k, void *workspace, int64_t workspace_bytes,
                                   void *stream);

/* --------------------------------------------------------------------------------------------
 * Difference-aware attention alignment: W = softmax_m(mk^T qk / sqrt(128)) (T = 1, all positions),
 * out[0][q] = sum_m pos16[m] W[m,q], out[1][q] = sum_m neg16[m] W[m,q]; W never materialised.
 * Replaces AttentionMemory.forward + the two [1 x HW] @ [HW x HW] products in
 * PropagationNetwork.get_attention (prop_net.py:115-129, 187-196) and attn_network.py:12-28,65-79.
 *   mk [n_obj][n_pos][128], qk [n_pos][128], pos16/neg16 [n_obj][n_pos], out [n_obj][2][n_pos]
 * -------------------------------------------------------------------------------------------- */
int mivos_attention_align(const float *mk, const float *qk, const float *pos16, const float *neg16,
                          float *out, int n_obj, int n_pos, void *stream);

/* Dense attention matrix W[o][m][q] = softmax over m of mk[o][m] . qk[q] / sqrt(128): AttentionMemory.forward /
 * PropagationNetwork.get_W (prop_net.py:115-129, 183-185) and attn_network.py:12-28.  The propagation path never needs
 * it (mivos_attention_align folds W into the pos/neg products); exported for callers that want W itself.
 *   mk [n_obj][n_mem][128], qk [n_q][128] per object at qk + o*qk_ostride (0 = one query map shared by all objects),
 *   w [n_obj][n_mem][n_q] dense. */
int mivos_attention_weights(const float *mk, const float *qk, int64_t qk_ostride, float *w, int n_obj, int n_mem,
                            int n_q, void *stream);

/* ---- planar single-channel maps ------------------------------------------------------------- */

/* F.interpolate(mode='area') to 1/16 resolution (prop_net.py:195-196): mean over 16x16 blocks. */
int mivos_area_pool16(const float *x, float *y, int planes, int H, int W, void *stream);

/* F.interpolate(mode='bilinear', align_corners=False) from h x w to H x W on `planes` maps
 * (prop_net.py:30 x4, :198 x16); act = 0 none, 1 sigmoid (prop_net.py:181). */
int mivos_resize_bilinear(const float *x, float *y, int planes, int h, int w, int H, int W, int act,
                          void *stream);

/* aggregate_wbg (model/aggregate.py:22-37): prob [K][P] -> out [K+1][P] (keep_bg=1) or [K][P]. */
int mivos_aggregate_wbg(const float *prob, float *out, int K, int64_t P, int keep_bg, int hard,
                        void *stream);
/* aggregate_sbg (model/aggregate.py:4-20): background fixed at 0.5. */
int mivos_aggregate_sbg(const float *prob, float *out, int K, int64_t P, int keep_bg, int hard,
                        void *stream);
/* aggregate_wbg_channel (model/aggregate.py:39-53; FusionNet training, fusion_model.py:87): prob [B][K][P] ->
 * logits [B][K+1][P] (background first, log(p/(1-p)) of the clamped probabilities, x1000 when hard) and their softmax
 * soft [B][K+1][P] (keep_bg=1) or [B][K][P].  Either output pointer may be NULL.  Any K >= 1. */
int mivos_aggregate_wbg_channel(const float *prob, float *logits, float *soft, int B, int K, int64_t P,
                                int keep_bg, int hard, void *stream);

/* torch.argmax(prob, dim=0) -> uint8 (inference_core.py:259-260, :280); first index wins ties.
 * prob plane c at prob + c*plane_stride. */
int mivos_argmax_u8(const float *prob, int64_t plane_stride, uint8_t *out, int planes, int64_t P,
                    void *stream);

/* pos = clamp(mask - prob, 0, 1), neg = clamp(prob - mask, 0, 1)  (inference_core.py:231-233). */
int mivos_mask_diff(const float *mask, const float *prob, float *pos, float *neg, int64_t n,
                    void *stream);

/* y = 1/(1+exp(-x)) (inference_core.py:214). */
int mivos_sigmoid(const float *x, float *y, int64_t n, void *stream);

/* "others" masks of memorize (prop_net.py:150-157): others[i] = sum_{j != i} masks[j]. */
int mivos_mask_others(const float *masks, float *others, int K, int64_t P, void *stream);

/* Interleave up to 16 planar sources into dense NHWC [N][P][C] (torch.cat along channels of
 * modules.py:54 and fusion_net.py:38, plus zero padding of the channel dimension).
 * channel c of batch n reads plane[c] + n*nstride[c] (NULL plane: the constant cval[c]). */
typedef struct mivos_interleave_desc_s {
  const float *plane[16];
  int64_t nstride[16];
  float cval[16];
  int32_t C;
} mivos_interleave_desc;
int mivos_interleave_planes(const mivos_interleave_desc *d, float *out, int N, int64_t P, void *stream);

/* ResNet stem of the encoders (modules.py:54-58 / 81-83: cat([frame, mask, others]) -> conv1 7x7 / 2 / pad 3 -> bn1 -> relu) in
 * one launch from the PLANAR inputs: channel c of image n = planes->plane[c] + n * planes->nstride[c] (n_planes <= 8, all non-NULL),
 * w_ohwi = fp32 weights [64][7][7][cin] (cin >= n_planes, <= 8; extra input channels have zero weights), the f16x3 arithmetic of
 * mivos_conv2d_fused precision 1 (weights pre-scaled by the power of two `mult`, scale[c] includes 1 / mult and the BN factor),
 * y [N][Ho][Wo][64] fp32 = relu(conv * scale + bias), Ho = (H - 1) / 2 + 1. */
int mivos_stem7x7s2_planes(const mivos_interleave_desc *planes, int n_planes, const float *w_ohwi, int cin, float mult,
                           const float *scale, const float *bias, float *y, int N, int H, int W, void *stream);

/* ---- scribble-to-mask network (model/s2m: DeepLabV3+ / ResNet-50, the step before the path, davis_processor.py:38-70) ----
 * Its convolutions run on mivos_conv2d_fused (dilation field); these are the remaining operators. */

/* F.interpolate(mode='bilinear', align_corners=False) of an NHWC map [N][h][w][C] to H x W, written to
 * y + n*y_nstride + (Y*W + X)*y_pstride + c (a channel slice of the concatenation buffer, _deeplab.py:48-49). */
int mivos_resize_bilinear_nhwc(const float *x, float *y, int64_t y_nstride, int64_t y_pstride, int N, int h, int w, int H,
                               int W, int C, void *stream);
/* nn.AdaptiveAvgPool2d(1) on NHWC (ASPPPooling, _deeplab.py:120-131): x [N][P][C] -> y [N][C]. */
int mivos_global_avgpool(const float *x, float *y, int N, int64_t P, int C, void *stream);
/* cv2.dilate(mask, ones(3,3)) on binary float planes (davis_processor.py:55-60). */
int mivos_dilate3x3(const float *x, float *y, int planes, int H, int W, void *stream);

/* ---- clip ingest (the step before the path: dataset/davis_test_dataset.py:66-110, dataset/yv_test_dataset.py:54-119) ----
 * Outputs are written with explicit plane / row strides and a (top, left) offset, i.e. straight into the interior of the
 * zero-padded [1,T,3,nh,nw] tensor InferenceCore keeps (pad_divide_by, util/tensor_util.py:62-80). */

/* Decoded frames [T][H][W][3] uint8 -> (float(u8)/255 - mean[c]) / std[c] (torchvision ToTensor + Normalize,
 * dataset/range_transform.py:5-8; true divisions, bit-identical), channel c of frame t at out + t*out_tstride +
 * c*out_cstride.  mean3 / std3 are HOST pointers. */
int mivos_ingest_u8(const uint8_t *frames, float *out, int T, int H, int W, int64_t out_tstride, int64_t out_cstride,
                    int64_t out_rstride, int pad_top, int pad_left, const float *mean3, const float *std3, void *stream);
/* F.interpolate(mode='bicubic', align_corners=False) (yv_test_dataset.py:107): planes [P][h][w] -> [P][H][W]. */
int mivos_resize_bicubic(const float *x, float *out, int planes, int h, int w, int H, int W, int64_t out_pstride,
                         int64_t out_rstride, int pad_top, int pad_left, void *stream);
/* Palette-index label map [h][w] uint8 -> n_labels + 1 one-hot float planes (plane 0 = background = none of the labels,
 * plane 1 + k = labels[k]; labels is a device pointer), resized with the 'nearest' rule of yv_test_dataset.py:108
 * (src = floor(dst * in / out)): the `mask` argument of InferenceCore.interact (inference_core.py:219). */
int mivos_onehot_nearest(const uint8_t *label_map, const uint8_t *labels, int n_labels, float *out, int h, int w, int H,
                         int W, int64_t out_pstride, int64_t out_rstride, int pad_top, int pad_left, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MIVOS_HIP_H_ */
