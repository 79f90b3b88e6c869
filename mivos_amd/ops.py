"""Thin tensor-level wrappers over the C ABI (include/mivos_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every arithmetic op below is
one call into libmivos_hip.so with raw device pointers.  Activations are fp32 NHWC tensors
``[N, H, W, C]`` whose channel axis has stride 1; channel slices / batch-strided views (memory-bank
slots) are passed as strides, never copied.  CPU tensors are rejected: there is no fallback.
"""
import contextlib
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, FusionNetDesc, InterleaveDesc, MivosHipError, check

_checked_devices = set()
# "f16x3": error-compensated fp16 MFMA convolutions (3 products per term, fp32-class accuracy, 5.3x the
# fp32-MFMA rate); "f32": exact fp32 MFMA everywhere (the verification path).
CONV_PRECISION = "f16x3"
AFFINITY_PRECISION = None    # None: the memory-read affinity follows CONV_PRECISION; "f32" / "f16x3" pin it (diagnostics: which arithmetic a closed-loop deviation comes from)


def affinity_precision():
    return AFFINITY_PRECISION or CONV_PRECISION


SPLITK_WORKSPACE_BYTES = 64 << 20      # scratch handed to mivos_conv2d_fused for split-K partial tiles
# bench.py sets this to a list to time every conv launch with HIP events on the launch stream:
# entries (kernel variant [+10 for the f16x3 back-end], algorithmic FLOPs = 2*M*Cout*KH*KW*Cin, start event, end event)
PROFILE = None


def _ensure_device(t):
    """Every entry point passes its first operand through here: the tensor must live on the CURRENT HIP device (kernels
    launch on the current device's stream; a tensor of another GPU would be touched from the wrong device, unordered with
    its own stream), and that device must be a gfx950."""
    if not t.is_cuda:
        raise MivosHipError("mivos_amd runs on MI355X (gfx950) only and has no CPU fallback; got a CPU tensor")
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        raise MivosHipError(f"tensor lives on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}: wrap the "
                            f"call in `with ops.on_device(tensor)` (InferenceCore and the network classes do)")
    if idx not in _checked_devices:
        check(_lib.load().mivos_device_check(idx))
        _checked_devices.add(idx)


def on_device(t):
    """Context manager making the device of tensor / torch.device `t` current (no-op when it already is)."""
    dev = t.device if isinstance(t, (torch.Tensor, Act)) else torch.device(t)
    if dev.type != "cuda":
        raise MivosHipError("mivos_amd runs on MI355X (gfx950) only and has no CPU fallback; got a CPU tensor / device")
    return torch.cuda.device(dev)


def _stream():
    return torch.cuda.current_stream().cuda_stream


_PUBLISH = [True]


@contextlib.contextmanager
def single_stream_region():
    """Inside: `publish_constants` is a no-op.  For code that re-packs weights every iteration and runs on ONE stream (the FusionNet
    training step repacks after every optimiser update): stream order alone covers it."""
    old, _PUBLISH[0] = _PUBLISH[0], False
    try:
        yield
    finally:
        _PUBLISH[0] = old


@contextlib.contextmanager
def chip_share(n):
    """Inside: the kernels are told that `n` independent launch streams share this GPU: the persistent select kernels leave the other streams their
    share of the CUs (CUs / n workgroups).  Results are bit-identical for every n (tests/test_gpu_engine.py::test_concurrent_passes_and_suite_lanes_are_bit_identical).
    Process-wide state, set from the one host thread that drives the lanes; not thread-safe."""
    global CHIP_SHARE
    old, CHIP_SHARE = CHIP_SHARE, max(1, int(n))
    lib, old_wgs = None, 0
    if torch.cuda.is_available():           # the persistent select kernels give the other streams their share of the CUs
        lib = _lib.load()
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        old_wgs = lib.mivos_memory_read_set_workgroups(cus // CHIP_SHARE if CHIP_SHARE > 1 else 0)
    try:
        yield
    finally:
        CHIP_SHARE = old
        if lib is not None:
            lib.mivos_memory_read_set_workgroups(old_wgs)


_side_streams = {}


def side_stream(device, purpose):
    """The side HIP stream `purpose` ("fuse", "pass", ...) of the CURRENT stream on `device`, created once per process and reused by every core /
    generator that works under that stream.  A fresh torch.cuda.Stream per InferenceCore (rounds 3-5) meant a fresh allocator pool and fresh
    per-stream workspaces / scratch for every clip: ~30 ms of hipMalloc with the GPU idle at the first fused frame of every session (rocprofv3
    timeline, profiles/r06a_*) and workspaces that were never released (48 GB allocated after two 480p sessions)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, purpose)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def publish_constants():
    """Called once after device-resident constants (packed weights, folded BN vectors, compiled plans) were produced by launches on the
    current stream: waits for the device, so that launches on ANY stream may read them afterwards without a stream dependency (two
    passes / two clips advance on two streams: inference_core.InferenceCore._run_passes, eval_suite.run_suite(lanes=2)).  A one-time
    cost per layer and process (first use), never inside a steady-state step."""
    if _PUBLISH[0]:
        torch.cuda.synchronize()


# LDS-DMA convolutions address their (zero-bordered) input with 32-bit buffer offsets: one tensor < 2 GB
ACT_BYTES_LIMIT = 0x7ff00000


def max_act_batch(h, w, c):
    """How many [h, w, c] images fit one SH32 activation tensor (>= 1; callers chunk their batch to this)."""
    return max(1, ACT_BYTES_LIMIT // ((h + 2) * (w + 2) * c * 4))


def _f32(t):
    if t.dtype != torch.float32:
        raise MivosHipError(f"expected float32, got {t.dtype}")
    return t


def _nhwc_strides(t):
    """(nstride, pstride) of an NHWC view with dense rows; raises if the view is not expressible."""
    n, h, w, c = t.shape
    sn, sh, sw, sc = t.stride()
    if c > 1 and sc != 1:
        raise MivosHipError("NHWC tensor must have channel stride 1")
    if h > 1 and sh != w * sw:
        raise MivosHipError("NHWC tensor must have dense rows (stride_h == W * stride_w)")
    return (sn if n > 1 else h * w * sw), sw


class ConvLayer:
    """Device-resident packed convolution: OHWI weights (+ folded BN scale / bias)."""
    __slots__ = ("w", "scale", "bias", "cin", "cout", "k", "stride", "pad", "split", "w16", "scale16", "wdma", "proj", "dil", "mult16")

    def __init__(self, w_ohwi, scale, bias, stride, pad, split=None, dil=1):
        self.w = w_ohwi.contiguous()
        self.cout, self.k, _, self.cin = w_ohwi.shape
        self.scale, self.bias = scale, bias
        self.stride, self.pad, self.dil = stride, pad, dil
        self.split = self.cout if split is None else split
        self.w16 = self.scale16 = self.wdma = self.proj = self.mult16 = None

    def projection(self):
        """A 3x3 / pad 1 layer with ONE output channel as a 1x1 layer with 16 outputs, row k = the weights of tap k
        (9 real rows): the GEMM kernels then read the input once and `tap_sum9` adds the nine shifted products."""
        if self.proj is None:
            assert self.cout == 1 and self.k == 3 and self.scale is None and self.dil == 1
            w = self.w.new_zeros((16, 1, 1, self.cin))
            w[:9, 0, 0] = self.w[0].reshape(9, self.cin)
            self.proj = ConvLayer(w, None, None, 1, 0)
            publish_constants()
        return self.proj

    def slice_cin(self, c0, c1, keep_bias):
        """The same layer restricted to input channels [c0, c1) (a convolution over concatenated inputs is the sum of the
        convolutions over the parts; the bias / BN shift goes with one part)."""
        assert self.scale is None and self.split == self.cout
        return ConvLayer(self.w[..., c0:c1].contiguous(), None, self.bias if keep_bias else None, self.stride, self.pad, dil=self.dil)

    def _f16x3_scale(self):
        """(2^s, scale * 2^-s): the exact power-of-two pre-scaling of the fp16 hi/lo weight split."""
        import math
        wmax = float(self.w.abs().max())
        s = 14 - math.floor(math.log2(wmax)) if wmax > 0 else 0     # max |w * 2^s| in [2^14, 2^15)
        mult = 2.0 ** s
        base = self.scale if self.scale is not None else torch.ones(self.cout, dtype=torch.float32, device=self.w.device)
        return mult, (base * (1.0 / mult)).contiguous()

    def split_scale(self):
        """(2^s, epilogue scale incl. 2^-s) of the fp16 hi/lo weight split, computed once (kernels that split the fp32 weights
        themselves: mivos_stem7x7s2_planes)."""
        if self.mult16 is None:
            self.mult16, sc = self._f16x3_scale()
            if self.scale16 is None:
                self.scale16 = sc
            publish_constants()
        return self.mult16, self.scale16

    def dma(self):
        """(weights packed for the LDS-DMA kernels (precision 2), epilogue scale); needs cin % 32 == 0."""
        if self.wdma is None:
            _ensure_device(self.w)
            lib = _lib.load()
            mult, scale16 = self._f16x3_scale()
            wd = torch.empty(lib.mivos_pack_weights_f16x3_dma_bytes(self.cout, self.k, self.k, self.cin), dtype=torch.uint8, device=self.w.device)
            check(lib.mivos_pack_weights_f16x3_dma(self.w.data_ptr(), wd.data_ptr(), self.cout, self.k, self.k, self.cin, mult, _stream()))
            self.wdma = wd
            if self.scale16 is None:
                self.scale16 = scale16
            publish_constants()
        return self.wdma, self.scale16

    def f16x3(self):
        """(packed hi/lo fp16 weights, epilogue scale incl. the 2^-s of the weight pre-scaling); built on
        first use on the GPU the layer lives on."""
        if self.w16 is None:
            _ensure_device(self.w)
            ktot = self.k * self.k * self.cin
            kpad = (ktot + 63) // 64 * 64
            mult, scale16 = self._f16x3_scale()
            w16 = torch.empty(self.cout * kpad * 4, dtype=torch.uint8, device=self.w.device)
            check(_lib.load().mivos_pack_weights_f16x3(self.w.data_ptr(), w16.data_ptr(), self.cout, self.k, self.k, self.cin, mult, _stream()))
            self.w16, self.scale16 = w16, scale16
            publish_constants()
        return self.w16, self.scale16

    @staticmethod
    def pack(weight, bias=None, bn=None, stride=1, pad=0, cin_pad=None, eps=1e-5, dilation=1):
        """weight [Cout,Cin,k,k] (+conv bias) (+ eval BatchNorm (gamma, beta, mean, var)) ->
        y = conv(x, w) * scale + bias'.   BN(conv + b) = conv*s + (b - mean)*s + beta, s = gamma/sqrt(var+eps)."""
        w = weight.detach().float()
        cout, cin = w.shape[:2]
        if cin_pad is not None and cin_pad > cin:
            w = torch.cat([w, w.new_zeros(cout, cin_pad - cin, *w.shape[2:])], 1)
        w = w.permute(0, 2, 3, 1).contiguous()
        b = bias.detach().float() if bias is not None else None
        if bn is not None:
            gamma, beta, mean, var = (t.detach().float() for t in bn)
            s = gamma / torch.sqrt(var + eps)
            b2 = beta - mean * s if b is None else (b - mean) * s + beta
            return ConvLayer(w, s.contiguous(), b2.contiguous(), stride, pad, dil=dilation)
        return ConvLayer(w, None, None if b is None else b.contiguous(), stride, pad, dil=dilation)

    @staticmethod
    def fuse_outputs(a, b):
        """One GEMM for two convolutions of the same input (KeyValue: key_proj | val_proj)."""
        assert a.k == b.k and a.cin == b.cin and a.stride == b.stride and a.scale is None and b.scale is None
        bias = None
        if a.bias is not None:
            bias = torch.cat([a.bias, b.bias]).contiguous()
        return ConvLayer(torch.cat([a.w, b.w], 0), None, bias, a.stride, a.pad, split=a.cout)

    def to(self, device):
        for n in ("w", "scale", "bias"):
            v = getattr(self, n)
            if v is not None:
                setattr(self, n, v.to(device))
        self.w16 = self.scale16 = self.wdma = self.proj = self.mult16 = None
        return self


class Act:
    """Activation tensor of the LDS-DMA convolution path (csrc/conv_f16x3_dma.hip): "SH32" layout (per pixel and 32
    channels 32 fp16 hi | 32 fp16 lo, the same 4 bytes per element as fp32) stored inside a buffer with a one-pixel
    border of zeros, [N, H+2, W+2, C] float32-sized elements, so that 3x3 / pad-1 im2col needs no masks."""
    __slots__ = ("buf", "n", "h", "w", "c")
    BORDER = 1

    def __init__(self, buf, n, h, w, c):
        self.buf, self.n, self.h, self.w, self.c = buf, n, h, w, c

    @property
    def shape(self):
        return (self.n, self.h, self.w, self.c)

    @property
    def device(self):
        return self.buf.device

    def interior_ptr(self):
        return self.buf.data_ptr() + 4 * ((self.w + 2) * self.c + self.c)

    def strides(self):
        """(image, row, pixel) strides in floats."""
        return (self.h + 2) * (self.w + 2) * self.c, (self.w + 2) * self.c, self.c

    def __getitem__(self, sl):
        """Batch slice (view of the same storage)."""
        sub = self.buf[sl]
        return Act(sub, sub.shape[0], self.h, self.w, self.c)


import collections

_act_scratch = collections.OrderedDict()
ACT_SCRATCH_ENTRIES = 256   # scratch buffers kept per process (least recently used ones go first)
CHIP_SHARE = 1          # independent launch streams the caller keeps busy on this GPU (lanes of run_suite / bench, the two passes of an interaction):
                        # the persistent select kernels take CUs / CHIP_SHARE workgroups (exact selection: same results); also passed to every convolution as
                        # mivos_conv_desc.chip_share, which since round 6 does not enter any decision that changes the arithmetic (results are bit-identical for every value)
COUT1_PROJECTION = True # one-output-channel 3x3 layers as a 1x1 projection to nine tap products + tap_sum9 (False: the generic kernels; diagnostics)
USE_ACT_PATH = True     # run conv -> conv edges on the LDS-DMA kernels (needs CONV_PRECISION == "f16x3")


def act_path():
    return USE_ACT_PATH and CONV_PRECISION == "f16x3"



def alloc_act(n, h, w, c, device, tag=None):
    """Zero-bordered SH32 buffer.  tag=None: fresh storage (for tensors the caller keeps); otherwise a scratch buffer
    cached per (tag, shape, device, stream) whose interior the next producer overwrites completely (the border is zeroed once).
    Sharing scratch between call sites - and between several InferenceCores - is safe because every use is stream ordered: a
    scratch tensor is written by one launch and read by launches enqueued right after it on the SAME stream, so a later
    producer (of whichever core) is ordered behind the last reader.  Nothing that outlives the enclosing network call may live in
    scratch: cached query features, bank slots and public results use tag=None / their own tensors."""
    if c % 32:
        raise MivosHipError(f"SH32 activations need a multiple of 32 channels, got {c}")
    if tag is None:
        return Act(torch.zeros((n, h + 2, w + 2, c), dtype=torch.float32, device=device), n, h, w, c)
    key = (tag, n, h, w, c, device.index, torch.cuda.current_stream().cuda_stream)
    a = _act_scratch.get(key)
    if a is None:
        # many different clip sizes in one process (config 4's suite): the least recently used buffer goes, the ones in use stay
        # (an evicted buffer that a live Act still references is kept alive by that reference; a new one is zeroed on allocation,
        # so the "border is zero" invariant holds for every buffer handed out)
        while len(_act_scratch) >= ACT_SCRATCH_ENTRIES:
            _act_scratch.popitem(last=False)
        a = _act_scratch[key] = Act(torch.zeros((n, h + 2, w + 2, c), dtype=torch.float32, device=device), n, h, w, c)
    else:
        _act_scratch.move_to_end(key)
    return a


def to_act(x, relu=False, tag=None, out=None):
    """fp32 NHWC view -> Act (optionally through ReLU: the DMA-fed kernels cannot apply relu_in)."""
    _ensure_device(x)
    n, h, w, c = x.shape
    a = out if out is not None else alloc_act(n, h, w, c, x.device, tag)
    assert a.shape == (n, h, w, c)
    if x.stride(3) != 1:
        x = x.contiguous()
    sn, sh, sw, _ = x.stride()
    sw = sw if w > 1 else c                 # strides of size-1 dimensions are arbitrary in torch: use the dense ones
    sh = sh if h > 1 else w * sw
    sn = sn if n > 1 else h * sh
    an, ar, ap = a.strides()
    check(_lib.load().mivos_pack_activation_sh32(_f32(x).data_ptr(), sn, sh, sw, a.interior_ptr(), an, ar, ap, n, h, w, c, int(relu), _stream()))
    return a


def to_f32(a):
    """Act -> dense fp32 NHWC tensor (x = hi + lo)."""
    y = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    an, ar, ap = a.strides()
    n, h, w, c = a.shape
    check(_lib.load().mivos_unpack_activation_sh32(a.interior_ptr(), an, ar, ap, y.data_ptr(), h * w * c, w * c, c, n, h, w, c, _stream()))
    return y


def conv(x, L, relu_in=False, relu_out=False, res=None, out=None, out2=None, out_act=False, tag=None, bias=None):
    """y = act(conv(act_in(x)) * scale + bias + res).  x: fp32 [N,H,W,Cin] view or an Act (then the LDS-DMA kernels run);
    res: fp32 view or Act; out_act=True returns an Act (only from Act inputs; `out` may be a preallocated Act, `tag` selects
    a scratch buffer); bias: per-call replacement of the layer's bias vector ([Cout] fp32 on the device; the packed weights
    and their caches stay the layer's).  Returns out (and out2 when the layer is split, i.e. (out, out2))."""
    if isinstance(x, Act) and L.dil > 1:
        x = to_f32(x)                       # atrous convolutions run on the register-staged kernels (fp32 input, bounds masks)
    from_act = isinstance(x, Act)
    if not from_act:
        _ensure_device(x)
    n, h, w, cin = x.shape
    dev = x.device
    if cin != L.cin:
        raise MivosHipError(f"conv: input has {cin} channels, layer expects {L.cin}")
    if from_act and (relu_in or L.cout == 1 or CONV_PRECISION != "f16x3"):
        raise MivosHipError("conv: an Act input needs the f16x3 back-end, Cout > 1 and relu applied by its producer")
    if out_act and not from_act:
        raise MivosHipError("conv: SH32 outputs are written by the LDS-DMA kernels only (Act input)")
    if (L.cout == 1 and L.k == 3 and L.stride == 1 and L.pad == 1 and L.dil == 1 and L.scale is None and CONV_PRECISION == "f16x3" and COUT1_PROJECTION
            and not from_act and res is None and not relu_out and cin % 32 == 0):
        # one output channel: 1x1 projection to the nine tap products (reads x once) + 9-point sum, instead of a dot-product
        # kernel that re-reads every pixel for each of its nine neighbours
        t = conv(x, L.projection(), relu_in=relu_in)
        if out is None:
            out = torch.empty((n, h, w, 1), dtype=torch.float32, device=dev)
        assert out.is_contiguous() and out.numel() == n * h * w
        check(_lib.load().mivos_tap_sum9(t.data_ptr(), L.bias.data_ptr() if L.bias is not None else None, out.data_ptr(), n, h, w, _stream()))
        return out
    ho = (h + 2 * L.pad - L.dil * (L.k - 1) - 1) // L.stride + 1
    wo = (w + 2 * L.pad - L.dil * (L.k - 1) - 1) // L.stride + 1
    dual = L.split < L.cout
    if out is None:
        out = alloc_act(n, ho, wo, L.split, dev, tag) if out_act else torch.empty((n, ho, wo, L.split), dtype=torch.float32, device=dev)
    if dual and out2 is None:
        out2 = torch.empty((n, ho, wo, L.cout - L.split), dtype=torch.float32, device=dev)
    d = ConvDesc()
    if from_act:
        wd, scale16 = L.dma()
        d.x, d.w, d.scale, d.precision = x.interior_ptr(), wd.data_ptr(), scale16.data_ptr(), 2
        d.x_nstride, d.x_rstride, d.x_pstride = x.strides()
        d.x_border, d.x_format = Act.BORDER, 1
    elif CONV_PRECISION == "f16x3" and L.cout > 1:
        w16, scale16 = L.f16x3()
        d.x, d.w, d.scale, d.precision = _f32(x).data_ptr(), w16.data_ptr(), scale16.data_ptr(), 1
        d.x_nstride, d.x_pstride = _nhwc_strides(x)
    else:
        d.x, d.w, d.precision = _f32(x).data_ptr(), L.w.data_ptr(), 0
        d.scale = L.scale.data_ptr() if L.scale is not None else None
        d.x_nstride, d.x_pstride = _nhwc_strides(x)
    if bias is not None:
        assert bias.shape == (L.cout,) and bias.dtype == torch.float32 and bias.is_contiguous() and bias.device == dev
        d.bias = bias.data_ptr()
    else:
        d.bias = L.bias.data_ptr() if L.bias is not None else None
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = n, h, w, cin, L.cout, L.k, L.k
    d.stride, d.pad, d.Ho, d.Wo, d.split, d.dilation = L.stride, L.pad, ho, wo, L.split, L.dil
    d.relu_in, d.relu_out = int(relu_in), int(relu_out)
    d.chip_share = CHIP_SHARE
    assert tuple(out.shape) == (n, ho, wo, L.split), (out.shape, (n, ho, wo, L.split))
    if isinstance(out, Act):
        d.y, d.y_format = out.interior_ptr(), 1
        d.y_nstride, d.y_rstride, d.y_pstride = out.strides()
    else:
        d.y = out.data_ptr()
        d.y_nstride, d.y_pstride = _nhwc_strides(out)
    if dual:
        assert out2.shape == (n, ho, wo, L.cout - L.split)
        d.y2 = out2.data_ptr()
        d.y2_nstride, d.y2_pstride = _nhwc_strides(out2)
    if res is not None:
        assert tuple(res.shape[1:]) == (ho, wo, L.cout) and res.shape[0] in (1, n)
        if isinstance(res, Act):
            d.res, d.res_format = res.interior_ptr(), 1
            rn, d.res_rstride, rp = res.strides()
        else:
            d.res = _f32(res).data_ptr()
            rn, rp = _nhwc_strides(res)
        d.res_nstride, d.res_pstride = (0 if (res.shape[0] == 1 and n > 1) else rn), rp
    ws = _workspace(SPLITK_WORKSPACE_BYTES, dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() - STATUS_BYTES
    # fp16-range guard of the f16x3 epilogues (check_activation_range): the owner's word, else the tail of this stream's workspace
    d.status = RANGE_STATUS.data_ptr() if (RANGE_STATUS is not None and RANGE_STATUS.device == dev) else ws.data_ptr() + ws.numel() - STATUS_BYTES
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.load().mivos_conv2d_fused(C.byref(d), _stream()))
    if PROFILE is not None:
        ev1.record()
        m = n * ho * wo
        lib = _lib.load()
        if d.precision == 2:
            var = lib.mivos_conv2d_variant_pp(m, L.cout, L.k * L.k * cin // 32)
        elif d.precision == 1:
            var = lib.mivos_conv2d_variant_f16x3(m, L.cout) + 10
            if L.cout == 32 and L.k == 3 and L.stride == 1 and cin in (16, 32):
                var = 19
        else:
            var = lib.mivos_conv2d_variant(m, L.cout)
        PROFILE.append((var, 2.0 * m * L.cout * L.k * L.k * cin, ev0, ev1, (m, cin, L.cout, L.k, L.stride, res is not None)))
    return (out, out2) if dual else out


def _fusion_layer(L):
    w16, scale16 = L.f16x3()
    fl = _lib.FusionLayer()
    fl.w16, fl.scale16, fl.bias = w16.data_ptr(), scale16.data_ptr(), (L.bias.data_ptr() if L.bias is not None else None)
    return fl


def fusion_net_forward(x16, layers, final, planes=None, shape=None):
    """FusionNet.forward as one C-ABI call (mivos_fusion_net_forward): x16 [B,H,W,16] fp32 - or, instead, the nine planar inputs
    `planes` = [(tensor_or_float, batch_stride)] x 9 with shape = (B, H, W): conv1 then gathers them itself and no x16 tensor
    exists; layers = the five packed 3x3 ConvLayers (conv1[0], conv2[0], conv2[2], conv3[0], conv3[2]), final = final_conv's
    ConvLayer -> logits [B,H,W,1].  The f16x3 back-end only (FusionNet.run(layered=True) serves "f32")."""
    if planes is not None:
        b, h, w = shape
        pd, keep = _interleave_desc(planes, 16)
        assert len(planes) == 9
        dev = final.w.device
    else:
        _ensure_device(x16)
        b, h, w, c = x16.shape
        assert c == 16 and x16.is_contiguous()
        dev = x16.device
    assert CONV_PRECISION == "f16x3" and final.cout == 1 and final.cin == 32 and final.scale is None
    lib = _lib.load()
    d = FusionNetDesc()
    for i, L in enumerate(layers):
        d.layer[i] = _fusion_layer(L)
    d.final_w = final.w.data_ptr()
    d.final_bias = final.bias.data_ptr() if final.bias is not None else None
    out = torch.empty((b, h, w, 1), dtype=torch.float32, device=dev)
    n = lib.mivos_fusion_net_scratch_floats(b, h, w)
    scratch = _workspace(4 * n, dev, "fusion_net")
    ws = _workspace(SPLITK_WORKSPACE_BYTES, dev)
    if planes is not None:
        d.x16, d.planes = None, C.cast(C.pointer(pd), C.c_void_p)
    else:
        d.x16, d.planes = _f32(x16).data_ptr(), None
    d.logits, d.scratch, d.scratch_floats = out.data_ptr(), scratch.data_ptr(), n
    d.batch, d.height, d.width = b, h, w
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() - STATUS_BYTES      # (the tail of the split-K workspace is the status block)
    check(lib.mivos_fusion_net_forward(C.byref(d), _stream()))
    return out


def fusion_conv1_planes(planes, shape, conv1):
    """conv1 of FusionNet (9 -> 32, ReLU) straight from the nine planar inputs (mivos_fusion_conv1_planes) -> [B,H,W,32]."""
    b, h, w = shape
    pd, keep = _interleave_desc(planes, 16)
    assert len(planes) == 9 and (conv1.cin, conv1.cout, conv1.k) == (16, 32, 3)
    la = _fusion_layer(conv1)
    out = torch.empty((b, h, w, 32), dtype=torch.float32, device=conv1.w.device)
    check(_lib.load().mivos_fusion_conv1_planes(C.byref(pd), C.byref(la), out.data_ptr(), b, h, w, _stream()))
    return out


def fusion_resblock(x, conv_a, conv_b, out=None):
    """relu(x + conv_b(relu(conv_a(x)))) in one launch (mivos_fusion_resblock): x dense fp32 [B,H,W,32], conv_a / conv_b
    packed 3x3 / pad 1 ConvLayers 32 -> 32 without BN scale."""
    _ensure_device(x)
    b, h, w, c = x.shape
    assert c == 32 and x.is_contiguous() and x.dtype == torch.float32
    for L in (conv_a, conv_b):
        assert (L.cin, L.cout, L.k, L.stride, L.pad, L.dil) == (32, 32, 3, 1, 1, 1) and L.scale is None
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.is_contiguous() and out.data_ptr() != x.data_ptr()
    la, lb = _fusion_layer(conv_a), _fusion_layer(conv_b)
    check(_lib.load().mivos_fusion_resblock(x.data_ptr(), out.data_ptr(), C.byref(la), C.byref(lb), b, h, w, _stream()))
    return out


def fusion_head(x, final):
    """final_conv of FusionNet (3x3, pad 1, 32 -> 1) in exact fp32 (mivos_fusion_head): x [B,H,W,32] -> [B,H,W,1]."""
    _ensure_device(x)
    b, h, w, c = x.shape
    assert c == 32 and x.is_contiguous() and final.cout == 1 and final.cin == 32 and final.k == 3 and final.scale is None
    out = torch.empty((b, h, w, 1), dtype=torch.float32, device=x.device)
    check(_lib.load().mivos_fusion_head(x.data_ptr(), final.w.data_ptr(), final.bias.data_ptr() if final.bias is not None else None,
                                        out.data_ptr(), b, h, w, _stream()))
    return out


def stem_planes(planes, n, H, W, L):
    """ResNet stem (7x7 / 2 / pad 3 conv + folded BN + ReLU, 64 channels) straight from planar inputs (mivos_stem7x7s2_planes):
    planes = [(tensor, batch_stride)] (<= 8, no constants), L = the packed stem ConvLayer -> fp32 [n, H/2, W/2, 64]."""
    assert (L.k, L.stride, L.pad, L.cout, L.dil) == (7, 2, 3, 64, 1) and len(planes) <= L.cin <= 8 and CONV_PRECISION == "f16x3"
    d, keep = _interleave_desc(planes, 8)
    mult, scale16 = L.split_scale()
    out = torch.empty((n, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), dtype=torch.float32, device=L.w.device)
    check(_lib.load().mivos_stem7x7s2_planes(C.byref(d), len(planes), L.w.data_ptr(), L.cin, mult, scale16.data_ptr(),
                                              L.bias.data_ptr() if L.bias is not None else None, out.data_ptr(), n, H, W, _stream()))
    return out


# ---- FusionNet training step (csrc/fusion_train.hip; reference model/fusion_model.py:54-131, model/losses.py) -----------------

def fusion_wgrad3x3(x, g):
    """Weight / bias gradient of a 3x3, pad 1, stride 1 convolution y = conv(x): x [N,H,W,cx] (cx 16 / 32), g = dL/dy [N,H,W,cg]
    (cg 32 / 1), dense fp32 -> (dw OHWI [cg,3,3,cx], db [cg]).  Exact fp32 MFMA, deterministic."""
    _ensure_device(x)
    n, h, w, cx = x.shape
    cg = g.shape[3]
    assert x.is_contiguous() and g.is_contiguous() and g.shape[:3] == x.shape[:3] and x.dtype == g.dtype == torch.float32
    lib = _lib.load()
    dw = torch.empty((cg, 3, 3, cx), dtype=torch.float32, device=x.device)
    db = torch.empty((cg,), dtype=torch.float32, device=x.device)
    nf = lib.mivos_fusion_wgrad_scratch_floats()
    scratch = _workspace(4 * nf, x.device, "wgrad")
    check(lib.mivos_fusion_wgrad3x3(x.data_ptr(), cx, g.data_ptr(), cg, dw.data_ptr(), db.data_ptr(), scratch.data_ptr(), nf, n, h, w, _stream()))
    return dw, db


def fusion_loss(z1, z2, selector, cls_gt):
    """z1, z2 [B,P] FusionNet logits of object 1 / 2, selector [B,2], cls_gt [B,P] int32 -> (logits [B,3,P], mask [B,3,P] =
    aggregate_wbg_channel(sigmoid(z) * selector, keep_bg=True), per-pixel cross-entropy [B,P])."""
    _ensure_device(z1)
    b, p = z1.shape
    z1, z2, selector = _f32(z1).contiguous(), _f32(z2).contiguous(), _f32(selector).contiguous()
    assert cls_gt.dtype == torch.int32 and cls_gt.is_contiguous() and cls_gt.shape == (b, p) and selector.shape == (b, 2)
    logits = torch.empty((b, 3, p), dtype=torch.float32, device=z1.device)
    mask, loss = torch.empty_like(logits), torch.empty((b, p), dtype=torch.float32, device=z1.device)
    check(_lib.load().mivos_fusion_loss(z1.data_ptr(), z2.data_ptr(), selector.data_ptr(), cls_gt.data_ptr(), logits.data_ptr(), mask.data_ptr(),
                                        loss.data_ptr(), b, p, _stream()))
    return logits, mask, loss


def fusion_kth_loss(loss, k):
    """loss [B,P], k [B] int32 -> [B,4] = (k-th largest, #(loss > it), sum(loss > it), #(loss == it)) per sample."""
    _ensure_device(loss)
    b, p = loss.shape
    assert loss.is_contiguous() and k.dtype == torch.int32 and k.shape == (b,)
    out = torch.empty((b, 4), dtype=torch.float32, device=loss.device)
    check(_lib.load().mivos_fusion_kth_loss(loss.data_ptr(), k.data_ptr(), out.data_ptr(), b, p, _stream()))
    return out


def fusion_loss_grad(z1, z2, selector, cls_gt, loss, wsel):
    """d total_loss / d z1, d z2 [B,P]; wsel [B,3] = (tau, weight of a pixel with loss > tau, weight of a pixel with loss == tau)."""
    _ensure_device(z1)
    b, p = z1.shape
    z1, z2, selector, wsel = _f32(z1).contiguous(), _f32(z2).contiguous(), _f32(selector).contiguous(), _f32(wsel).contiguous()
    assert wsel.shape == (b, 3) and loss.is_contiguous() and cls_gt.dtype == torch.int32
    dz1, dz2 = torch.empty_like(z1), torch.empty_like(z1)
    check(_lib.load().mivos_fusion_loss_grad(z1.data_ptr(), z2.data_ptr(), selector.data_ptr(), cls_gt.data_ptr(), loss.data_ptr(), wsel.data_ptr(),
                                             dz1.data_ptr(), dz2.data_ptr(), b, p, _stream()))
    return dz1, dz2


def mul_positive(g, y):
    """In place g *= (y > 0): ReLU backward."""
    _ensure_device(g)
    assert g.is_contiguous() and y.is_contiguous() and g.shape == y.shape and g.dtype == y.dtype == torch.float32
    check(_lib.load().mivos_mul_positive(g.data_ptr(), y.data_ptr(), g.numel(), _stream()))
    return g


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step):
    """torch.optim.Adam's update on flat fp32 vectors (in place)."""
    _ensure_device(param)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() == param.numel()
    check(_lib.load().mivos_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(), float(lr),
                                      float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), _stream()))


def maxpool3x3s2(x, act_tag=None, as_act=False):
    """MaxPool 3x3 / 2 / pad 1.  as_act=True writes the result straight into an SH32 Act (scratch `act_tag`)."""
    _ensure_device(x)
    n, h, w, c = x.shape
    assert x.is_contiguous()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if as_act:
        a = alloc_act(n, ho, wo, c, x.device, act_tag)
        an, ar, ap = a.strides()
        check(_lib.load().mivos_maxpool3x3s2_sh32(x.data_ptr(), a.interior_ptr(), an, ar, ap, n, h, w, c, _stream()))
        return a
    y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    check(_lib.load().mivos_maxpool3x3s2(x.data_ptr(), y.data_ptr(), n, h, w, c, _stream()))
    return y


def upsample2x_add_acts(skip, up, tag):
    """skip [1 or N, 2h, 2w, C] + bilinear_x2(up [N, h, w, C]) -> (SH32 Act of the sum, SH32 Act of its ReLU), scratch
    buffers `tag`: the two operands a pre-activation ResBlock reads (raw for its skip path, relu for conv1)."""
    _ensure_device(up)
    n, h, w, c = up.shape
    assert up.is_contiguous() and skip.is_contiguous() and skip.shape[1:] == (2 * h, 2 * w, c)
    raw, rel = alloc_act(n, 2 * h, 2 * w, c, up.device, (tag, "raw")), alloc_act(n, 2 * h, 2 * w, c, up.device, (tag, "relu"))
    sn = 0 if (skip.shape[0] == 1 and n > 1) else skip.stride(0)
    an, ar, ap = raw.strides()
    check(_lib.load().mivos_upsample2x_add_multi(skip.data_ptr(), sn, up.data_ptr(), None, raw.interior_ptr(), rel.interior_ptr(), an, ar, ap,
                                                 n, h, w, c, _stream()))
    return raw, rel


def upsample2x_add(skip, up):
    """skip [1 or N, 2h, 2w, C] + bilinear_x2(up [N, h, w, C])."""
    _ensure_device(up)
    n, h, w, c = up.shape
    assert up.is_contiguous() and skip.is_contiguous() and skip.shape[1:] == (2 * h, 2 * w, c)
    out = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=up.device)
    sn = 0 if (skip.shape[0] == 1 and n > 1) else skip.stride(0)
    check(_lib.load().mivos_upsample2x_add(skip.data_ptr(), sn, up.data_ptr(), out.data_ptr(), n, h, w, c, _stream()))
    return out


_ws_cache = {}


STATUS_BYTES = 64       # tail of every split-K workspace: status words the kernels raise (word 0 bit 0: an f16x3 output left the fp16 range)


def _workspace(nbytes, device, purpose="splitk"):
    key = (purpose, device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        ws[-STATUS_BYTES:].zero_()
        _ws_cache[key] = ws
    return ws


RANGE_STATUS = None     # int32 device tensor (>= 1 word) in which the f16x3 convolution epilogues launched from now on raise their fp16-range flag
                        # (mivos_conv_desc.status); None: the tail of the launching stream's split-K workspace.  Owners (InferenceCore, FusionGenerator,
                        # the S2M network) point it at THEIR OWN word for the duration of a step (range_status) and check that word only, so that
                        # several cores in flight (lanes) never read or clear each other's evidence.


def new_range_status(device):
    return torch.zeros(STATUS_BYTES // 4, dtype=torch.int32, device=device)


@contextlib.contextmanager
def range_status(word):
    """Inside: convolutions raise their fp16-range flag in `word` (see RANGE_STATUS).  Not to be held across a generator's yield."""
    global RANGE_STATUS
    old, RANGE_STATUS = RANGE_STATUS, word
    try:
        yield
    finally:
        RANGE_STATUS = old


def check_activation_range(target):
    """Raise MivosHipError if an f16x3 convolution produced an output beyond the fp16 range (|y| > 65504: the next layer's hi / lo operand split
    would turn it into inf, and the ReLUs / `aggregate_wbg`'s clamp downstream would turn the NaNs that follow into finite numbers - silently).
    `target`: the status tensor an owner passed to `range_status` for its launches (new_range_status; read with one 4-byte copy on the current
    stream - the caller's launches must be on it or joined into it - and cleared) or, for code that launched convolutions outside any
    `range_status` region, a device: then the tails of that device's split-K workspaces are swept (every stream's; ordered only against the
    current stream).  InferenceCore checks its own word once per interaction, next to the mask download it waits for anyway.  Covers the
    convolution epilogues (trunks, KeyValue, decoder, S2M) - the stems, FusionNet's own kernels and the pointwise kernels are not instrumented
    (INTEGRATION.md "Limits")."""
    bad = False
    if isinstance(target, torch.Tensor):
        if int(target[0].item()) != 0:
            bad = True
            target.zero_()
    else:
        device = torch.device(target)
        for (purpose, index, _stream_id), ws in list(_ws_cache.items()):
            if purpose != "splitk" or index != device.index:
                continue
            word = ws[-STATUS_BYTES:-STATUS_BYTES + 4].view(torch.int32)
            if int(word.item()) != 0:
                bad = True
                word.zero_()
    if bad:
        raise MivosHipError("an activation left the fp16 range (|y| > 65504) inside an f16x3 convolution: the default precision carries operands as fp16 hi + lo "
                            "pairs (INTEGRATION.md 'Limits').  Results of this interaction are invalid; use ops.CONV_PRECISION = 'f32' for these weights.")


def _rows(t, width):
    """[n_obj, n_rows, width] view whose rows are dense -> (tensor, object stride)."""
    assert t.dim() == 3 and t.shape[2] == width
    if t.stride(2) != 1 or t.stride(1) != width:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1] * width)


def split_keys(keys, out=None):
    """fp32 key rows [K, ..., 128] (dense rows, any object stride) -> the pre-split rows mivos_memory_read_select_f16x3 streams
    (same shape / dtype container, bits are packed fp16 hi/lo pairs).  InferenceCore keeps a split copy of its key bank and
    converts every memorised frame once."""
    _ensure_device(keys)
    k = keys.shape[0]
    rows = keys.reshape(k, -1, 128) if keys.dim() != 3 else keys
    rows, ko = _rows(_f32(rows), 128)
    if out is None:
        out = torch.empty(rows.shape, dtype=torch.float32, device=keys.device)
    orows = out.view(k, -1, 128) if out.dim() != 3 else out
    assert orows.shape == rows.shape and orows.stride(2) == 1 and orows.stride(1) == 128
    if rows.shape[1] == 0:
        return out
    oo = orows.stride(0) if k > 1 else rows.shape[1] * 128
    check(_lib.load().mivos_memory_split_keys(rows.data_ptr(), ko, orows.data_ptr(), oo, k, rows.shape[1], _stream()))
    return out


STREAMING_MAX_TOP_K = 64     # csrc/memory_read.hip: candidate lists of the streaming select kernels


def _memread_select(lib, keys, ko, keys_split, qk, k, n_mem, n_q, top_k, ws):
    """The affinity + streaming top-k launch in the engine's precision: "f16x3" streams pre-split keys (the caller's split
    bank, or a conversion of `keys` into scratch when there is none), "f32" the fp32 rows through the exact fp32 MFMA kernel."""
    if affinity_precision() == "f32":
        check(lib.mivos_memory_read_select(keys.data_ptr(), ko, qk.data_ptr(), k, n_mem, n_q, top_k, ws.data_ptr(), ws.numel(), _stream()))
        return
    if keys_split is None:
        sp, so = _workspace(k * n_mem * 512, keys.device, "memread_ksplit").data_ptr(), n_mem * 128
        check(lib.mivos_memory_split_keys(keys.data_ptr(), ko, sp, so, k, n_mem, _stream()))
    else:
        rows, so = _rows(_f32(keys_split), 128)
        assert rows.shape == keys.shape
        sp = rows.data_ptr()
    check(lib.mivos_memory_read_select_f16x3(sp, so, qk.data_ptr(), k, n_mem, n_q, top_k, ws.data_ptr(), ws.numel(), _stream()))


def _memread_args(keys, values, qk, top_k):
    keys, ko = _rows(_f32(keys), 128)
    k, n_mem, _ = keys.shape
    n_q = qk.shape[0]
    assert qk.is_contiguous() and qk.shape[1] == 128 and qk.dtype == torch.float32
    vo = 0
    if values is not None:
        values, vo = _rows(_f32(values), 512)
        assert values.shape[:2] == (k, n_mem)
    return keys, ko, values, vo, k, n_mem, n_q


def _memread_profile(ev, k, n_mem, n_q, top_k, out_rows):
    # affinity: 2*K*n_mem*n_q*128 FLOP, keys + queries read once; finalize: k value rows of 2 KB per (object, query) + the output
    PROFILE.append((90, 2.0 * k * n_mem * n_q * 128, ev[0], ev[1], (k, n_mem, n_q, top_k, 4.0 * 128 * (k * n_mem + n_q))))
    PROFILE.append((91, 2.0 * k * n_q * top_k * 512, ev[2], ev[3], (k, n_mem, n_q, top_k, 4.0 * 512 * k * n_q * (top_k + out_rows))))


def memory_read(keys, values, qk, top_k, out=None, keys_split=None):
    """keys [K, n_mem, 128], values [K, n_mem, 512], qk [n_q, 128] -> out [K, n_q, 512]
    (out may be a channel-slice view [K, n_q, 512] of a wider [K, n_q, C] buffer).  keys_split: split_keys(keys) kept by the
    caller (otherwise converted here on every call).  top_k=None: softmax over all memory positions (exact fp32)."""
    _ensure_device(keys)
    keys, ko, values, vo, k, n_mem, n_q = _memread_args(keys, values, qk, top_k)
    if out is None:
        out = torch.empty((k, n_q, 512), dtype=torch.float32, device=keys.device)
    assert out.shape == (k, n_q, 512) and out.stride(2) == 1
    lib = _lib.load()
    if top_k is None:            # full softmax over the bank (PropagationNetwork(top_k=None), prop_net.py:99-102)
        ws = _workspace(lib.mivos_memory_read_dense_workspace_bytes(k, n_mem, n_q), keys.device, "memread_dense")
        check(lib.mivos_memory_read_dense(keys.data_ptr(), ko, values.data_ptr(), vo, qk.data_ptr(), out.data_ptr(), out.stride(0), out.stride(1),
                                          None, None, 0, 0, 0, 1, k, n_mem, n_q, ws.data_ptr(), ws.numel(), _stream()))
        return out
    if top_k > STREAMING_MAX_TOP_K:     # beyond the streaming kernels' candidate lists: scores to scratch + radix select per query
        ws = _workspace(lib.mivos_memory_read_topk_any_workspace_bytes(k, n_mem, n_q), keys.device, "memread_any")
        check(lib.mivos_memory_read_topk_any(keys.data_ptr(), ko, values.data_ptr(), vo, qk.data_ptr(), out.data_ptr(), out.stride(0), out.stride(1),
                                             None, None, k, n_mem, n_q, top_k, ws.data_ptr(), ws.numel(), _stream()))
        return out
    ws = _workspace(lib.mivos_memory_read_workspace_bytes(k, n_mem, n_q, top_k), keys.device, "memread")
    ev = None
    if PROFILE is not None:      # bench.py: HIP events on the launch stream around each of the two launches
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
    _memread_select(lib, keys, ko, keys_split, qk, k, n_mem, n_q, top_k, ws)
    if ev:
        ev[1].record()
        ev[2].record()
    check(lib.mivos_memory_read_finalize(values.data_ptr(), vo, out.data_ptr(), out.stride(0), out.stride(1), k, n_mem, n_q, top_k,
                                         ws.data_ptr(), ws.numel(), _stream()))
    if ev:
        ev[3].record()
        _memread_profile(ev, k, n_mem, n_q, top_k, 1)
    return out


def memory_read_acts(keys, values, qk, top_k, h, w, tag="memread", keys_split=None):
    """memory_read whose readout lands pre-split in two SH32 Acts [K, h, w, 512]: (raw, relu(raw)) - what the decoder's first
    ResBlock reads.  keys [K, n_mem, 128], values [K, n_mem, 512], qk [h*w, 128]."""
    _ensure_device(keys)
    keys, ko, values, vo, k, n_mem, n_q = _memread_args(keys, values, qk, top_k)
    assert n_q == h * w
    raw, rel = alloc_act(k, h, w, 512, keys.device, (tag, "raw")), alloc_act(k, h, w, 512, keys.device, (tag, "relu"))
    lib = _lib.load()
    an, ar, ap = raw.strides()
    if top_k is None:
        ws = _workspace(lib.mivos_memory_read_dense_workspace_bytes(k, n_mem, n_q), keys.device, "memread_dense")
        check(lib.mivos_memory_read_dense(keys.data_ptr(), ko, values.data_ptr(), vo, qk.data_ptr(), None, 0, 0, raw.interior_ptr(), rel.interior_ptr(),
                                          an, ar, ap, w, k, n_mem, n_q, ws.data_ptr(), ws.numel(), _stream()))
        return raw, rel
    if top_k > STREAMING_MAX_TOP_K:
        dense = memory_read(keys, values, qk, top_k).view(k, h, w, 512)
        return to_act(dense, out=raw), to_act(dense, relu=True, out=rel)
    ws = _workspace(lib.mivos_memory_read_workspace_bytes(k, n_mem, n_q, top_k), keys.device, "memread")
    ev = None
    if PROFILE is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
    _memread_select(lib, keys, ko, keys_split, qk, k, n_mem, n_q, top_k, ws)
    if ev:
        ev[1].record()
        ev[2].record()
    check(lib.mivos_memory_read_finalize_sh32(values.data_ptr(), vo, raw.interior_ptr(), rel.interior_ptr(), an, ar, ap, w, k, n_mem, n_q, top_k,
                                              ws.data_ptr(), ws.numel(), _stream()))
    if ev:
        ev[3].record()
        _memread_profile(ev, k, n_mem, n_q, top_k, 2)
    return raw, rel


def memory_read_indices(keys, qk, top_k, keys_split=None):
    """Test/debug: the selected memory indices [K, n_q, k] (best first) and softmax weights."""
    _ensure_device(keys)
    qk = qk.contiguous()
    keys, ko, _, _, k, n_mem, n_q = _memread_args(keys, None, qk, top_k)
    idx = torch.empty((k, n_q, top_k), dtype=torch.int32, device=keys.device)
    wgt = torch.empty((k, n_q, top_k), dtype=torch.float32, device=keys.device)
    lib = _lib.load()
    if top_k > STREAMING_MAX_TOP_K:     # (survivors in list order, not ranked)
        ws = _workspace(lib.mivos_memory_read_topk_any_workspace_bytes(k, n_mem, n_q), keys.device, "memread_any")
        check(lib.mivos_memory_read_topk_any(keys.data_ptr(), ko, None, 0, qk.data_ptr(), None, 0, 0, idx.data_ptr(), wgt.data_ptr(), k, n_mem, n_q,
                                             top_k, ws.data_ptr(), ws.numel(), _stream()))
        return idx, wgt
    ws = _workspace(lib.mivos_memory_read_workspace_bytes(k, n_mem, n_q, top_k), keys.device, "memread")
    _memread_select(lib, keys, ko, keys_split, qk, k, n_mem, n_q, top_k, ws)
    check(lib.mivos_memory_read_finalize_indices(idx.data_ptr(), wgt.data_ptr(), k, n_mem, n_q, top_k, ws.data_ptr(), ws.numel(), _stream()))
    return idx, wgt


def attention_align(mk, qk, pos16, neg16):
    """mk [K, n_pos, 128], qk [n_pos, 128], pos16/neg16 [K, n_pos] -> [K, 2, n_pos]."""
    _ensure_device(mk)
    k, n_pos, _ = mk.shape
    mk, qk, pos16, neg16 = (_f32(t).contiguous() for t in (mk, qk, pos16, neg16))
    out = torch.empty((k, 2, n_pos), dtype=torch.float32, device=mk.device)
    check(_lib.load().mivos_attention_align(mk.data_ptr(), qk.data_ptr(), pos16.data_ptr(), neg16.data_ptr(), out.data_ptr(), k, n_pos, _stream()))
    return out


def attention_weights(mk, qk):
    """mk [B, n_mem, 128], qk [n_q, 128] (shared) or [B, n_q, 128] -> dense W [B, n_mem, n_q] = softmax over n_mem
    (AttentionMemory.forward; prop_net.py:115-129)."""
    _ensure_device(mk)
    mk, qk = _f32(mk).contiguous(), _f32(qk).contiguous()
    b, n_mem, _ = mk.shape
    n_q = qk.shape[-2]
    qs = 0 if qk.dim() == 2 else n_q * 128
    w = torch.empty((b, n_mem, n_q), dtype=torch.float32, device=mk.device)
    check(_lib.load().mivos_attention_weights(mk.data_ptr(), qk.data_ptr(), qs, w.data_ptr(), b, n_mem, n_q, _stream()))
    return w


def resize_bilinear_nhwc(x, H, W, out=None):
    """x [N,h,w,C] fp32 dense -> [N,H,W,C] (align_corners=False); `out` may be a channel-slice view of a wider NHWC buffer."""
    _ensure_device(x)
    n, h, w, c = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty((n, H, W, c), dtype=torch.float32, device=x.device)
    assert tuple(out.shape) == (n, H, W, c) and out.stride(3) == 1 and out.stride(1) == W * out.stride(2)
    check(_lib.load().mivos_resize_bilinear_nhwc(x.data_ptr(), out.data_ptr(), out.stride(0), out.stride(2), n, h, w, H, W, c, _stream()))
    return out


def global_avgpool(x):
    """x [N,H,W,C] fp32 dense -> [N,1,1,C] (nn.AdaptiveAvgPool2d(1))."""
    _ensure_device(x)
    n, h, w, c = x.shape
    assert x.is_contiguous()
    y = torch.empty((n, 1, 1, c), dtype=torch.float32, device=x.device)
    check(_lib.load().mivos_global_avgpool(x.data_ptr(), y.data_ptr(), n, h * w, c, _stream()))
    return y


def dilate3x3(x):
    """Binary float planes [P,H,W] -> 3x3 dilation (cv2.dilate with a 3x3 kernel of ones)."""
    _ensure_device(x)
    x = _f32(x).contiguous()
    p, h, w = x.shape
    y = torch.empty_like(x)
    check(_lib.load().mivos_dilate3x3(x.data_ptr(), y.data_ptr(), p, h, w, _stream()))
    return y


def area_pool16(x):
    """x [planes, H, W] -> [planes, H/16, W/16]."""
    _ensure_device(x)
    x = _f32(x).contiguous()
    p, h, w = x.shape
    y = torch.empty((p, h // 16, w // 16), dtype=torch.float32, device=x.device)
    check(_lib.load().mivos_area_pool16(x.data_ptr(), y.data_ptr(), p, h, w, _stream()))
    return y


def resize_bilinear(x, H, W, act=0, out=None):
    """x [planes, h, w] -> [planes, H, W], align_corners=False; act=1 applies a sigmoid."""
    _ensure_device(x)
    x = _f32(x).contiguous()
    p, h, w = x.shape
    if out is None:
        out = torch.empty((p, H, W), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.numel() == p * H * W
    check(_lib.load().mivos_resize_bilinear(x.data_ptr(), out.data_ptr(), p, h, w, H, W, act, _stream()))
    return out


def aggregate(prob, keep_bg=False, hard=False, soft_bg=True):
    """prob [K, ...] -> [K+1, ...] (keep_bg) or [K, ...]; model/aggregate.py semantics."""
    _ensure_device(prob)
    prob = _f32(prob).contiguous()
    k = prob.shape[0]
    p = prob[0].numel()
    out = torch.empty((k + 1 if keep_bg else k,) + tuple(prob.shape[1:]), dtype=torch.float32, device=prob.device)
    fn = _lib.load().mivos_aggregate_wbg if soft_bg else _lib.load().mivos_aggregate_sbg
    check(fn(prob.data_ptr(), out.data_ptr(), k, p, int(keep_bg), int(hard), _stream()))
    return out


def aggregate_channel(prob, keep_bg=False, hard=False):
    """prob [B, K, H, W] -> (logits [B, K+1, H, W], softmax [B, K(+1), H, W]); aggregate.py:39-53."""
    _ensure_device(prob)
    prob = _f32(prob).contiguous()
    b, k, h, w = prob.shape
    logits = torch.empty((b, k + 1, h, w), dtype=torch.float32, device=prob.device)
    soft = torch.empty((b, k + 1 if keep_bg else k, h, w), dtype=torch.float32, device=prob.device)
    check(_lib.load().mivos_aggregate_wbg_channel(prob.data_ptr(), logits.data_ptr(), soft.data_ptr(), b, k, h * w, int(keep_bg), int(hard), _stream()))
    return logits, soft


def argmax_u8(prob, out=None):
    """prob [C, ...] planes (plane stride = prob.stride(0)) -> uint8 [...]."""
    _ensure_device(prob)
    c = prob.shape[0]
    p = prob[0].numel()
    assert prob[0].is_contiguous()
    if out is None:
        out = torch.empty(tuple(prob.shape[1:]), dtype=torch.uint8, device=prob.device)
    assert out.is_contiguous() and out.numel() == p
    check(_lib.load().mivos_argmax_u8(_f32(prob).data_ptr(), prob.stride(0) if c > 1 else p, out.data_ptr(), c, p, _stream()))
    return out


def mask_diff(mask, prob):
    _ensure_device(mask)
    mask, prob = _f32(mask).contiguous(), _f32(prob).contiguous()
    pos, neg = torch.empty_like(mask), torch.empty_like(mask)
    check(_lib.load().mivos_mask_diff(mask.data_ptr(), prob.data_ptr(), pos.data_ptr(), neg.data_ptr(), mask.numel(), _stream()))
    return pos, neg


def sigmoid(x):
    _ensure_device(x)
    x = _f32(x).contiguous()
    y = torch.empty_like(x)
    check(_lib.load().mivos_sigmoid(x.data_ptr(), y.data_ptr(), x.numel(), _stream()))
    return y


def mask_others(masks):
    """masks [K, ...] -> others[i] = sum_{j != i} masks[j]."""
    _ensure_device(masks)
    masks = _f32(masks).contiguous()
    out = torch.empty_like(masks)
    check(_lib.load().mivos_mask_others(masks.data_ptr(), out.data_ptr(), masks.shape[0], masks[0].numel(), _stream()))
    return out


def _interleave_desc(planes, c_out):
    """mivos_interleave_desc of a plane list [(tensor_or_float, batch_stride), ...] (+ the tensors to keep alive)."""
    d = InterleaveDesc()
    d.C = c_out
    keep = []
    for c in range(16):
        d.plane[c], d.nstride[c], d.cval[c] = None, 0, 0.0
    for c, (src, ns) in enumerate(planes):
        if isinstance(src, (int, float)):
            d.cval[c] = float(src)
        else:
            _ensure_device(src)
            assert src.dtype == torch.float32
            keep.append(src)
            d.plane[c], d.nstride[c] = src.data_ptr(), ns
    return d, keep


def interleave(planes, n, p, c_out, device):
    """planes: list of (tensor_or_float, batch_stride) per channel (missing channels up to c_out are
    zero) -> dense NHWC [n, p, c_out].  A float entry is a constant plane."""
    d, keep = _interleave_desc(planes, c_out)
    out = torch.empty((n, p, c_out), dtype=torch.float32, device=device)
    check(_lib.load().mivos_interleave_planes(C.byref(d), out.data_ptr(), n, p, _stream()))
    return out
