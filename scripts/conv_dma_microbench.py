"""LDS-DMA fed f16x3 conv (precision 2, SH32 input) vs the register-staged kernel (precision 1): same shapes,
bit-level comparison of the outputs and throughput.  Run on the GPU box."""
import sys, os, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mivos_amd import ops, _lib
from mivos_amd._lib import ConvDesc, check
from mivos_amd.ops import ConvLayer
torch.set_grad_enabled(False)
DEV = "cuda:0"
SHAPES = [  # name, N, H, W, Cin, Cout, k, stride
    ("dec.up_8_4 3x3 256->256 b5", 5, 120, 216, 256, 256, 3, 1),
    ("l2res 3x3 256->256 b80 30x54", 80, 30, 54, 256, 256, 3, 1),
    ("dec.up_8_4 3x3 256->256 b1", 1, 120, 216, 256, 256, 3, 1),
    ("dec.up_8_4 3x3 256->256 b2", 2, 120, 216, 256, 256, 3, 1),
    ("dec.up_16_8 3x3 512->256 b5", 5, 60, 108, 512, 256, 3, 1),
    ("dec.skip4 3x3 256->256 b8", 8, 120, 216, 256, 256, 3, 1),
    ("enc.l1 1x1 64->256 b5", 5, 120, 216, 64, 256, 1, 1),
    ("enc.l1 3x3 64->64 b5", 5, 120, 216, 64, 64, 3, 1),
    ("enc.l1 1x1 256->64 b5", 5, 120, 216, 256, 64, 1, 1),
    ("enc.l2 3x3 128->128 b5", 5, 60, 108, 128, 128, 3, 1),
    ("enc.l2 1x1 128->512 b5", 5, 60, 108, 128, 512, 1, 1),
    ("enc.l2 3x3 128->128 s2 b5", 5, 120, 216, 128, 128, 3, 2),
    ("enc.l3 3x3 256->256 b5", 5, 30, 54, 256, 256, 3, 1),
    ("enc.l3 1x1 256->1024 b5", 5, 30, 54, 256, 1024, 1, 1),
    ("enc.l3 1x1 1024->256 b5", 5, 30, 54, 1024, 256, 1, 1),
    ("ragged 3x3 64->96", 2, 37, 53, 64, 96, 3, 1),
    ("fusion 3x3 32->32 b5", 5, 480, 864, 32, 32, 3, 1),
    ("kv 3x3 1024->640 b5", 5, 30, 54, 1024, 640, 3, 1),
    ("kv 3x3 1024->640 b1", 1, 30, 54, 1024, 640, 3, 1),
    ("enc.l3 3x3 256->256 b1", 1, 30, 54, 256, 256, 3, 1),
    ("enc.l3 1x1 1024->256 b1", 1, 30, 54, 1024, 256, 1, 1),
    ("enc.l3 1x1 256->1024 b1", 1, 30, 54, 256, 1024, 1, 1),
    ("dec.compress 3x3 1024->512 b5", 5, 30, 54, 1024, 512, 3, 1),
    ("dec.skip8 3x3 512->512 b8", 8, 60, 108, 512, 512, 3, 1),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
reps = int(os.environ.get("REPS", "10"))
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, n, h, w, cin, cout, k, s in SHAPES:
    if only and only not in name:
        continue
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin, device=DEV)
    L = ConvLayer.pack(torch.randn(cout, cin, k, k) * 0.05, torch.randn(cout) * 0.1, None, s, k // 2).to(DEV)
    res = torch.randn(n, (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1, cout, device=DEV)
    y1 = ops.conv(x, L, relu_out=True, res=res)
    t1 = timeit(lambda: ops.conv(x, L, relu_out=True, res=res, out=y1))
    # precision 2 operands: SH32 activations inside a one-pixel zero border
    pad = k // 2
    xs = torch.zeros(n, h + 2, w + 2, cin, device=DEV)
    xin = xs[:, 1:h + 1, 1:w + 1]
    rs, ns = (w + 2) * cin, (h + 2) * (w + 2) * cin
    check(lib.mivos_pack_activation_sh32(x.data_ptr(), h * w * cin, w * cin, cin, xin.data_ptr(), ns, rs, cin, n, h, w, cin, 0, st()))
    wmax = float(L.w.abs().max())
    sh = 14 - math.floor(math.log2(wmax))
    wd = torch.empty(lib.mivos_pack_weights_f16x3_dma_bytes(cout, k, k, cin), dtype=torch.uint8, device=DEV)
    check(lib.mivos_pack_weights_f16x3_dma(L.w.data_ptr(), wd.data_ptr(), cout, k, k, cin, 2.0 ** sh, st()))
    _, scale16 = L.f16x3()
    y2 = torch.full_like(y1, float("nan"))
    d = ConvDesc()
    d.x, d.w, d.scale, d.bias, d.res, d.y = xin.data_ptr(), wd.data_ptr(), scale16.data_ptr(), L.bias.data_ptr(), res.data_ptr(), y2.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = n, h, w, cin, cout, k, k
    d.stride, d.pad, d.Ho, d.Wo, d.split = s, pad, y1.shape[1], y1.shape[2], cout
    d.relu_in, d.relu_out, d.precision = 0, 1, 2
    d.x_nstride, d.x_rstride, d.x_pstride, d.x_border, d.x_format = ns, rs, cin, 1, 1
    d.y_nstride, d.y_pstride = y1.shape[1] * y1.shape[2] * cout, cout
    d.res_nstride, d.res_pstride = d.y_nstride, cout
    sh_io = bool(os.environ.get("SH"))          # SH=1: SH32 output and SH32 residual as in the engine
    if sh_io:
        res_a, out_a = ops.to_act(res), ops.alloc_act(*y1.shape, x.device)
        d.res, d.res_format, d.y, d.y_format = res_a.interior_ptr(), 1, out_a.interior_ptr(), 1
        d.res_nstride, d.res_rstride, d.res_pstride = res_a.strides()
        d.y_nstride, d.y_rstride, d.y_pstride = out_a.strides()
    ws = ops._workspace(ops.SPLITK_WORKSPACE_BYTES, x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() - ops.STATUS_BYTES      # (the tail is the status block ops.conv reserves)
    run = lambda: check(lib.mivos_conv2d_fused(C.byref(d), st()))
    t2 = timeit(run)
    m = y1.shape[0] * y1.shape[1] * y1.shape[2]
    fl = 2.0 * m * cout * k * k * cin
    if sh_io:
        y2 = ops.to_f32(out_a)
    if os.environ.get("MIVOS_ABL") == "8":
        c = ws.view(torch.int64)[:3].cpu().tolist()
        print(f"   workgroup 0: K loop {c[0]} shader cycles for {c[1]} steps = {c[0] / max(c[1], 1):.0f} cycles/step; loop + epilogue {c[2]} cycles; kernel wall {t2 * 1e3:.1f} us")
    diff = float((y1 - y2).abs().max())
    print(f"{name:30s} M={m:7d} var {lib.mivos_conv2d_variant_f16x3(m, cout)}/{lib.mivos_conv2d_variant_pp(m, cout, k * k * cin // 32)}  reg {t1*1e3:8.1f} us {fl/t1/1e9:6.1f} TF/s | "
          f"dma {t2*1e3:8.1f} us {fl/t2/1e9:6.1f} TF/s  x{t1/t2:4.2f}  max|diff| {diff:.3g} (|y| max {float(y1.abs().max()):.3g})")
