"""Per-shape throughput of mivos_conv2d_fused on the shapes of the 480p / K=5 hot path (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mivos_amd import ops, _lib
from mivos_amd.ops import ConvLayer
torch.set_grad_enabled(False)
DEV = "cuda:0"
SHAPES = [  # name, N, H, W, Cin, Cout, k, stride
    ("dec.up_8_4 3x3 256->256 b5", 5, 120, 216, 256, 256, 3, 1),
    ("dec.up_16_8 3x3 512->256 b5", 5, 60, 108, 512, 256, 3, 1),
    ("dec.compress 3x3 1024->512 b5", 5, 30, 54, 1024, 512, 3, 1),
    ("dec.skip4 3x3 256->256 b1", 1, 120, 216, 256, 256, 3, 1),
    ("dec.skip8 3x3 512->512 b1", 1, 60, 108, 512, 512, 3, 1),
    ("kv 3x3 1024->640 b5", 5, 30, 54, 1024, 640, 3, 1),
    ("kv 3x3 1024->640 b1", 1, 30, 54, 1024, 640, 3, 1),
    ("enc.l1 1x1 64->256 b5", 5, 120, 216, 64, 256, 1, 1),
    ("enc.l1 3x3 64->64 b5", 5, 120, 216, 64, 64, 3, 1),
    ("enc.l1 1x1 256->64 b5", 5, 120, 216, 256, 64, 1, 1),
    ("enc.l2 3x3 128->128 b5", 5, 60, 108, 128, 128, 3, 1),
    ("enc.l2 1x1 128->512 b5", 5, 60, 108, 128, 512, 1, 1),
    ("enc.l3 3x3 256->256 b5", 5, 30, 54, 256, 256, 3, 1),
    ("enc.l3 1x1 256->1024 b5", 5, 30, 54, 256, 1024, 1, 1),
    ("enc.l3 1x1 1024->256 b5", 5, 30, 54, 1024, 256, 1, 1),
    ("enc.l3 3x3 256->256 b1", 1, 30, 54, 256, 256, 3, 1),
    ("enc.l3 1x1 1024->256 b1", 1, 30, 54, 1024, 256, 1, 1),
    ("stem 7x7 8->64 s2 b5", 5, 480, 864, 8, 64, 7, 2),
    ("fusion 3x3 32->32 b5", 5, 480, 864, 32, 32, 3, 1),
    ("fusion 3x3 16->32 b5", 5, 480, 864, 16, 32, 3, 1),
    ("pred 3x3 256->1 b5", 5, 120, 216, 256, 1, 3, 1),
    ("fusion head 3x3 32->1 b5", 5, 480, 864, 32, 1, 3, 1),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
ops.CONV_PRECISION = os.environ.get("PREC", ops.CONV_PRECISION)
print("precision", ops.CONV_PRECISION)
reps = int(os.environ.get("REPS", "10"))
tot_t = tot_f = 0.0
for name, n, h, w, cin, cout, k, s in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(n, h, w, cin, device=DEV)
    L = ConvLayer.pack(torch.randn(cout, cin, k, k) * 0.05, torch.randn(cout) * 0.1, None, s, k // 2).to(DEV)
    y = ops.conv(x, L)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv(x, L, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = y.shape[0] * y.shape[1] * y.shape[2]
    fl = 2.0 * m * cout * k * k * cin
    var = _lib.load().mivos_conv2d_variant(m, cout)
    print(f"{name:34s} M={m:7d} variant {var}  {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TF/s")
