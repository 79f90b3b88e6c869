"""Throughput of the fused memory read (affinity MFMA + streaming top-k + readout) on hot-path sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mivos_amd import ops
torch.set_grad_enabled(False)
DEV = "cuda:0"
for name, K, T, hw, topk, scale in [("480p K=1 T=5 top20", 1, 5, 1620, 20, 3.0), ("480p K=5 T=5 top50", 5, 5, 1620, 50, 3.0),
                                    ("480p K=5 T=12 top50", 5, 12, 1620, 50, 3.0), ("480p K=5 T=12 top50 flat", 5, 12, 1620, 50, 1.0),
                                    ("1080p K=3 T=20 top50", 3, 20, 8160, 50, 3.0)]:
    g = torch.Generator().manual_seed(0)
    keys = (torch.randn(K, T * hw, 128, generator=g) * scale).to(DEV)
    vals = torch.randn(K, T * hw, 512, generator=g).to(DEV)
    q = (torch.randn(hw, 128, generator=g) * scale).to(DEV)
    out = ops.memory_read(keys, vals, q, topk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.memory_read(keys, vals, q, topk, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * K * T * hw * hw * 128
    print(f"{name:28s} {ms*1e3:9.1f} us   affinity {fl/ms/1e9:6.1f} TF/s ({fl/ms/1e9/157.3*100:4.1f}% of f32 MFMA peak)")
