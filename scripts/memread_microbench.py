"""Throughput of the memory read on hot-path sizes: select (affinity MFMA + streaming top-k; both the exact fp32 MFMA kernel
and the error-compensated fp16 one on pre-split keys) and finalize (merge + softmax + value gather) timed separately with HIP
events through the staged C ABI.  `MIVOS_ABL=1` (environment) runs the
MFMA + staging skeleton of the select kernel without selection (ablation); `--check` compares the index sets with a
torch top-k of the fp64 affinity on the smallest case."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mivos_amd import _lib, ops  # noqa: E402
from mivos_amd._lib import check  # noqa: E402

torch.set_grad_enabled(False)
DEV = "cuda:0"
CASES = [("480p K=1 T=5 top20", 1, 5, 1620, 20, 3.0), ("480p K=1 T=12 top20", 1, 12, 1620, 20, 3.0),
         ("480p K=5 T=5 top50", 5, 5, 1620, 50, 3.0), ("480p K=5 T=7 top50", 5, 7, 1620, 50, 3.0),
         ("480p K=5 T=12 top50", 5, 12, 1620, 50, 3.0), ("480p K=5 T=12 top50 flat", 5, 12, 1620, 50, 1.0),
         ("480p K=3 T=12 top50", 3, 12, 1620, 50, 3.0),
         ("1080p K=3 T=20 top50", 3, 20, 8160, 50, 3.0), ("1080p K=3 T=100 top50", 3, 100, 8160, 50, 3.0)]
torch.manual_seed(0)
reps = 5
lib = _lib.load()
for name, K, T, hw, topk, scale in CASES:
    g = torch.Generator().manual_seed(0)
    n_mem = T * hw
    keys = torch.randn(K, n_mem, 128, device=DEV) * scale
    vals = torch.randn(K, n_mem, 512, device=DEV)
    q = torch.randn(hw, 128, device=DEV) * scale
    out = ops.memory_read(keys, vals, q, topk)
    torch.cuda.synchronize()
    ws = ops._workspace(lib.mivos_memory_read_workspace_bytes(K, n_mem, hw, topk), keys.device, "memread")
    st = ops._stream()
    ksplit = ops.split_keys(keys)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record()
    for _ in range(reps):
        check(lib.mivos_memory_read_select(keys.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
    ev[1].record()
    old_min = lib.mivos_memory_read_set_q128_min(1 << 40)          # 16 queries per wave
    for _ in range(reps):
        check(lib.mivos_memory_read_select_f16x3(ksplit.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
    ev[2].record()
    lib.mivos_memory_read_set_q128_min(0)                          # 32 queries per wave
    for _ in range(reps):
        check(lib.mivos_memory_read_select_f16x3(ksplit.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
    ev[3].record()
    lib.mivos_memory_read_set_q128_min(old_min)
    for _ in range(reps):
        check(lib.mivos_memory_read_finalize(vals.data_ptr(), n_mem * 512, out.data_ptr(), out.stride(0), out.stride(1), K, n_mem, hw, topk,
                                             ws.data_ptr(), ws.numel(), st))
    ev[4].record()
    torch.cuda.synchronize()
    sel, sel16, sel32, fin = (ev[i].elapsed_time(ev[i + 1]) / reps for i in range(4))
    fl = 2.0 * K * n_mem * hw * 128
    gather = 4.0 * 512 * K * hw * (topk + 1)
    print(f"{name:28s} select f32 {sel * 1e3:9.1f} us {fl / sel / 1e9:6.1f} TF/s ({fl / sel / 1e9 / 157.3 * 100:4.1f}% of f32 MFMA peak)   "
          f"f16x3 q64 {sel16 * 1e3:9.1f} us {fl / sel16 / 1e9:6.1f} TF/s ({3 * fl / sel16 / 1e9 / 2500 * 100:4.1f}% of 3-product fp16 peak, x{sel / sel16:4.2f})   "
          f"q128 {sel32 * 1e3:9.1f} us {fl / sel32 / 1e9:6.1f} TF/s ({3 * fl / sel32 / 1e9 / 2500 * 100:4.1f}%, x{sel / sel32:4.2f})   "
          f"finalize {fin * 1e3:7.1f} us ({gather / fin / 1e6:6.0f} GB/s)", flush=True)

if "--check" in sys.argv:
    K, T, hw, topk = 2, 3, 300, 50
    keys = torch.randn(K, T * hw, 128, device=DEV) * 3
    q = torch.randn(hw, 128, device=DEV) * 3
    aff = torch.einsum("kmc,qc->kmq", keys.double(), q.double() / (128 ** 0.5))
    ref = torch.topk(aff, topk, dim=1)[1].permute(0, 2, 1)
    for mode in ("f32", "f16x3", "f16x3-q128"):
        ops.CONV_PRECISION = mode.split("-")[0]
        lib.mivos_memory_read_set_q128_min(0 if mode.endswith("q128") else 1 << 40)
        idx, wgt = ops.memory_read_indices(keys, q, topk)
        same = (torch.sort(idx.long(), 2)[0] == torch.sort(ref, 2)[0]).all(2).float().mean()
        print(mode, "index sets equal to fp64 top-k:", float(same), " weights sum:", float(wgt.sum(2).mean()))
