"""Select-kernel timing over bank depths: 64 queries per workgroup (memread_select_kernel), 128 (memread_select32_kernel) and 256
(memread_select256_kernel, candidate regions in global scratch), HIP events through the staged C ABI; index sets of the 128- and 256-query
kernels compared.  `--wide` adds the shallow banks (where does the 256-query kernel start to pay?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mivos_amd import _lib, ops  # noqa: E402
from mivos_amd._lib import check  # noqa: E402

torch.set_grad_enabled(False)
lib = _lib.load()
st = ops._stream()
CASES = ((3, 50, 8160, 50), (3, 100, 8160, 50), (3, 200, 8160, 50))
if "--wide" in sys.argv:
    CASES = ((5, 12, 1620, 50), (5, 40, 1620, 50), (5, 100, 1620, 50), (3, 5, 8160, 50), (3, 10, 8160, 50), (3, 20, 8160, 50), (3, 30, 8160, 50)) + CASES
for K, T, hw, topk in CASES:
    n_mem = T * hw
    keys = torch.randn(K, n_mem, 128, device="cuda") * 3
    q = torch.randn(hw, 128, device="cuda") * 3
    ks = ops.split_keys(keys)
    ws = ops._workspace(lib.mivos_memory_read_workspace_bytes(K, n_mem, hw, topk), keys.device, "memread")
    res = {}
    modes = (("q64", 1 << 40, 1 << 60, 0), ("q128", 0, 1 << 60, 0), ("q256", 1 << 40, 0, 0), ("q256hf", 1 << 40, 0, 1))
    if "--hf" in sys.argv:
        modes = modes[2:]
    for mode, q128, q256, hf in modes:
        lib.mivos_memory_read_set_q128_min(q128)
        lib.mivos_memory_read_set_q256_min(q256)
        lib.mivos_memory_read_set_hifirst(hf)
        check(lib.mivos_memory_read_select_f16x3(ks.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            check(lib.mivos_memory_read_select_f16x3(ks.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        idx, _ = ops.memory_read_indices(keys, q, topk, keys_split=ks)
        res[mode] = (ms, torch.sort(idx.long(), 2)[0])
    lib.mivos_memory_read_set_q128_min(400000)
    lib.mivos_memory_read_set_q256_min(200000)
    lib.mivos_memory_read_set_hifirst(0)
    fl = 2.0 * K * n_mem * hw * 128
    if "--hf" in sys.argv:
        same = float((res["q256"][1] == res["q256hf"][1]).all(2).float().mean())
        print(f"{'1080p' if hw == 8160 else '480p'} K={K} T={T}: q256 {res['q256'][0]:8.3f} ms ({fl / res['q256'][0] / 1e9:6.1f} TF/s)   hi-first {res['q256hf'][0]:8.3f} ms "
              f"({fl / res['q256hf'][0] / 1e9:6.1f} TF/s)   x{res['q256'][0] / res['q256hf'][0]:4.2f}   index sets equal on {same:.6f} of the queries", flush=True)
        continue
    same = float((res["q128"][1] == res["q256"][1]).all(2).float().mean())
    print(f"{'1080p' if hw == 8160 else '480p'} K={K} T={T}: q64 {res['q64'][0]:8.3f} ms   q128 {res['q128'][0]:8.3f} ms ({fl / res['q128'][0] / 1e9:6.1f} TF/s)   q256 {res['q256'][0]:8.3f} ms ({fl / res['q256'][0] / 1e9:6.1f} TF/s)   "
          f"hi-first {res['q256hf'][0]:8.3f} ms   x{res['q128'][0] / res['q256'][0]:4.2f}   index sets equal on {same:.6f} of the queries", flush=True)
