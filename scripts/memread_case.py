"""One memory-read shape, the select kernels launched a few times each (profiler target): K T hw topk"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mivos_amd import _lib, ops
from mivos_amd._lib import check
torch.set_grad_enabled(False)
K, T, hw, topk = [int(x) for x in sys.argv[1:5]]
lib = _lib.load()
n_mem = T * hw
keys = torch.randn(K, n_mem, 128, device="cuda") * 3
q = torch.randn(hw, 128, device="cuda") * 3
ks = ops.split_keys(keys)
ws = ops._workspace(lib.mivos_memory_read_workspace_bytes(K, n_mem, hw, topk), keys.device, "memread")
st = ops._stream()
for mode in sys.argv[5:]:
    for _ in range(3):
        if mode == "f32":
            check(lib.mivos_memory_read_select(keys.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
        else:
            lib.mivos_memory_read_set_q128_min(0 if mode == "q128" else 1 << 40)
            lib.mivos_memory_read_set_q256_min(0 if mode == "q256" else 1 << 60)
            check(lib.mivos_memory_read_select_f16x3(ks.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, topk, ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
