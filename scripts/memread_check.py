"""Exact top-k membership of the fp16 select kernels against a torch fp64 top-k on one shape (T h w K top_k), with the
missing / extra positions of the first failing queries (tile, row) - the tool that located the MFMA read hazard.
   python scripts/memread_check.py 23 30 54 1 50        (MIVOS_MEMREAD_BR_MIN=1 forces the branchy append)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mivos_amd import _lib, ops
torch.set_grad_enabled(False)
lib = _lib.load()
T, h, w, K, topk = [int(x) for x in sys.argv[1:6]]
g = torch.Generator().manual_seed(T * 100 + K)
mk = torch.randn(K, 128, T, h, w, generator=g)
mv = torch.randn(K, 512, T, h, w, generator=g)
qk = torch.randn(1, 128, h, w, generator=g)
keys = mk.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, 128).contiguous().cuda()
q = qk.permute(0, 2, 3, 1).reshape(h * w, 128).contiguous().cuda()
aff = torch.einsum("kmc,qc->kmq", keys.double(), q.double() / (128 ** 0.5))
vals, ref = torch.topk(aff, topk + 1, dim=1)
clear = (vals[:, topk - 1] - vals[:, topk]) > 1e-5
ref = ref[:, :topk].permute(0, 2, 1)
for mode in ("f16x3", "f16x3-q128"):
    ops.CONV_PRECISION = "f16x3"
    lib.mivos_memory_read_set_q128_min(0 if mode.endswith("q128") else 1 << 40)
    idx, wgt = ops.memory_read_indices(keys, q, topk)
    same = (torch.sort(idx.long(), 2)[0] == torch.sort(ref, 2)[0]).all(2)
    bad = (~same & clear).nonzero()
    print(mode, "bad queries:", bad.shape[0], "of", same.numel())
    plan = (__import__("ctypes").c_int32 * 7)()
    lib.mivos_memory_read_plan(K, T * h * w, h * w, topk, 1, plan)
    print(" plan", list(plan))
    for o, qq in bad[:12].tolist():
        got, want = set(idx[o, qq].tolist()), set(ref[o, qq].tolist())
        miss, extra = sorted(want - got), sorted(got - want)
        print("  q", qq, "wave", (qq % 128) // 32, "j", qq % 32, "missing", miss, "tile", [m // 32 for m in miss], "row", [m % 32 for m in miss], "extra", extra[:4],
              "dups", topk - len(got))
