#!/bin/bash
# ablations of the select kernel: cycles per tile on one 1080p and one 480p case (results of ABL builds are garbage by design)
export TMPDIR=/tmp
for abl in 0 1 2 3 4; do
  echo "== ABL=$abl"
  MIVOS_ABL=$abl MIVOS_MEMREAD_DBG=1 timeout 120 python - <<'PY' 2>&1 | grep "memread_select\]" | awk 'NR%3==0'
import sys, os
sys.path.insert(0, ".")
import torch
from mivos_amd import _lib, ops
from mivos_amd._lib import check
lib = _lib.load()
torch.manual_seed(0)
for K, T, hw in [(5, 12, 1620), (3, 20, 8160)]:
    n_mem = T * hw
    keys = torch.randn(K, n_mem, 128, device="cuda") * 3
    q = torch.randn(hw, 128, device="cuda") * 3
    ws = ops._workspace(lib.mivos_memory_read_workspace_bytes(K, n_mem, hw, 50), keys.device, "memread")
    for _ in range(3):
        check(lib.mivos_memory_read_select(keys.data_ptr(), n_mem * 128, q.data_ptr(), K, n_mem, hw, 50, ws.data_ptr(), ws.numel(), ops._stream()))
    torch.cuda.synchronize()
PY
done
