#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value scripts/ubench/mfma_f16_tile.hip -o /tmp/mfma_f16_tile && timeout 120 /tmp/mfma_f16_tile > $O/r2x_mfma_f16_tile.txt 2>&1
cat $O/r2x_mfma_f16_tile.txt
