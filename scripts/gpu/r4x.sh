#!/bin/bash
# MFMA utilisation (rocprofv3 derived counter MfmaUtil) of one config-3 session and of one deep-bank memory read
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
cd /tmp
rm -rf /tmp/mu3 /tmp/mu5
timeout 70 rocprofv3 --pmc MfmaUtil -d /tmp/mu3 --output-format csv -- python $R/bench.py --steps 137 --warmup 8 --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 > /dev/null 2> /tmp/mu3.err
echo "config3 rc $?"; tail -2 /tmp/mu3.err | cut -c1-200
python $R/scripts/pmc_mfma_util.py $(find /tmp/mu3 -name "*counter_collection.csv" | head -1) $R/gpurun_out/r4x_config3_mfma_util.json | head -14
timeout 50 rocprofv3 --pmc MfmaUtil -d /tmp/mu5 --output-format csv -- python $R/scripts/memread_case.py 3 100 8160 50 q256 > /dev/null 2> /tmp/mu5.err
echo "memread rc $?"
python $R/scripts/pmc_mfma_util.py $(find /tmp/mu5 -name "*counter_collection.csv" | head -1) $R/gpurun_out/r4x_config5_memread256_T100_mfma_util.json | head -4
