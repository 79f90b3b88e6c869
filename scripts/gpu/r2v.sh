#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== default BR"; timeout 200 python scripts/memread_check.py 23 30 54 1 50 2>&1 | tail -20
echo "== BR off"; MIVOS_MEMREAD_BR_MIN=100000000 timeout 200 python scripts/memread_check.py 23 30 54 1 50 2>&1 | tail -20
echo "== BR on small"; MIVOS_MEMREAD_BR_MIN=1 timeout 200 python scripts/memread_check.py 7 30 54 5 50 2>&1 | tail -20
