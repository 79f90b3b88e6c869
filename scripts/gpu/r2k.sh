#!/bin/bash
export TMPDIR=/tmp
MIVOS_MEMREAD_DBG=1 timeout 200 python scripts/memread_microbench.py 2>&1 | grep -E "memread_select\]" | awk 'NR%6==0'
