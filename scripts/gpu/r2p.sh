#!/bin/bash
set +e
export TMPDIR=/tmp
for v in 0 96 128 200 256; do
  echo "== MIVOS_PP_SMALL_WGS=$v"; MIVOS_PP_SMALL_WGS=$v timeout 300 python bench.py --steps 274 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 2>/dev/null | cut -c1-150
done
echo "== baseline again"; timeout 300 python bench.py --steps 274 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 2>/dev/null | cut -c1-150
