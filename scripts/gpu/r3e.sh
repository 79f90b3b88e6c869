#!/bin/bash
set +e
bash scripts/profile_bench.sh r02i --config 3 2>&1 | tail -20
