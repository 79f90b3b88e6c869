# PMC passes over the precision-2 conv microbench (one shape); usage: pp_pmc.sh "<counter set>" ["<counter set>" ...]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  REPS=1 timeout 200 rocprofv3 --pmc $set -d /tmp/pm$i --output-format csv -- python $R/scripts/conv_dma_microbench.py "up_8_4 3x3 256->256 b5" > /dev/null 2>&1
  f=$(find /tmp/pm$i -name "*counter_collection.csv" | head -1)
  python $R/scripts/pmc_summary.py $f conv_f16x3_pp >> $R/gpurun_out/pp_pmc.txt 2>&1
done
