#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash profiles/collect.sh rNN
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and summaries under gpurun_out/summary_<tag>/
# (copy the summaries into profiles/ afterwards; gpurun_out/ is scratch).
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
SUM=$REPO/gpurun_out/summary_$TAG
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
EXTRA=${2:-}    # e.g. --two-calls: the step as two launches (k_grid_rows + k_flux_rows) instead of the fused pass
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --preheat-s 0 --no-cpu-baseline --headline-only $EXTRA"   # (the other legs replay thousands of launches: minutes under the profiler)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $SUM/bench_under_trace.json 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 profiles/summarise.py $OUT $SUM
ls -la $SUM
