#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash profiles/collect_wh.sh rNN
# rocprofv3 evidence for the widening rows SURVEY.md 8f-1 / 8f-4 on the current build: bench_configs.py legs
# W (Wannier90 silicon: 2972 hopping terms, 8 states) and H (hybrid Wannier centres of a 16-layer slab): one kernel-trace + stats
# run, then FETCH_SIZE, WRITE_SIZE and a VALU / LDS group each in its own run.  Summaries under gpurun_out/summary_<tag>f/.
set -u
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}f
SUM=$REPO/gpurun_out/summary_${TAG}f
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench_configs.py W H --reps 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $SUM/bench_configs_under_trace.jsonl 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 profiles/summarise.py $OUT $SUM > $SUM/summary.txt 2>&1
python3 $REPO/bench_configs.py W H --reps 5 > $SUM/bench_configs.jsonl 2> $SUM/bench_configs.err
ls -la $SUM
