#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/pmc_wilson_lanes.sh TAG
# kernel trace + counters of the Wilson-loop kernels of 3 / 4 bands (profiles/wilson_lanes_probe.py, default routes)
set -u
TAG=${1:-r06w}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
SUM=$REPO/gpurun_out/summary_$TAG
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/profiles/wilson_lanes_probe.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $SUM/probe_under_trace.txt 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 profiles/summarise.py $OUT $SUM > $SUM/summary.txt 2>&1
ls -la $SUM
