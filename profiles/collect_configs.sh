#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash profiles/collect_configs.sh rNN "D E"
# rocprofv3 evidence for the single-GPU legs of BASELINE.json configs[3] (Kane-Mele 4096x512: k_grid_rows<4,1>,
# k_chain_partial, k_chain_final_wave) and configs[4] (cubic16: k_solve_wave<1,true,64>, k_link_det_big, ...):
# one kernel-trace + stats run, then each PMC group in its own run (the pool refuses mixed runs).
# Raw output under gpurun_out/prof_<tag>cfg/, summaries under gpurun_out/summary_<tag>cfg/ (copy to profiles/).
set -u
TAG=${1:-r02}
WHICH=${2:-"D E"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}cfg
SUM=$REPO/gpurun_out/summary_${TAG}cfg
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
export TBK_TW16_STREAMS=${TBK_TW16_STREAMS:-1}   # per-kernel durations and counters: one chunk of the n = 9..16 path in flight at a time
CMD="python3 $REPO/bench_configs.py $WHICH --reps 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $SUM/bench_configs_under_trace.jsonl 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 profiles/summarise.py $OUT $SUM > $SUM/summary.txt 2>&1
unset TBK_TW16_STREAMS
python3 $REPO/bench_configs.py $WHICH --reps 5 > $SUM/bench_configs.jsonl 2> $SUM/bench_configs.err   # (production setting: three chunks in flight)
ls -la $SUM
