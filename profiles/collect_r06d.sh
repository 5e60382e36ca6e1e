#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect_r06d.sh [TAG]
# The 17..32-state kernels of round 6's second half (k_hh32<..,2,..>, k_ql32_lanes, k_tw32_vectors): kernel trace + counters of
# profiles/n17_probe.py (33^3 points, 16 / 17 / 24 / 32 states, with eigenvectors), then the sweeps that quote them.
set -u
TAG=${1:-r06d}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
SUM=$REPO/gpurun_out/summary_$TAG
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/profiles/n17_probe.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $SUM/n17_probe_under_trace.txt 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 profiles/summarise.py $OUT $SUM > $SUM/summary.txt 2>&1
python3 profiles/n17_probe.py > $SUM/n17_probe.txt 2>&1
python3 profiles/tw32_sweep.py > $SUM/tw32_sweep.txt 2>&1
python3 profiles/qlw_streams_probe.py > $SUM/qlw_streams_probe.txt 2>&1
python3 profiles/position_cliff_sweep.py > $SUM/position_cliff_sweep.txt 2>&1
python3 profiles/many_R_sweep.py > $SUM/many_R_sweep.txt 2>&1
python3 profiles/hh32_vec_probe.py > $SUM/eigenvalues_only_probe.txt 2>&1
TBK_QLW_MIN=0 STRESS_SIZES=17,20,24,25,29,32 python3 profiles/evecs_stress.py > $SUM/evecs_stress_17_32.txt 2>&1
python3 bench_configs.py R > $SUM/bench_configs_R.jsonl 2> $SUM/bench_configs_R.err
ls -la $SUM
