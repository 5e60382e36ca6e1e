#!/bin/bash
# PMC passes on config E's 64^3 sub-mesh solve (bench_configs.py E): instruction mix of the n = 16 solver.
# bash profiles/pmc_config_E.sh [tag]   -> gpurun_out/prof_E_<tag>/summary.txt
set -u
TAG=${1:-r02}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_E_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench_configs.py E --reps 2 > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_solve' in r['Kernel_Name'] or 'k_ql16' in r['Kernel_Name'] or 'k_tw16' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn, cs in acc.items():
    print(kn)
    for k,v in sorted(cs.items()): print('   %-22s %.4g  (%d dispatches)'%(k, sum(v)/len(v), len(v)))
PY
cat $OUT/summary.txt
