#!/bin/bash
# Run on the GPU box from the repo root: instruction-cache and issue-stall counters of config E's n = 16 kernels
# (one chunk in flight, so that each kernel's counters are its own).  bash profiles/pmc_icache_E.sh [tag]
set -u
TAG=${1:-r04}
REPO=$(pwd); OUT=$REPO/gpurun_out/icache_E_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TBK_TW16_STREAMS=1
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_IFETCH SQ_WAIT_ANY SQ_IFETCH_LEVEL" \
            "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench_configs.py E --reps 2 > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in ('k_solve', 'k_ql16', 'k_tw16', 'k_e16', 'k_chain')):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn, cs in acc.items():
    print(kn)
    for k,v in sorted(cs.items()): print('   %-28s %.4g  (%d dispatches)'%(k, sum(v)/len(v), len(v)))
PY
cat $OUT/summary.txt
