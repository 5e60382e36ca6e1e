#!/bin/bash
# Run on the GPU box from the repo root: PMC passes on the microbench of the fused n = 16 kernel (profiles/microbench/e16_bench):
# bash profiles/pmc_e16_bench.sh <binary> [args...]  ->  prints per-kernel counter means
set -u
BIN=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_e16_bench; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $REPO/$BIN "$@" > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_e16' in r['Kernel_Name'] or 'k_tw16' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn, cs in acc.items():
    print(kn)
    for k,v in sorted(cs.items()): print('   %-28s %.4g  (%d dispatches)'%(k, sum(v)/len(v), len(v)))
PY
