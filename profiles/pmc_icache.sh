#!/bin/bash
# Run on the GPU box from the repo root: instruction-cache counters of the register-resident eigen-solvers
# (k_solve_row16: 106-144 KB of code, k_solve_reg<8>: 71-140 KB, against a 64 KB instruction cache).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/row16 -- python3 $REPO/profiles/row16_ab.py > $OUT/row16.txt 2> $OUT/row16.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_IFETCH --output-format csv -d $OUT/row16b -- python3 $REPO/profiles/row16_ab.py > /dev/null 2> $OUT/row16b.err
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for tag in ("row16", "row16b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob("gpurun_out/pmc_icache/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "solve" in k:
            print(tag, k, {c: "%.3g" % v for c, v in d.items()})
PY
