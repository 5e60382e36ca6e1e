#!/bin/bash
# Run on the GPU box from the repo root: dynamic vector-ALU instruction counts of k_e16<1> per phase, by ablation
# (profiles/microbench/e16_bench_sk<mask>, -DE16_SKIP=<mask>: the results of a masked build are wrong on purpose).
#   bash profiles/pmc_e16_phases.sh [side = 65] [masks ...]   ->  per build: wave-instructions per wavefront, and the kernel's time
set -u
SIDE=${1:-65}; shift
MASKS=${@:-0 1 4 8 16 32 64}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_e16_phases; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in $MASKS; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv \
      -d $OUT/sk$m -- $REPO/profiles/microbench/e16_bench_sk$m mesh $SIDE 3 > $OUT/sk$m.out 2> $OUT/sk$m.err
done
cd $REPO
python3 - $OUT $MASKS <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for m in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(out + '/sk%s/**/*counter_collection.csv' % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('void k_e16') or r['Kernel_Name'].startswith('k_e16'):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    dur = []
    for f in glob.glob(out + '/sk%s/**/*kernel_trace.csv' % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_e16' in r['Kernel_Name']:
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    mean = lambda v: sum(v) / len(v) if v else float('nan')
    w = mean(acc['SQ_WAVES'])
    print('skip %3s: VALU %8.1f  SALU %7.1f  LDS %6.1f per wavefront (%d wavefronts), %.1f us under the counters' % (
        m, mean(acc['SQ_INSTS_VALU']) / w, mean(acc['SQ_INSTS_SALU']) / w, mean(acc['SQ_INSTS_LDS']) / w, w, mean(dur)))
    for line in open(out + '/sk%s.out' % m):
        if 'k_e16' in line or 'fused' in line: print('      ' + line.rstrip())
PY
