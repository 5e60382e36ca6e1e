#!/bin/bash
# Run on the GPU box from the repo root: probe (accuracy + scope timings) and a rocprofv3 kernel trace of the same probe;
# prints the per-kernel durations of the last batches.  bash profiles/qlw_trace.sh "32,64"
R=$(pwd)
SIZES=${1:-"24,32,48,64"}
TBK_QLW=1 timeout 600 python3 profiles/qlw_probe.py child > gpurun_out/qlw_probe.txt 2>&1
tail -1 gpurun_out/qlw_probe.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/qlw_prof
QLW_SIZES=$SIZES timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qlw_prof -- python3 $R/profiles/qlw_probe.py child > $R/gpurun_out/qlw_prof.out 2> $R/gpurun_out/qlw_prof.err
cd $R
python3 - <<'PY'
import csv, re, glob
f = sorted(glob.glob('gpurun_out/qlw_prof/*/*_kernel_trace.csv'))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    nm = r['Kernel_Name']
    if 'tridiag' in nm or 'backtransform' in nm or 'replay' in nm:
        print(re.sub(r'\(.*', '', nm)[:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'grid', r.get('Grid_Size_X', r.get('Grid_Size')), 'vgpr', r.get('VGPR_Count'))
PY
