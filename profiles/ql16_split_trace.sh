#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 durations of the kernels of the three-kernel n = 9..16 form on config E's 64^3 sub-mesh
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ql16s_prof
TBK_QL16_SPLIT=${1:-1} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ql16s_prof -- python3 $R/profiles/ql16_split_probe.py child > $R/gpurun_out/ql16s_prof.out 2> $R/gpurun_out/ql16s_prof.err
cd $R
python3 - <<'PY'
import csv, re, glob
f = sorted(glob.glob('gpurun_out/ql16s_prof/*/*_kernel_trace.csv'))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    nm = r['Kernel_Name']
    if 'ql16' in nm:
        print(re.sub(r'\(.*', '', nm)[:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'grid', r.get('Grid_Size_X', r.get('Grid_Size')))
PY
