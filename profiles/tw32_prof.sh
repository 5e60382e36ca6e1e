#!/bin/bash
# per-kernel times of the 17..32-state path (profiles/n17_probe.py) under rocprofv3; summary to gpurun_out/tw32prof/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tw32prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tw32prof -- python3 $R/profiles/${1:-n17_probe.py} > $R/gpurun_out/tw32prof.out 2> $R/gpurun_out/tw32prof.err
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/tw32prof/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
with open("gpurun_out/tw32prof/summary.txt","w") as o:
    for r in rows[:24]:
        line="%-110s calls %4s total %10s avg %10s" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
        print(line); o.write(line+"\n")
PY
