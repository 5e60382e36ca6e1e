#!/bin/bash
# Where do the stores of the fused kernel wait?  L2 -> fabric (TCC_EA0_WRREQ_*), texture path (TA / TCP) and SQ counters for
# the headline step and, beside it, the same store stream without arithmetic (microbench/store_pattern).
#   bash profiles/pmc_store_path.sh <tag>      (GPU box, repo root; raw output under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-store}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
# (few TCC counters per pass: "Request exceeds the capabilities of the hardware" aborts the run otherwise; every run under
# `timeout`: a profiler that aborts can hang for the rest of the call)
for pass in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE" \
            "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
            "TCC_BUSY_sum TCC_TAG_STALL_sum" \
            "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
            "SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/bench_p$i -- python3 $REPO/bench.py --steps 20 --warmup 3 --preheat-s 0 --no-cpu-baseline --headline-only > /dev/null 2> $OUT/bench_p$i.err
  timeout 60 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pat_p$i -- $REPO/profiles/microbench/store_pattern > /dev/null 2> $OUT/pat_p$i.err
done
cd $REPO
python3 - <<PY
import csv, glob, collections, json
out = {}
for tag in ("bench", "pat"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        out[tag + ": " + k] = {c: {"per_dispatch": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in cs.items():
        print("   %-40s %14.4g  (%d)" % (c, v["per_dispatch"], v["n"]))
PY
