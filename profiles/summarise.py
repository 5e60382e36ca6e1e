#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into small per-kernel tables."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    """Kernel name WITH its template arguments (k_grid_rows<2, 1> and <4, 1> are different kernels), without
    the return type and the parameter list."""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].replace(" ", "")[:80]


# kernel trace -> durations
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "trace*", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
with open(os.path.join(dst, "kernel_stats.csv"), "w") as out:
    out.write("kernel,calls,total_us,avg_us,min_us,max_us\n")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        out.write("%s,%d,%.1f,%.2f,%.2f,%.2f\n" % (k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))

# the tool's own stats file, verbatim
for f in glob.glob(os.path.join(src, "trace*", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "rocprofv3_kernel_stats.csv"), "w").write(open(f).read())

# PMC passes -> per kernel mean of each counter per dispatch
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
table = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}
json.dump(table, open(os.path.join(dst, "pmc_per_dispatch.json"), "w"), indent=1, sort_keys=True)

# HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B requests at 64 B -> x2 for wide streaming reads.
traffic = {}
for k, cs in table.items():
    if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
        rd = cs.get("FETCH_SIZE", 0.0) * 1024.0
        wr = cs.get("WRITE_SIZE", 0.0) * 1024.0
        traffic[k] = {"fetch_bytes_raw": rd, "fetch_bytes_x2": 2.0 * rd, "write_bytes": wr,
                      "hbm_bytes_per_launch": 2.0 * rd + wr}
# aliases under the names bench.py uses for its HIP-event brackets
for alias, names in (("solve_grid_flux", ("k_grid_rows_flux",)), ("solve_grid", ("k_grid_rows<", "k_grid_small", "k_solve_wave")),
                     ("berry_flux", ("k_flux_rows", "k_flux<"))):
    for nm in names:
        for full in sorted(traffic):
            if full.startswith(nm) and alias not in traffic:
                traffic[alias] = dict(traffic[full], kernel=full)
# where the numbers come from (bench.py quotes it next to roofline.traffic)
for k in traffic:
    traffic[k]["source"] = "profiles/%s/pmc_per_dispatch.json" % os.path.basename(os.path.normpath(dst)).replace("summary_", "")
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, "kernel_stats.csv")).read())
print(json.dumps(traffic, indent=1))
