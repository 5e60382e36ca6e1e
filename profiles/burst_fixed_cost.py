#!/usr/bin/env python3
"""What a K-step timed region of bench.py pays besides K steps: median wall time of [sync; n steps; sync] against n.
python profiles/burst_fixed_cost.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pythtb_amd as tb
from pythtb_amd import _lib
import bench
lib, ctx = _lib.lib, _lib.default_context()
g = bench.Grid(lib, _lib, ctx, bench.haldane(tb), [2049, 2049])
occ = np.array([0], dtype=np.int32)
start = [-0.5, -0.5]
for _ in range(3000):
    g.solve_flux(start, occ)
ctx.sync()
rows = []
for n in (0, 1, 2, 5, 10, 20, 50, 100, 200, 1000):
    ts = []
    for rep in range(15):
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            g.solve_flux(start, occ)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        ts.append((t2 - t0, t1 - t0))
    ts.sort()
    tot, enq = ts[len(ts) // 2]
    rows.append({"n": n, "total_us": tot * 1e6, "enqueue_us": enq * 1e6, "per_step_us": tot * 1e6 / max(n, 1)})
    print(json.dumps(rows[-1]), flush=True)
