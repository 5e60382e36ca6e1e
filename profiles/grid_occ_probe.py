#!/usr/bin/env python3
"""k_grid_rows (solve_on_grid, 2 states) with its resident wavefronts per SIMD capped (TBK_GRID_OCC) at 2048^2 and 4096^2."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pythtb_amd as tb
from pythtb_amd import _lib
import bench
lib, ctx = _lib.lib, _lib.default_context()
for side in (2049, 4097):
    g = bench.Grid(lib, _lib, ctx, bench.haldane(tb), [side, side])
    start = [-0.5, -0.5]
    occ = np.array([0], dtype=np.int32)
    for _ in range(1000 if side < 3000 else 200): g.solve(start)
    res = {"side": side}
    for o in (0, 3, 4, 5, 6, 0):
        with _lib.knob("TBK_GRID_OCC", o):
            for _ in range(20): g.solve(start)
            ctx.sync(); t0 = time.perf_counter()
            n = 500 if side < 3000 else 100
            for _ in range(n): g.solve(start)
            ctx.sync(); res.setdefault("occ%d_us" % o, []).append(round((time.perf_counter() - t0) / n * 1e6, 2))
    print(json.dumps(res), flush=True)
    g.free()
