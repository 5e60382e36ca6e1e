#!/usr/bin/env python3
"""The fused step with the flux total inside the seam kernel (default) against a kernel of its own (TBK_FUSED_SUM=0):
same totals, us per back-to-back step."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pythtb_amd as tb
from pythtb_amd import _lib
import bench
lib, ctx = _lib.lib, _lib.default_context()
for side in (65, 2049, 4097):
    g = bench.Grid(lib, _lib, ctx, bench.haldane(tb), [side, side])
    occ = np.array([0], dtype=np.int32)
    start = [-0.5, -0.5]
    for _ in range(2000 if side < 3000 else 300):
        g.solve_flux(start, occ)
    res = {"side": side}
    for rep in range(2):
        for mode in (1, 0):
            with _lib.knob("TBK_FUSED_SUM", mode):
                for _ in range(20): g.solve_flux(start, occ)
                ctx.sync(); t0 = time.perf_counter()
                n = 500 if side < 3000 else 100
                for _ in range(n): g.solve_flux(start, occ)
                ctx.sync(); res["sum%d_rep%d_us" % (mode, rep)] = round((time.perf_counter() - t0) / n * 1e6, 2)
                res["sum%d_chern" % mode] = float(g.flux_total()[0] / (2 * np.pi))
    print(json.dumps(res), flush=True)
    g.free()
