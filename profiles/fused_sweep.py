#!/usr/bin/env python3
"""fused solve + flux step: rows per tile x chunks per tile (TBK_FUSED_ROWS x TBK_GRID_SEG), us per step.  python profiles/fused_sweep.py [side]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import bench
lib, ctx = _lib.lib, _lib.default_context()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
g = bench.Grid(lib, _lib, ctx, bench.haldane(tb), [side, side])
occ = np.array([0], dtype=np.int32); start = np.array([-0.5, -0.5])
def fused():
    _lib.check(lib.tbk_wfs_solve_grid_flux_async(g.h, g.hm, _lib.dptr(start), _lib.dptr(g.pbc.view(float)), 0, side, _lib.iptr(occ), 1))
out = {}
for R in (3, 4, 5, 6, 8, 10, 12):
    for seg in (1, 2, 3, 4):
        with _lib.knob("TBK_FUSED_ROWS", R), _lib.knob("TBK_GRID_SEG", seg):
            for _ in range(10): fused()
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(300): fused()
            ctx.sync(); out["R%d_seg%d" % (R, seg)] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
print(json.dumps(out))
