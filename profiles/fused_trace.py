#!/usr/bin/env python3
"""the fused solve + flux step, 200 launches at 2048^2 (for rocprofv3 --kernel-trace --stats / --pmc)"""
import os, sys
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
for _ in range(200):
    _lib.check(lib.tbk_wfs_solve_grid_flux_async(g.h, g.hm, _lib.dptr(start), _lib.dptr(g.pbc.view(float)), 0, side, _lib.iptr(occ), 1))
ctx.sync()
print(g.flux_total()[0] / (2 * np.pi))
