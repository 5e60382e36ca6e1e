#!/usr/bin/env python3
"""Headline kernels past the last-level cache (Haldane 4096^2, 1.07 GB of eigenvectors): tile-shape sweeps of
k_flux_rows (TBK_FLUX_TI) and k_grid_rows (TBK_GRID_SEG), HIP-event brackets around every launch."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
import pythtb_amd as tb
from pythtb_amd import _lib
lib, ctx = _lib.lib, _lib.default_context()
ev = ctx.prof_calibrate(50)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = bench.haldane(tb)
g = bench.Grid(lib, _lib, ctx, m, [side + 1, side + 1])
occ = np.array([0], dtype=np.int32)
def step():
    g.solve([-0.5, -0.5]); g.flux(occ)
step(); ctx.sync()
npt = side * side
for name, vals in (("TBK_FLUX_TI", [None, 2, 4, 8, 16, 32, 64]), ("TBK_GRID_SEG", [None, 1, 2, 4, 8, 16])):
    for v in vals:
        with _lib.knob(name, v):
            step(); ctx.sync()
            kt = bench.kernel_times(ctx, step, 10, ev)
        print("%-12s %-5s solve %7.1f us (%.2f TB/s)   flux %7.1f us (%.2f TB/s)   chern %.12f" % (
            name, v, kt["solve_grid"]["avg_bracket_ms"] * 1e3, 64 * npt / kt["solve_grid"]["avg_bracket_ms"] / 1e9,
            kt["berry_flux"]["avg_bracket_ms"] * 1e3, 32 * npt / kt["berry_flux"]["avg_bracket_ms"] / 1e9,
            g.flux_total()[0] / (2 * np.pi)))
