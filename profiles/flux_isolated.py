#!/usr/bin/env python3
"""k_flux_rows at 4096^2 right after solve_on_grid wrote the array (the write-back of 1.07 GB is still draining)
against the same launch repeated on a settled array."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
import pythtb_amd as tb
from pythtb_amd import _lib
lib, ctx = _lib.lib, _lib.default_context()
ev = ctx.prof_calibrate(50)
for side in (2048, 4096):
    m = bench.haldane(tb)
    g = bench.Grid(lib, _lib, ctx, m, [side + 1, side + 1])
    occ = np.array([0], dtype=np.int32)
    def both():
        g.solve([-0.5, -0.5]); g.flux(occ)
    both(); ctx.sync()
    a = bench.kernel_times(ctx, both, 10, ev)["berry_flux"]["avg_bracket_ms"]
    b = bench.kernel_times(ctx, lambda: g.flux(occ), 10, ev)["berry_flux"]["avg_bracket_ms"]
    s = bench.kernel_times(ctx, lambda: g.solve([-0.5, -0.5]), 10, ev)["solve_grid"]["avg_bracket_ms"]
    npt = side * side
    print("%d^2: flux after solve %.1f us (%.2f TB/s) | flux after flux %.1f us (%.2f TB/s) | solve after solve %.1f us (%.2f TB/s)" % (
        side, a * 1e3, 32 * npt / a / 1e9, b * 1e3, 32 * npt / b / 1e9, s * 1e3, 64 * npt / s / 1e9))
    g.free()
