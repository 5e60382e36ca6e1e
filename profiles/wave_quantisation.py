#!/usr/bin/env python3
"""solve_on_grid time against the number of mesh rows (2049 points each, 5 one-wave tiles per row):
does the time step at multiples of the resident-wave capacity (a tail effect) or grow linearly?"""
import os
import sys

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
from pythtb_amd import _lib  # noqa: E402
import helpers as hp  # noqa: E402
from bench_configs import grid_handle, timed  # noqa: E402

lib, ctx = _lib.lib, _lib.default_context()
m = hp.haldane(tb.tb_model, 0.0)
hm = m._device_model()
start = np.array([-0.5, -0.5])
for rows in ((int(os.environ["WQ_ROWS"]),) if os.environ.get("WQ_ROWS") else (205, 410, 615, 820, 1024, 1229, 1434, 1638, 1843, 2049, 2254, 2458, 2663, 2868, 3072, 3277, 3482, 3686, 4097)):
    mesh = [rows, 2049]
    hw, pbc = grid_handle(ctx, m, mesh)
    t = timed(ctx, lambda: _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0])), int(os.environ.get("WQ_REPS", "8")))
    nb = mesh[0] * mesh[1] * 64
    print("rows %5d  tiles %6d (%.2f x 6144)   %.1f us   %.2f TB/s   %.2f ns/tile" % (rows, rows * 5, rows * 5 / 6144, t * 1e3, nb / t / 1e9, t * 1e6 / (rows * 5)))
    _lib.check(lib.tbk_wfs_free(hw))
