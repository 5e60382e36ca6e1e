#!/usr/bin/env python3
"""solve_on_grid time against the length of the last mesh axis (row pitch = 32 B x points for n = 2):
is the store stream sensitive to rows that are not 1 KiB aligned?"""
import ctypes as C
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
for mesh in ([2049, 2049], [2049, 2048], [2049, 2050], [2049, 2080], [2049, 2112], [2113, 1985], [4097, 1025], [1025, 4097], [1025, 4096]):
    hw, pbc = grid_handle(ctx, m, mesh)
    t = timed(ctx, lambda: _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0])), 8)
    nb = mesh[0] * mesh[1] * 64
    print("mesh %5d x %5d   %.1f us   %.2f TB/s" % (mesh[0], mesh[1], t * 1e3, nb / t / 1e9))
    _lib.check(lib.tbk_wfs_free(hw))
