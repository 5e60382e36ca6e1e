#!/usr/bin/env python3
"""Config E's solve_on_grid through the library: per-kernel HIP-event brackets of one call on a side^3 mesh (the fused kernel
k_e16 by default, round 3's three kernels with TBK_E16=0), and how many matrices went to the QL-replay fallback.
    python profiles/e16_probe.py [side = 65] [reps = 3]"""
import contextlib, io, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
import ctypes as C
ctx = _lib.default_context()
lib = _lib.lib
side = int(sys.argv[1]) if len(sys.argv) > 1 else 65
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
mesh = [side] * 3
n = m._nsta
h = C.c_void_p()
m32 = np.ascontiguousarray(mesh, dtype=np.int32)
_lib.check(lib.tbk_wfs_create(ctx.handle, 3, _lib.iptr(m32), n, n, C.byref(h)))
pbc = np.ascontiguousarray(np.array([np.exp(-2j * np.pi * m._orb[:, m._per[d]]) for d in range(3)]))
hm = m._device_model()
start = np.zeros(3)
call = lambda: _lib.check(lib.tbk_wfs_solve_grid_async(h, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0]))
call(); ctx.sync()
best = None
for _ in range(reps):
    ctx.prof_enable(1); ctx.prof_reset()
    ctx.timer_begin(); call(); t = ctx.timer_end()
    r = ctx.prof_report(); ctx.prof_enable(0)
    if best is None or t < best[0]:
        best = (t, r)
gaps = np.zeros(n - 1)
_lib.check(lib.tbk_wfs_solve_grid_result(h, _lib.dptr(gaps)))
print(json.dumps({"side": side, "points": side ** 3, "call_ms": best[0], "kpts_per_s": (side - 1) ** 3 / best[0] * 1e3,
                  "kernels": {k: {"launches": v["launches"], "total_ms": v["total_ms"]} for k, v in best[1].items()},
                  "gap78": float(gaps[7]), "TBK_E16": os.environ.get("TBK_E16", "1")}))
