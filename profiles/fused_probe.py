#!/usr/bin/env python3
"""solve_on_grid + berry_flux in one pass (tbk_wfs_solve_grid_flux_async) against the two calls: stored array bit for bit,
min gaps, flux totals; timings at 2048^2 and 4096^2.    python profiles/fused_probe.py"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
lib, ctx = _lib.lib, _lib.default_context()

def cases():
    hal = hp.haldane(tb.tb_model, 0.0)
    km = hp.kane_mele(tb.tb_model, "odd")
    yield "haldane 65x65 occ0", hal, [65, 65], [0]
    yield "haldane 130x77 occ1", hal, [130, 77], [1]
    yield "haldane 7x300 occ0", hal, [7, 300], [0]
    yield "haldane 300x5 occ0", hal, [300, 5], [0]
    yield "haldane 2x2 occ0", hal, [2, 2], [0]
    yield "haldane 33x64 both", hal, [33, 64], [0, 1]
    yield "haldane 33x129 occ0", hal, [33, 129], [0]
    yield "kane-mele 40x70 occ01", km, [40, 70], [0, 1]
    yield "kane-mele 70x200 occ2", km, [70, 200], [2]
    yield "kane-mele 9x513 occ1,3", km, [9, 513], [1, 3]

for tag, m, mesh, occ in cases():
    start = [-0.5, 0.13]
    w1 = tb.wf_array(m, mesh); g1 = w1.solve_on_grid(start); f1 = w1.berry_flux(occ)
    w2 = tb.wf_array(m, mesh); g2, f2 = w2.solve_on_grid_flux(start, occ)
    a1, a2 = w1.to_host(), w2.to_host()
    same = float(np.max(np.abs(a1 - a2)))          # (two kernels: the compiler contracts a b - c d differently, 1 ulp apart)
    print(json.dumps({"case": tag, "array_max_diff": same, "gaps_equal": bool(np.array_equal(g1, g2)), "flux_two_calls": f1,
                      "flux_diff": abs(f1 - f2), "plaq": (mesh[0] - 1) * (mesh[1] - 1)}), flush=True)
    for R in (1, 2, 7):
        with _lib.knob("TBK_FUSED_ROWS", R), _lib.knob("TBK_GRID_SEG", 1 + R % 3):
            w3 = tb.wf_array(m, mesh); g3, f3 = w3.solve_on_grid_flux(start, occ)
            # the fused kernel's own bits do not depend on its tiling
            assert np.array_equal(a2, w3.to_host()) and np.array_equal(g3, g2) and abs(f3 - f1) < 1e-11 * max(1.0, ((mesh[0] - 1) * (mesh[1] - 1)) ** 0.5), (tag, R, f1, f3)

# timings
sys.path.insert(0, ROOT)
import bench
hal = bench.haldane(tb)
for side in (2049, 4097):
    g = bench.Grid(lib, _lib, ctx, hal, [side, side])
    occ = np.array([0], dtype=np.int32)
    start = np.array([-0.5, -0.5])
    def two():
        g.solve(start); g.flux(occ)
    def fused():
        _lib.check(lib.tbk_wfs_solve_grid_flux_async(g.h, g.hm, _lib.dptr(start), _lib.dptr(g.pbc.view(float)), 0, side, _lib.iptr(occ), 1))
    res = {"side": side}
    for name, fn in (("two_calls", two), ("fused", fused)):
        for _ in range(5): fn()
        ctx.sync(); t0 = time.perf_counter()
        nrep = 200 if side < 3000 else 50
        for _ in range(nrep): fn()
        ctx.sync(); dt = (time.perf_counter() - t0) / nrep
        res[name + "_us"] = dt * 1e6
        res[name + "_chern"] = float(g.flux_total()[0] / (2 * np.pi))
        res[name + "_gap"] = float(g.gaps()[0])
    for R in (2, 3, 4, 6, 8):
        with _lib.knob("TBK_FUSED_ROWS", R):
            for _ in range(5): fused()
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(50): fused()
            ctx.sync(); res["fused_R%d_us" % R] = (time.perf_counter() - t0) / 50 * 1e6
    print(json.dumps(res), flush=True)
    g.free()
