#!/usr/bin/env python3
"""Kane-Mele (4 states) 4097 x 513: solve_on_grid + berry_flux([0,1]) as two calls against the fused pass."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
m = hp.kane_mele(tb.tb_model, "odd")
w = tb.wf_array(m, [4097, 513])
start = [-0.5, -0.5]
def two():
    w.solve_on_grid(start); return w.berry_flux([0, 1])
def fused():
    return w.solve_on_grid_flux(start, [0, 1])[1]
res = {}
for name, fn in (("two_calls", two), ("fused", fused), ("two_calls_again", two), ("fused_again", fused)):
    for _ in range(3): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(20): f = fn()
    ctx.sync(); res[name + "_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4); res[name + "_flux"] = float(f)
for R in (2, 3, 4, 6, 8):
    with _lib.knob("TBK_FUSED_ROWS", R):
        for _ in range(3): fused()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(20): fused()
        ctx.sync(); res["fused_R%d_ms" % R] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
print(json.dumps(res))
