#!/usr/bin/env python3
"""Config E's berry_phase(range(nocc), dir) on a side^3 cubic16 array: per-kernel HIP-event brackets, the product form
(k_chain_prod_tile, default) against link determinants through the workspace (TBK_CHAIN_PROD=0), and the phases of the two.
    python profiles/chain_prod_probe.py [side = 129] [nocc = 8] [dir = 2]"""
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
side = int(sys.argv[1]) if len(sys.argv) > 1 else 129
nocc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
w = tb.wf_array(m, [side] * 3)
w.solve_on_grid([0.0, 0.0, 0.0])
h = w._ensure_dev()
occ32 = np.arange(nocc, dtype=np.int32)
out = {}
res = {}
for prod in (1, 0):
    with _lib.knob("TBK_CHAIN_PROD", prod):
        ph = np.zeros(side * side)
        _lib.check(lib.tbk_berry_phase(h, _lib.iptr(occ32), nocc, dr, 0, _lib.dptr(ph)))
        best = None
        for _ in range(3):
            ctx.prof_enable(1); ctx.prof_reset()
            ctx.timer_begin()
            _lib.check(lib.tbk_berry_phase(h, _lib.iptr(occ32), nocc, dr, 0, _lib.dptr(ph)))
            t = ctx.timer_end()
            r = ctx.prof_report(); ctx.prof_enable(0)
            if best is None or t < best[0]:
                best = (t, r)
        res[prod] = ph.copy()
        out["prod" if prod else "link_dets"] = {"call_ms": best[0], "kernels": {k: round(v["total_ms"], 4) for k, v in best[1].items()}}
d = np.angle(np.exp(1j * (res[1] - res[0])))
out["max_phase_difference"] = float(np.abs(d).max())
out["side"], out["nocc"], out["dir"] = side, nocc, dr
out["links"] = side * side * (side - 1)
out["algorithmic_GB"] = 16 * nocc * 16 * side ** 3 / 1e9
print(json.dumps(out))
