#!/usr/bin/env python3
"""berry_flux / berry_phase of a 3-D array along every choice of directions: ns per plaquette / link.
    python profiles/berry_dirs_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n, nspin, occ in ((2, 1, [0]), (2, 2, [0, 1]), (8, 1, [0, 1, 2, 3])):
    m = hp.random_model(tb.tb_model, n, 3, nspin, seed=5 + n, nhop=4 * n, rmax=1)
    side = 129 if n * nspin <= 4 else 65
    w = tb.wf_array(m, [side] * 3)
    w.solve_on_grid([0.0, 0.0, 0.0])
    out = {"states": n * nspin, "occ": len(occ), "side": side}
    for dirs in ([0, 1], [1, 0], [0, 2], [2, 0], [1, 2], [2, 1]):
        w.berry_flux(occ, dirs)
        ctx.timer_begin(); w.berry_flux(occ, dirs); t = ctx.timer_end()
        out["flux%d%d_ns_per_plaq" % tuple(dirs)] = round(1e6 * t / side ** 3, 4)
    for d in (0, 1, 2):
        w.berry_phase(occ, d, contin=False)
        ctx.timer_begin(); w.berry_phase(occ, d, contin=False); t = ctx.timer_end()
        out["phase%d_ns_per_link" % d] = round(1e6 * t / side ** 3, 4)
        if len(occ) > 1:
            w.berry_phase(occ, d, contin=False, berry_evals=True)
            ctx.timer_begin(); w.berry_phase(occ, d, contin=False, berry_evals=True); t = ctx.timer_end()
            out["wilson%d_ns_per_link" % d] = round(1e6 * t / side ** 3, 4)
    print(json.dumps(out))
