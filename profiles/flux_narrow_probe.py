#!/usr/bin/env python3
"""berry_flux of 1..4 bands of states with 5..7 components (k_flux: a thread per plaquette) -- kernel brackets.
    python3 profiles/flux_narrow_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n, mesh in ((6, [1025, 257]), (6, [513, 513]), (5, [65, 65, 65])):
    mw = hp.random_model(tb.tb_model, n, len(mesh), 1, 7 + n // 2)
    ww = tb.wf_array(mw, mesh); ww.solve_on_grid([0.0] * len(mesh))
    for nb in (1, 2, 3, 4):
        occ = list(range(nb))
        ww.berry_flux(occ); ctx.sync()
        ctx.prof_enable(1); ctx.prof_reset()
        for _ in range(3): ww.berry_flux(occ)
        rep = ctx.prof_report(); ctx.prof_enable(0)
        print(n, mesh, nb, {k: round(v["total_ms"] / 3 * 1e3, 1) for k, v in rep.items()}, flush=True)
