#!/usr/bin/env python3
"""det-form berry_phase of 1..4 bands of NARROW states (fewer than 8 components: k_chain_partial, a thread per (string, segment))
next to the eigenphase form of the same bands: kernel brackets on leg P's arrays.   python3 profiles/det_narrow_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n, mesh in ((6, [1025, 257]), (6, [257, 1025]), (4, [1025, 257])):
    mw = hp.random_model(tb.tb_model, n, 2, 1, 7 + n // 2)
    ww = tb.wf_array(mw, mesh); ww.solve_on_grid([0.0, 0.0])
    d = 0 if mesh[0] > mesh[1] else 1
    for nb in (1, 2, 3, 4):
        if nb > n: continue
        for ev in (False, True):
            occ = list(range(nb))
            ww.berry_phase(occ, d, contin=False, berry_evals=ev); ctx.sync()
            ctx.prof_enable(1); ctx.prof_reset()
            for _ in range(3): ww.berry_phase(occ, d, contin=False, berry_evals=ev)
            rep = ctx.prof_report(); ctx.prof_enable(0)
            print(n, mesh, d, nb, "evals" if ev else "det", {k: round(v["total_ms"] / 3 * 1e3, 1) for k, v in rep.items()}, flush=True)
