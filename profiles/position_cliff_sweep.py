#!/usr/bin/env python3
"""Sweep for cliffs in position_hwf_mesh (hybrid Wannier centres of nocc bands at every point of a 257 x 129 array of n-state models,
finite along direction 2 of a 3-D lattice): device us per call and ns per point.   python3 profiles/position_cliff_sweep.py"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
mesh = [257, 129]
npt = mesh[0] * mesh[1]
for nl in (2, 3, 4, 6, 8, 12, 16, 20, 24, 32):
    with contextlib.redirect_stdout(io.StringIO()):
        m3 = tb.tb_model(3, 3, np.identity(3), [[0, 0, 0]])
        for Rv in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
            m3.set_hop(-1.0, 0, 0, Rv)
        m3.set_hop(0.3j, 0, 0, [1, 1, 0])
        slab = m3.cut_piece(nl, 2, glue_edgs=False)
    ws = tb.wf_array(slab, mesh)
    ws.solve_on_grid([0.0, 0.0])
    for nocc in sorted(set([1, 2, max(1, nl // 2), nl])):
        if nocc > nl: continue
        occ = list(range(nocc))
        ws.position_hwf_mesh(occ, 2); ctx.sync(); ctx.prof_enable(1); ctx.prof_reset()
        for _ in range(3): ws.position_hwf_mesh(occ, 2)
        rep = ctx.prof_report(); ctx.prof_enable(0)
        tot = sum(v["total_ms"] for v in rep.values()) / 3 * 1e3
        print("n %2d nocc %2d  %8.1f us  %6.2f ns per point  %s" % (nl, nocc, tot, tot * 1e3 / npt, {k: round(v["total_ms"] / 3 * 1e3, 1) for k, v in rep.items()}), flush=True)
    del ws
