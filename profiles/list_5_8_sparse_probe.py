#!/usr/bin/env python3
"""Regime probe: models of 5..8 states with MANY lattice vectors but FEW terms (one or two per vector): the R-grouped table is mostly
zeros there -- k list and mesh, ns per point.   python profiles/list_5_8_sparse_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (5, 8, 12, 16):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=16, nhop=100, rmax=3)
    k = m.k_uniform_mesh([33] * 3)
    out = []
    for vec in (False, True):
        m.solve_all(k, eig_vectors=vec)
        ctx.prof_enable(1); ctx.prof_reset(); m.solve_all(k, eig_vectors=vec); rep = ctx.prof_report(); ctx.prof_enable(0)
        out.append(sum(v["total_ms"] for v in rep.values()) * 1e6 / len(k))
    w = tb.wf_array(m, [33] * 3)
    w.solve_on_grid([0.0, 0.0, 0.0]); ctx.sync()
    ctx.timer_begin(); w.solve_on_grid([0.0, 0.0, 0.0]); t = ctx.timer_end()
    print("n=%2d, 100 hoppings over |R| <= 3: list %.1f (eigenvalues) / %.1f (vectors), mesh %.1f ns per point" % (n, out[0], out[1], t * 1e6 / 33 ** 3))
