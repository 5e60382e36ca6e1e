#!/usr/bin/env python3
"""Sweep for cliffs between the Berry kernels' regimes: ns per link / plaquette of berry_phase (determinant and eigenphase forms) and
berry_flux over (components, bands) on a 513 x 257 array, both string directions.   python3 profiles/berry_cliff_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
mesh = [513, 257]
npt = mesh[0] * mesh[1]


def dev_us(fn):
    fn(); ctx.sync(); ctx.prof_enable(1); ctx.prof_reset()
    for _ in range(3): fn()
    rep = ctx.prof_report(); ctx.prof_enable(0)
    return sum(v["total_ms"] for v in rep.values()) / 3 * 1e3


for n in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
    mw = hp.random_model(tb.tb_model, n, 2, 1, 30 + n)
    ww = tb.wf_array(mw, mesh); ww.solve_on_grid([0.0, 0.0])
    for nb in (1, 2, 3, 4, 5, 6, 8):
        if nb > n: continue
        occ = list(range(nb))
        row = [dev_us(lambda: ww.berry_phase(occ, 0, contin=False)), dev_us(lambda: ww.berry_phase(occ, 1, contin=False)),
               dev_us(lambda: ww.berry_phase(occ, 0, contin=False, berry_evals=True)), dev_us(lambda: ww.berry_phase(occ, 1, contin=False, berry_evals=True)),
               dev_us(lambda: ww.berry_flux(occ))]
        print("n %2d nb %d  det d0 %7.1f d1 %7.1f  evals d0 %7.1f d1 %7.1f  flux %7.1f us   (ns per point: %s)" %
              (n, nb, *row, " ".join("%.2f" % (x * 1e3 / npt) for x in row)), flush=True)
    del ww
