#!/usr/bin/env python3
"""Eigenvalue-only k lists of a spinful 16-state model WITHOUT spin-orbit coupling (every level double: every lane of k_e16<.., false>
takes the bisection rescue) against a generic 16-state model, and both on round 3's pair of kernels (TBK_E16_EVALS=0).
    python profiles/evals16_twins_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
rng = np.random.default_rng(4)
tw = hp.quiet(tb.tb_model, 3, 3, np.identity(3), rng.random((8, 3)), nspin=2)
tw.set_onsite(list(rng.standard_normal(8)))
for R in ([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]):
    for i in range(8):
        for j in range(8):
            if (R != [0, 0, 0] or i < j) and rng.random() < 0.5:
                tw.set_hop(0.3 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
ge = hp.random_model(tb.tb_model, 16, 3, 1, seed=5, nhop=96, rmax=1)
for name, m in (("twins (8 orbitals x spin, no SOC)", tw), ("generic 16 states", ge)):
    k = m.k_uniform_mesh([48] * 3)
    for knob in (1, 0):
        with _lib.knob("TBK_E16_EVALS", knob):
            m.solve_all(k)
            ctx.prof_enable(1); ctx.prof_reset(); ev = m.solve_all(k); rep = ctx.prof_report(); ctx.prof_enable(0)
        t = sum(v["total_ms"] for v in rep.values())
        print("%-36s TBK_E16_EVALS=%d: %.3f ms = %.2f ns per point, checksum %.10f" % (name, knob, t, t * 1e6 / len(k), ev.sum()))
