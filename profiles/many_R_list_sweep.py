#!/usr/bin/env python3
"""Regime sweep on k LISTS: solve_all (eigenvalues / with eigenvectors), ns per k-point of the kernels, for sparse (|R| <= 1) and dense
(|R| <= 2, 60 n hoppings) random models of n states on the 33^3 points of k_uniform_mesh.   python profiles/many_R_list_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (2, 3, 4, 5, 8, 9, 12, 16, 17, 24):
    row = []
    for name, model in (("sparse", hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)),
                        ("dense", hp.random_model(tb.tb_model, n, 3, 1, seed=6, nhop=60 * n, rmax=2))):
        k = model.k_uniform_mesh([33] * 3)
        for vec in (False, True):
            model.solve_all(k, eig_vectors=vec)
            ctx.prof_enable(1); ctx.prof_reset(); model.solve_all(k, eig_vectors=vec); rep = ctx.prof_report(); ctx.prof_enable(0)
            row.append(sum(v["total_ms"] for v in rep.values()) * 1e6 / len(k))
    print("n = %2d: eigenvalues sparse %6.2f dense %6.2f | with vectors sparse %6.2f dense %6.2f  ns per point" % (n, row[0], row[2], row[1], row[3]))
