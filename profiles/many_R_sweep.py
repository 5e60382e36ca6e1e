#!/usr/bin/env python3
"""Regime sweep: solve_on_grid, ns per mesh point, for sparse (|R| <= 1) and dense (|R| <= 2, 60 n hoppings) random models of
n states on a side^3 mesh -- looks for cliffs between the kernels' assembly forms.   python profiles/many_R_sweep.py [side = 33]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
side = int(sys.argv[1]) if len(sys.argv) > 1 else 33
ctx = _lib.default_context()
for n in (2, 3, 4, 5, 6, 8, 9, 12, 16, 17, 24, 32):
    row = {}
    for name, model in (("sparse", hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)),
                        ("dense", hp.random_model(tb.tb_model, n, 3, 1, seed=6, nhop=60 * n, rmax=2))):
        w = tb.wf_array(model, [side] * 3)
        w.solve_on_grid([0.0, 0.0, 0.0]); ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_begin(); w.solve_on_grid([0.0, 0.0, 0.0]); best = min(best, ctx.timer_end())
        row[name] = best * 1e6 / side ** 3
        del w
    print("n = %2d: sparse %7.2f ns per point, dense %7.2f  (x %.1f)" % (n, row["sparse"], row["dense"], row["dense"] / row["sparse"]))
