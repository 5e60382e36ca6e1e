#!/usr/bin/env python3
"""solve_all on k lists of path-like sizes for models of 3..128 states: ms per call (kernel brackets summed), eigenvalues only and
with eigenvectors -- where the dispatch changes regime the time should not jump.   python profiles/list_sizes_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
rng = np.random.default_rng(0)
for n in (3, 4, 6, 8, 9, 12, 16, 17, 24, 32, 48, 64, 65, 100, 128):
    m = hp.random_model(tb.tb_model, n, 2, 1, seed=n, nhop=4 * n, rmax=1)
    row = {"n": n}
    for nk in (100, 1000, 3000, 10000, 40000):
        k = rng.random((nk, 2))
        for vec in (False, True):
            m.solve_all(k, eig_vectors=vec)
            ctx.timer_begin(); m.solve_all(k, eig_vectors=vec); t = ctx.timer_end()
            row["%d%s" % (nk, "v" if vec else "e")] = round(t, 3)
    print(json.dumps(row))
