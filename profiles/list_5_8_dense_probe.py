import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (5, 6, 8):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=6, nhop=60 * n, rmax=2)
    k = m.k_uniform_mesh([33] * 3)
    for vec in (False, True):
        m.solve_all(k, eig_vectors=vec)
        ctx.prof_enable(1); ctx.prof_reset(); m.solve_all(k, eig_vectors=vec); rep = ctx.prof_report(); ctx.prof_enable(0)
        kern = sum(v["total_ms"] for v in rep.values())
        print("n=%d dense list vectors=%d: %.3f ms = %.1f ns per point" % (n, vec, kern, kern * 1e6 / len(k)))
