import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (17, 18, 19, 20, 21, 22, 23, 24, 26, 28, 30, 32):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
    w = tb.wf_array(m, [33] * 3)
    out = []
    for hh in (1, 0):
        with _lib.knob("TBK_HH32", hh):
            w.solve_on_grid([0.0, 0.0, 0.0]); ctx.sync()
            ctx.prof_enable(1); ctx.prof_reset(); w.solve_on_grid([0.0, 0.0, 0.0]); rep = ctx.prof_report(); ctx.prof_enable(0)
        out.append(round(rep["solve_grid"]["total_ms"], 3))
    print(n, "hh32", out[0], "lds", out[1], flush=True)
