"""Round 6: eigenvalue-only k lists of 17..32 states -- where does lane-per-matrix QL (k_ql32_lanes) overtake one-thread-per-eigenvalue
bisection (k_tridiag_bisect)?  The rule is 16 matrices per CU (4096).  Device ms of the solve, bisection | QL forced."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
def dev_ms(f):
    f(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.prof_enable(1); ctx.prof_reset(); f(); rep = ctx.prof_report(); ctx.prof_enable(0)
        best = min(best, sum(v["total_ms"] for k, v in rep.items() if k.startswith("solve_list")))
    return round(best, 3)
for n in (17, 24, 32):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
    for nk in (512, 1024, 2048, 4096, 8192, 16384):
        k = np.random.default_rng(nk).uniform(-0.5, 0.5, (nk, 3))
        out = []
        for b in (1, 0):
            with _lib.knob("TBK_QLW_BISECT", b):
                out.append(dev_ms(lambda: m.solve_all(k)))
        print("n %d, %5d k-points: bisection %.3f | QL %.3f ms" % (n, nk, out[0], out[1]), flush=True)
