"""Round 6: where does the direct path for 17..32 states (k_hh32 + k_ql32_lanes + k_tw32_vectors) overtake the workgroup Jacobi kernels
on SMALL batches?  The dispatcher's rule is 8 matrices per CU (2048); TBK_QLW_MIN=0 forces the direct path.  k lists of a random
model, device ms of the solve (the context's brackets), with eigenvectors and without."""
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
    for nk in (128, 256, 512, 1024, 2048, 4096):
        k = np.random.default_rng(nk).uniform(-0.5, 0.5, (nk, 3))
        row = []
        for vec in (True, False):
            a = dev_ms(lambda: m.solve_all(k, eig_vectors=vec))
            with _lib.knob("TBK_QLW_MIN", 0):
                b = dev_ms(lambda: m.solve_all(k, eig_vectors=vec))
            row.append((a, b))
        print("n %d, %5d k-points: with vectors default %.3f / direct %.3f ms; eigenvalues default %.3f / direct %.3f" % (n, nk, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
