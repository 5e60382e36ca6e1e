"""k_hh32 with and without the accumulation of Z (VEC): solve_on_grid against solve_all_mesh-style eigenvalues only on 33^3 points;
run under rocprofv3 --kernel-trace --stats (profiles/tw32_prof.sh pattern) to read the kernels' own times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (24, 32):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
    k = np.random.default_rng(n).uniform(-0.5, 0.5, (33 ** 3, 3))
    for vec in (False, True):
        m.solve_all(k, eig_vectors=vec); ctx.sync()
        ctx.prof_enable(1); ctx.prof_reset(); m.solve_all(k, eig_vectors=vec); rep = ctx.prof_report(); ctx.prof_enable(0)
        print(n, vec, {a: round(v["total_ms"], 3) for a, v in rep.items()})
