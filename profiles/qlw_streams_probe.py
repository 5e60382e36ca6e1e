"""Round 6: 17..32 states, chunks of the batch in flight on side streams (TBK_QLW_STREAMS = 1, 2, 3): whole-call ms of solve_on_grid
(eigenvectors) on 33^3 and 49^3 points and of eigenvalue-only solve_all on 33^3 random points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
def timed(f, rep=5):
    f(); ctx.sync()
    best = 1e9
    for _ in range(rep):
        ctx.sync(); t = time.perf_counter(); f(); ctx.sync(); best = min(best, time.perf_counter() - t)
    return round(best * 1e3, 3)
for n in (17, 24, 32):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
    k = np.random.default_rng(n).uniform(-0.5, 0.5, (33 ** 3, 3))
    for mesh in (33, 49):
        w = tb.wf_array(m, [mesh] * 3)
        out = []
        for ns in (1, 2, 3):
            with _lib.knob("TBK_QLW_STREAMS", ns):
                out.append(timed(lambda: w.solve_on_grid([0.0, 0.0, 0.0])))
        print("n", n, "mesh", mesh, "vectors, streams 1/2/3:", out, flush=True)
    out = []
    for ns in (1, 2, 3):
        with _lib.knob("TBK_QLW_STREAMS", ns):
            out.append(timed(lambda: m.solve_all(k)))
    print("n", n, "eigenvalues of 33^3 listed points (host copies included), streams 1/2/3:", out, flush=True)
