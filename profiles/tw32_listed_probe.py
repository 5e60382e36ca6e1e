"""Round 6: which matrices does k_tw32_vectors put on its list, and why?  H(k) of the random 24-state model of tw32_sweep.py on 33^3
mesh points as supplied matrices; the context's listed-matrix counter (tbk_ctx_solver_stats) after each call; bisection down to
single matrices; their spectra (smallest gaps) and the twisted-factorisation residuals recomputed in numpy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
g = np.arange(33) / 32.0
k = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
H = np.array([m._gen_ham(kk) for kk in k[:12000]])
def solve(h):
    ev = np.zeros((n, len(h))); vec = np.zeros((n, len(h), n), dtype=complex)
    hc = np.ascontiguousarray(h)
    _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), len(h), _lib.dptr(ev), _lib.dptr(vec.view(float))))
    return ev, vec
def listed(h):
    ctx.solver_stats(reset=True)
    solve(h)
    return ctx.solver_stats(reset=True)["listed_matrices"]
with _lib.knob("TBK_QLW_MIN", 0):
    tot = listed(H)
    print("n = %d: %d of %d matrices listed" % (n, tot, len(H)))
    with _lib.knob("TBK_TW16_GAPTOL", "0"):
        print("with no gap flags: %d" % listed(H))
    # bisect to individual matrices (a matrix is listed or not on its own)
    found = []
    stack = [(0, len(H))]
    while stack and len(found) < 6:
        a, b = stack.pop()
        if listed(H[a:b]) == 0:
            continue
        if b - a == 1:
            found.append(a)
            continue
        mid = (a + b) // 2
        stack.append((mid, b)); stack.append((a, mid))
for i in found:
    ev = np.linalg.eigvalsh(H[i])
    gaps = np.diff(ev)
    print("matrix %d: |H| = %.3f, smallest gaps %s (relative %s)" % (i, np.abs(ev).max(), np.sort(gaps)[:3], np.sort(gaps)[:3] / np.abs(ev).max()))
