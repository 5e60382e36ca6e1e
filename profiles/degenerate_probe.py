"""Round 6: the 17..32-state path on a spectrum the twisted-factorisation vectors cannot serve -- every level doubly degenerate at
every k (two decoupled identical copies of a random 12-orbital model, orbitals interleaved: the situation of spin-degenerate and
Kramers-paired bands).  Default (the library notices the pairs at upload and takes the rotation replay), TBK_TW32=3 (k_tw32_vectors
forced: every matrix listed) and TBK_TW32=0, with the context's count of listed matrices."""
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
rng = np.random.default_rng(3)
half = 12
lat = np.identity(3)
orb1 = rng.random((half, 3))
orb = np.repeat(orb1, 2, axis=0)                       # orbitals 2 i and 2 i + 1: the two copies of orbital i
m = hp.quiet(tb.tb_model, 3, 3, lat, orb)
ons = rng.standard_normal(half)
m.set_onsite(list(np.repeat(ons, 2)))
seen = set()
while len(seen) < 6 * half:
    i, j = int(rng.integers(half)), int(rng.integers(half))
    R = tuple(int(x) for x in rng.integers(-1, 2, size=3))
    if (i == j and R == (0, 0, 0)) or (i, j, R) in seen or (j, i, tuple(-r for r in R)) in seen:
        continue
    seen.add((i, j, R))
    amp = complex(rng.standard_normal(), rng.standard_normal())
    for c in (0, 1):
        m.set_hop(amp, 2 * i + c, 2 * j + c, list(R))
w = tb.wf_array(m, [33] * 3)
out = {}
for tw in (0, 1, 3, 0, 1):
    with _lib.knob("TBK_TW32", tw):
        out[tw] = timed(lambda: w.solve_on_grid([0.0, 0.0, 0.0]))
        ctx.solver_stats(reset=True)
        w.solve_on_grid([0.0, 0.0, 0.0]); ctx.sync()
        print("TBK_TW32=%d: %.3f ms, matrices listed for the replay: %d of %d" % (tw, out[tw], ctx.solver_stats(reset=True)["listed_matrices"], 33 ** 3))
k = rng.uniform(-0.5, 0.5, (300, 3))
ev, vec = m.solve_all(k, eig_vectors=True)
H = np.array([m._gen_ham(kk) for kk in k])
V = vec.transpose(1, 0, 2)
res = np.abs(np.einsum("kij,kbj->kbi", H, V) - V * ev.T[:, :, None]).max()
orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(2 * half)).max()
print("two decoupled copies, n = 24: residual %.1e, orthonormality %.1e (300 k-points of a list, default path)" % (res, orth))
