"""Round 6: the default path of 17..32 states at a size where two chunks are in flight by default (65^3 = 274 625 points, 24 and 29
states): residual and orthonormality at sampled points against H(k) from _gen_ham, the minimal gaps against a second solve on one
stream, and three shard windows against the whole array (bit for bit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib, shard
import helpers as hp
ctx = _lib.default_context()
for n in (24, 29):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=11, nhop=5 * n, rmax=1)
    mesh, start = [65, 65, 65], [0.0, 0.0, 0.0]
    w = tb.wf_array(m, mesh)
    ctx.sync(); t = time.perf_counter(); gaps = w.solve_on_grid(start); ctx.sync(); ms = (time.perf_counter() - t) * 1e3
    host = w.to_host()
    with _lib.knob("TBK_QLW_STREAMS", 1):
        w1 = tb.wf_array(m, mesh)
        g1 = w1.solve_on_grid(start)
    same = np.array_equal(gaps, g1) and np.array_equal(host, w1.to_host())
    rng = np.random.default_rng(n)
    worst_r = worst_o = 0.0
    for _ in range(40):
        idx = tuple(int(x) for x in rng.integers(0, 65, 3))
        k = np.array(idx) / 64.0
        H = m._gen_ham(k)
        V = host[idx]
        ev = np.linalg.eigvalsh(H)
        worst_r = max(worst_r, np.abs(H @ V.T - V.T * ev).max())
        worst_o = max(worst_o, np.abs(V.conj() @ V.T - np.eye(n)).max())
    ok = True
    for r in range(3):
        row0, nrows = shard.split_rows(mesh[0], 3, r)
        ww = tb.wf_array(m, [nrows, mesh[1], mesh[2]])
        ww.solve_on_grid_window(start, [row0, 0, 0], mesh)
        ok = ok and np.array_equal(ww.to_host(), host[row0:row0 + nrows])
    print("n = %d, 65^3 points: %.2f ms (first call), one stream bit-identical: %s, windows bit-identical: %s, residual %.1e, orthonormality %.1e, listed %d"
          % (n, ms, same, ok, worst_r, worst_o, ctx.solver_stats(reset=True)["listed_matrices"]), flush=True)
