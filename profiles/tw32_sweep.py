"""Round 6: eigenvectors of 17..32 states -- k_tw32_vectors (twisted factorisation + Newton-Schulz + back-transformation on the matrix
cores) against the replay of the QL rotations (TBK_TW32=0).  33^3 mesh points of a random model, whole call and per-kernel times (ms),
and the parity of the two against numpy on a sample."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
sizes = [int(a) for a in sys.argv[1:]] or [17, 18, 20, 22, 24, 25, 28, 32]
for n in sizes:
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
    w = tb.wf_array(m, [33] * 3)
    out = []
    for tw in (1, 0):
        with _lib.knob("TBK_TW32", tw):
            w.solve_on_grid([0.0, 0.0, 0.0]); ctx.sync()
            ctx.prof_enable(1); ctx.prof_reset(); w.solve_on_grid([0.0, 0.0, 0.0]); rep = ctx.prof_report(); ctx.prof_enable(0)
            st = ctx.solver_stats() if hasattr(ctx, "solver_stats") else None
            k = np.random.default_rng(n).uniform(-0.5, 0.5, (4000, 3))
            ev, vec = m.solve_all(k, eig_vectors=True)
        H = np.array([m._gen_ham(kk) for kk in k[:200]]) if hasattr(m, "_gen_ham") else None
        V = vec.transpose(1, 0, 2)[:200]
        res = orth = float("nan")
        if H is not None:
            res = np.abs(np.einsum("kij,kbj->kbi", H, V) - V * ev.T[:200, :, None]).max()
            orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).max()
        out.append((round(rep["solve_grid"]["total_ms"], 3), "%.1e" % res, "%.1e" % orth))
    print(n, "tw32", out[0], "replay", out[1], flush=True)
