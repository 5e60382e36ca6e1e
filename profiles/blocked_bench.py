#!/usr/bin/env python3
"""Wide-matrix batches (Haldane ribbons on a k-path): default solver dispatch against the block-Jacobi solver
(TBK_BLOCKED=1), eigenvalues only and with eigenvectors; errors against numpy.linalg.eigh."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
import helpers as hp  # noqa: E402
from oracle import tb_oracle as orc  # noqa: E402

cases = [(35, 256), (50, 128), (64, 512), (100, 101), (150, 101), (150, 16), (400, 8), (400, 1)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for width, nk in cases:
    rib = hp.haldane(tb.tb_model, 0.3).cut_piece(width, 1)
    n = 2 * width
    k = np.linspace(0.0, 1.0, nk, endpoint=False)[:, None] + 0.013
    ham = orc.ham_batch(rib, k)
    ref = np.linalg.eigvalsh(ham).T
    line = "n %4d x %4d k:" % (n, nk)
    for knob in ("0", "1"):
        os.environ["TBK_BLOCKED"] = knob
        tb._lib.lib.tbk_knobs_reload()
        for vec in (False, True):
            rib.solve_all(k[:2], eig_vectors=vec)
            t0 = time.perf_counter()
            out = rib.solve_all(k, eig_vectors=vec)
            t = time.perf_counter() - t0
            ev = out[0] if vec else out
            err = np.abs(ev - ref).max()
            res = 0.0
            if vec:
                V = out[1]              # [band][k][orb]
                for i in range(0, nk, max(1, nk // 4)):
                    res = max(res, np.abs(ham[i] @ V[:, i].T - V[:, i].T * ev[:, i]).max())
                    res = max(res, np.abs(V[:, i].conj() @ V[:, i].T - np.eye(n)).max())
            line += "  %s%s %8.1f ms (err %.0e%s)" % ("blocked" if knob == "1" else "default", " +vec" if vec else "     ", t * 1e3, err, ", res %.0e" % res if vec else "")
    print(line)
os.environ.pop("TBK_BLOCKED", None)
tb._lib.lib.tbk_knobs_reload()
