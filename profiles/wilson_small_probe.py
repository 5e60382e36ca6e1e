#!/usr/bin/env python3
"""Wilson-loop eigenphases, 2..20 occupied bands: per-thread chain kernels against the workgroup-level
pipeline (TBK_WILSON_BIG_FROM=2 forces the latter) over string counts and lengths."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
import helpers as hp  # noqa: E402


def timeit(w, occ, d):
    w.berry_phase(occ, d, contin=False, berry_evals=True)
    t0 = time.perf_counter()
    for _ in range(3):
        r = w.berry_phase(occ, d, contin=False, berry_evals=True)
    return (time.perf_counter() - t0) / 3, r


for nc in (2, 3, 4, 6, 8, 12, 16):
    m = hp.random_model(tb.tb_model, 2 * nc, 2, 1, 7 + nc)
    for mesh in ([101, 2], [33, 33], [17, 257], [129, 129]):
        w = tb.wf_array(m, mesh)
        w.solve_on_grid([0.0, 0.0])
        occ = list(range(nc))
        os.environ.pop("TBK_WILSON_BIG_FROM", None)
        tb._lib.lib.tbk_knobs_reload()
        ta, ra = timeit(w, occ, 0)
        os.environ["TBK_WILSON_BIG_FROM"] = "2"
        tb._lib.lib.tbk_knobs_reload()
        tb_, rb = timeit(w, occ, 0)
        d = np.abs(np.sort(ra, -1) - np.sort(rb, -1))
        print("nocc %2d  strings %4d x links %4d   per-thread %8.2f ms   workgroup %8.2f ms   diff %.1e" % (nc, mesh[1], mesh[0] - 1, ta * 1e3, tb_ * 1e3, np.minimum(d, 2 * np.pi - d).max()))
