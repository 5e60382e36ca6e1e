#!/usr/bin/env python3
"""berry_flux and det-type berry_phase, 3..16 occupied bands: thread-per-plaquette / per-segment kernels against
the workgroup-per-link LU path (TBK_DET_BIG_FROM=2 forces the latter)."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
import helpers as hp  # noqa: E402


def timeit(fn):
    fn()
    t0 = time.perf_counter()
    for _ in range(3):
        r = fn()
    return (time.perf_counter() - t0) / 3, r


for nc in (3, 4, 6, 8, 9, 12, 16):
    m = hp.random_model(tb.tb_model, 2 * nc, 2, 1, 7 + nc)
    for mesh in ([33, 33], [129, 129], [513, 513]):
        if nc >= 12 and mesh[0] > 200:
            continue
        w = tb.wf_array(m, mesh)
        w.solve_on_grid([0.0, 0.0])
        occ = list(range(nc))
        res = []
        for knob in (None, "2"):
            os.environ.pop("TBK_DET_BIG_FROM", None)
            if knob:
                os.environ["TBK_DET_BIG_FROM"] = knob
            tb._lib.lib.tbk_knobs_reload()
            tf, f = timeit(lambda: w.berry_flux(occ, individual_phases=True))
            tp, ph = timeit(lambda: w.berry_phase(occ, 0, contin=False))
            res.append((tf, tp, f, ph))
        os.environ.pop("TBK_DET_BIG_FROM", None)
        df = np.abs((res[0][2] - res[1][2] + np.pi) % (2 * np.pi) - np.pi).max()
        print("nocc %2d  mesh %3d^2   flux: per-thread %7.2f ms  per-link LU %7.2f ms    phase: %7.2f / %7.2f ms   diff %.1e" % (nc, mesh[0], res[0][0] * 1e3, res[1][0] * 1e3, res[0][1] * 1e3, res[1][1] * 1e3, df))
