#!/usr/bin/env python3
"""Wilson-loop eigenphases with more than 16 occupied bands (berry_phase(..., berry_evals=True)): time per
call and error against the NumPy oracle (svd polar factors + eigvals)."""
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


def set_err(a, b):
    a, b = np.sort(np.asarray(a), -1), np.sort(np.asarray(b), -1)
    best = np.inf
    for sh in (-1, 0, 1):          # values next to +-pi may wrap differently
        d = np.abs(np.roll(a, sh, -1) - b)
        best = min(best, np.max(np.minimum(d, 2 * np.pi - d)))
    return best


cases = [("random n=20, 17 bands, 6x5", hp.random_model(tb.tb_model, 20, 2, 1, 33), [6, 5], list(range(17)), 0),
         ("random n=40, 35 bands, 21x4", hp.random_model(tb.tb_model, 40, 2, 1, 5), [21, 4], list(range(35)), 0),
         ("Haldane ribbon 70 of 140, 41 k", hp.haldane(tb.tb_model, 1.2).cut_piece(70, 1), [41], list(range(70)), None),
         ("Haldane ribbon 150 of 300, 101 k", hp.haldane(tb.tb_model, 1.2).cut_piece(150, 1), [101], list(range(150)), None)]
for name, m, mesh, occ, d in cases:
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.0] * len(mesh))
    got = w.berry_phase(occ, d, contin=False, berry_evals=True)
    t0 = time.perf_counter()
    got = w.berry_phase(occ, d, contin=False, berry_evals=True)
    t1 = time.perf_counter()
    owfs, _ = orc.solve_on_grid(m, mesh, [0.0] * len(mesh), vectorised=True)
    t2 = time.perf_counter()
    ref = orc.berry_phase(owfs, len(mesh), occ, d, contin=False, berry_evals=True)
    t3 = time.perf_counter()
    print("%-34s  device %.2f ms   numpy %.1f ms   max error %.2e" % (name, (t1 - t0) * 1e3, (t3 - t2) * 1e3, set_err(got, ref)))
