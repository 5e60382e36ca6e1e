#!/usr/bin/env python3
"""Wall-clock of the Python-level calls of config C / D (solve_on_grid, berry_flux, berry_phase) against the
kernel times, and where the Python side spends it."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
import helpers as hp  # noqa: E402


def wall(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


m = hp.haldane(tb.tb_model, 0.0)
w = tb.wf_array(m, [2049, 2049])
print("C solve_on_grid          %.3f ms" % wall(lambda: w.solve_on_grid([-0.5, -0.5])))
print("C berry_flux([0])        %.3f ms" % wall(lambda: w.berry_flux([0])))
print("C berry_flux individual  %.3f ms" % wall(lambda: w.berry_flux([0], individual_phases=True), 3))
print("C berry_phase([0], 1)    %.3f ms" % wall(lambda: w.berry_phase([0], 1)))
km = hp.kane_mele(tb.tb_model) if hasattr(hp, "kane_mele") else None
if km is not None:
    wd = tb.wf_array(km, [4097, 513])
    print("D solve_on_grid          %.3f ms" % wall(lambda: wd.solve_on_grid([-0.5, -0.5]), 5))
    print("D berry_flux([0,1])      %.3f ms" % wall(lambda: wd.berry_flux([0, 1]), 5))
    print("D berry_phase evals      %.3f ms" % wall(lambda: wd.berry_phase([0, 1], 0, contin=False, berry_evals=True), 5))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    w.solve_on_grid([-0.5, -0.5])
    w.berry_flux([0])
    w.berry_phase([0], 1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
