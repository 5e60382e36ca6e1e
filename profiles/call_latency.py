#!/usr/bin/env python3
"""Wall-clock latency of small Python-level calls (what a plotting script does in a loop)."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
import helpers as hp  # noqa: E402


def wall(fn, reps=200):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


g = hp.graphene(tb.tb_model)
k_vec, _, _ = g.k_path([[0.0, 0.0], [2.0 / 3.0, 1.0 / 3.0], [0.5, 0.5], [0.0, 0.0]], 121, report=False)
print("solve_all(121 k, graphene)            %8.1f us" % wall(lambda: g.solve_all(k_vec)))
print("solve_all(121 k, vectors)             %8.1f us" % wall(lambda: g.solve_all(k_vec, eig_vectors=True)))
print("solve_one                             %8.1f us" % wall(lambda: g.solve_one([0.1, 0.2])))
print("_gen_ham                              %8.1f us" % wall(lambda: g._gen_ham([0.1, 0.2])))
h = hp.haldane(tb.tb_model, 0.2)
w = tb.wf_array(h, [31, 31])
print("solve_on_grid 31x31                   %8.1f us" % wall(lambda: w.solve_on_grid([-0.5, -0.5])))
print("berry_phase([0], 1) 31x31             %8.1f us" % wall(lambda: w.berry_phase([0], 1)))
print("berry_flux([0]) 31x31                 %8.1f us" % wall(lambda: w.berry_flux([0])))
print("wf[i, j] read (host mirror cached)    %8.1f us" % wall(lambda: w[3, 4]))


def edit_and_solve():
    h.set_onsite([0.1, -0.1], mode="reset")
    return h.solve_one([0.3, 0.1])


print("set_onsite + solve_one (re-upload)    %8.1f us" % wall(edit_and_solve))
c3 = hp.chain3(tb.tb_model, -1.0, 2.0, 0.3)
fin = c3.cut_piece(10, 0)
print("finite chain (n=30) solve_all+vec     %8.1f us" % wall(lambda: fin.solve_all(eig_vectors=True)))

km = hp.kane_mele(tb.tb_model, "odd")
wk = tb.wf_array(km, [41, 41])
wk.solve_on_grid([-0.5, -0.5])
print("Kane-Mele 41x41 berry_phase([0,1], 1, berry_evals=True)  %8.1f us" % wall(lambda: wk.berry_phase([0, 1], 1, contin=True, berry_evals=True)))
print("Kane-Mele 41x41 berry_phase([0,1], 1)                    %8.1f us" % wall(lambda: wk.berry_phase([0, 1], 1)))
