#!/usr/bin/env python3
"""Regime probe: solve_on_grid of 9..16 states when the model has MANY lattice vectors (a Wannier-interpolated model of 12 or 16
functions: ~100 R) -- k_e16's assembly stages every lattice vector with a last-axis component per point (cubic16 has two).
    python profiles/e16_many_R_probe.py [n = 12] [side = 41] [rmax = 2]"""
import contextlib, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
side = int(sys.argv[2]) if len(sys.argv) > 2 else 41
rmax = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = _lib.default_context()
out = {}
for name, model in (("cubic16-like nearest neighbours", hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)),
                    ("dense, |R| <= %d" % rmax, hp.random_model(tb.tb_model, n, 3, 1, seed=6, nhop=60 * n, rmax=rmax))):
    w = tb.wf_array(model, [side] * 3)
    w.solve_on_grid([0.0, 0.0, 0.0])
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin(); w.solve_on_grid([0.0, 0.0, 0.0]); best = min(best, ctx.timer_end())
    nR = len({tuple(int(x) for x in R) for R in np.asarray(model._hoppings, dtype=object)[:, 3]}) if hasattr(model, "_hoppings") and len(model._hoppings) else -1
    out[name] = {"ms": best, "ns_per_point": best * 1e6 / side ** 3, "lattice_vectors_one_sign": nR}
print(json.dumps({"n": n, "side": side, **out}, indent=1))

# the same two models on a k LIST (the mesh's points as a list): eigenvalues only and with eigenvectors
for name, model in (("nearest neighbours", hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)),
                    ("dense", hp.random_model(tb.tb_model, n, 3, 1, seed=6, nhop=60 * n, rmax=rmax))):
    k = model.k_uniform_mesh([side] * 3)
    for vec in (False, True):
        model.solve_all(k, eig_vectors=vec)
        t0 = time.perf_counter(); model.solve_all(k, eig_vectors=vec); t = time.perf_counter() - t0
        ctx.prof_enable(1); ctx.prof_reset(); model.solve_all(k, eig_vectors=vec); rep = ctx.prof_report(); ctx.prof_enable(0)
        kern = sum(v["total_ms"] for v in rep.values())
        print("list %-20s vectors=%d: kernels %.3f ms = %.1f ns per point (call %.1f ms)" % (name, vec, kern, kern * 1e6 / len(k), t * 1e3))
