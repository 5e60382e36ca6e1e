#!/usr/bin/env python3
"""How the fused n <= 16 kernel copes with bands that are degenerate on the WHOLE mesh (a spinful model without spin-orbit
coupling: every level twice): time of solve_on_grid on a 513^2 array and the share of the fallback kernels, beside a generic
16-state model of the same range.    python profiles/e16_degenerate_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()


def model(nspin, norb, seed):
    rng = np.random.default_rng(seed)
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((norb, 2)), nspin=nspin)
    m.set_onsite(list(rng.standard_normal(norb)))
    for i in range(norb):
        for j in range(i + 1, norb):
            m.set_hop(0.3 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0])
    for R in ([1, 0], [0, 1], [1, 1]):
        for i in range(norb):
            for j in range(norb):
                if rng.random() < 0.4:
                    m.set_hop(0.2 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    return m


for name, m in (("spin-degenerate 8 orbitals x 2 spins", model(2, 8, 17)), ("generic 16 orbitals", model(1, 16, 17))):
    w = tb.wf_array(m, [513, 513])
    w.solve_on_grid([0.05, -0.1])
    best = None
    for _ in range(3):
        ctx.prof_enable(1); ctx.prof_reset()
        ctx.timer_begin(); w.solve_on_grid([0.05, -0.1]); t = ctx.timer_end()
        r = ctx.prof_report(); ctx.prof_enable(0)
        if best is None or t < best[0]:
            best = (t, r)
    print(json.dumps({"model": name, "points": 513 * 513, "call_ms": round(best[0], 3),
                      "kernels_ms": {k: round(v["total_ms"], 3) for k, v in best[1].items()}}))
