#!/usr/bin/env python3
"""solve_on_grid of a 2-state and a 4-state model on a 1025^2 array against the range of its hoppings along the LAST mesh axis
(k_grid_rows<N, PM> is compiled for ranges 0..2, a generic-range instance and the per-point kernel take the rest).
    python profiles/hop_range_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for norb in (2, 4):
    for rng_last in (0, 1, 2, 3, 4, 6, 10, 20):
        rng = np.random.default_rng(7)
        m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((norb, 2)))
        m.set_onsite(list(rng.standard_normal(norb)))
        for i in range(norb):
            for j in range(norb):
                m.set_hop(0.3 + 0.1j * (i + 1), i, j, [1, 0], mode="add", allow_conjugate_pair=True)
                if rng_last > 0:
                    m.set_hop(0.2 - 0.1j * (j + 1), i, j, [0, rng_last], mode="add", allow_conjugate_pair=True)
        w = tb.wf_array(m, [1025, 1025])
        w.solve_on_grid([0.0, 0.0])
        ctx.prof_enable(1); ctx.prof_reset()
        w.solve_on_grid([0.0, 0.0])
        r = ctx.prof_report(); ctx.prof_enable(0)
        print(json.dumps({"states": norb, "range_along_last_axis": rng_last, "solve_ms": round(r["solve_grid"]["total_ms"], 4)}))
