#!/usr/bin/env python3
"""solve_on_grid + berry_flux per point on 1-D .. 4-D arrays of about 2 M points (2 and 4 states).   python profiles/mesh_dims_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (2, 4):
    for mesh in ([2097153], [1449, 1449], [129, 129, 129], [38, 38, 38, 38]):
        d = len(mesh)
        m = hp.random_model(tb.tb_model, n, d, 1, seed=3 + n + d, nhop=3 * n, rmax=1)
        w = tb.wf_array(m, mesh)
        st = [0.0] * d
        w.solve_on_grid(st)
        ctx.timer_begin(); w.solve_on_grid(st); t1 = ctx.timer_end()
        row = {"states": n, "mesh": mesh, "solve_ns_per_point": round(1e6 * t1 / np.prod(mesh), 4)}
        if d >= 2:
            w.berry_flux([0])
            ctx.timer_begin(); w.berry_flux([0]); t2 = ctx.timer_end()
            row["flux01_ns_per_point"] = round(1e6 * t2 / np.prod(mesh), 4)
        if d <= 3:                     # (berry_phase of a 4-D array raises, like the reference)
            w.berry_phase([0], d - 1, contin=False)
            t3 = 1e9
            for _ in range(3):
                ctx.timer_begin(); w.berry_phase([0], d - 1, contin=False); t3 = min(t3, ctx.timer_end())
            row["phase_last_ns_per_point"] = round(1e6 * t3 / np.prod(mesh), 4)
        print(json.dumps(row))
