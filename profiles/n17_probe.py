#!/usr/bin/env python3
"""Regime probe: kernel brackets of solve_on_grid on a 33^3 mesh for 16, 17, 24 and 32 states (the step between the fused 9..16 kernel
and the wavefront kernels above it: 0.24 -> 1.4 ms).   python profiles/n17_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n in (16, 17, 24, 32):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5, nhop=6 * n, rmax=1)
    w = tb.wf_array(m, [33] * 3)
    w.solve_on_grid([0.0, 0.0, 0.0]); ctx.sync()
    ctx.prof_enable(1); ctx.prof_reset(); w.solve_on_grid([0.0, 0.0, 0.0]); rep = ctx.prof_report(); ctx.prof_enable(0)
    print(n, {k: round(v["total_ms"], 3) for k, v in rep.items()})
