#!/usr/bin/env python3
"""Eigenvalue-only solves, n = 9..16 on k lists: the 512-thread form (one wavefront finishes 32 matrices, one per lane)
against the replicated 256-thread form (TBK_QL16_EVONLY=0); kernel time from HIP events, error against numpy."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
from oracle import tb_oracle as orc
ctx = _lib.default_context()
rng = np.random.default_rng(0)
for n in (16, 12, 9):
    if n == 16:
        with contextlib.redirect_stdout(io.StringIO()):
            m = hp.cubic16(tb.tb_model)
    else:
        m = hp.random_model(tb.tb_model, n, 3, 1, 100 + n, nhop=3 * n * n // 2, rmax=1)
    k = rng.random((262144, 3))
    ref = orc.solve_all_vec(m, k[:2000])
    for knob in (1, 0):
        with _lib.knob("TBK_QL16_EVONLY", knob):
            ev = m.solve_all(k)
            ctx.prof_enable(1); ctx.prof_reset()
            ev = m.solve_all(k)
            ctx.prof_enable(0)
            t = {a: round(b["total_ms"], 3) for a, b in ctx.prof_report().items()}
        print("n %2d evonly-form %d: kernels %s   max err vs numpy %.1e" % (n, knob, t, np.abs(ev[:, :2000] - ref).max()))
