#!/usr/bin/env python3
"""Wilson-loop eigenphases of 3 and 4 bands (tbk_berry_lanes.inl) on leg P's arrays (1025 x 257, strings along axis 0) and on their
transposes (257 x 1025, strings along the fastest axis): kernel brackets per form / segment length.
    python3 profiles/wilson_lanes_probe.py [sweep]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp

ctx = _lib.default_context()
sweep = len(sys.argv) > 1


def run(ww, occ, d, **knobs):
    import contextlib
    with contextlib.ExitStack() as st:
        for k, v in knobs.items():
            st.enter_context(_lib.knob(k, v))
        ww.berry_phase(occ, d, contin=False, berry_evals=True)
        ctx.sync(); ctx.prof_enable(1); ctx.prof_reset()
        for _ in range(5):
            ww.berry_phase(occ, d, contin=False, berry_evals=True)
        rep = ctx.prof_report(); ctx.prof_enable(0)
    return {k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in rep.items()}


for nb in (3, 4):
    mw = hp.random_model(tb.tb_model, 2 * nb, 2, 1, 7 + nb)
    for mesh, d in (([1025, 257], 0), ([257, 1025], 1)):
        ww = tb.wf_array(mw, mesh)
        ww.solve_on_grid([0.0, 0.0])
        occ = list(range(nb))
        print(nb, mesh, d, "default", run(ww, occ, d), flush=True)
        if sweep:
            for seg in (2, 3, 4, 6, 8, 12, 16, 32):
                print(nb, mesh, d, "S seg", seg, run(ww, occ, d, TBK_WILSON_FORM=0, TBK_WILSON_SEG=seg), flush=True)
            print(nb, mesh, d, "L", run(ww, occ, d, TBK_WILSON_FORM=1), flush=True)
            print(nb, mesh, d, "old", run(ww, occ, d, TBK_WILSON_REG=1, TBK_WILSON_MFMA=2), flush=True)
        del ww
