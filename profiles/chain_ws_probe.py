#!/usr/bin/env python3
"""berry_phase(range(8), dir) of cubic16 on a side^3 mesh against the size of the link-matrix workspace (TBK_CHAIN_WS_MB): a
workspace that stays in the 256 MiB last-level cache is written and read back without touching HBM."""
import contextlib, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 129
w = tb.wf_array(m, [side] * 3)
w.solve_on_grid([0, 0, 0])
for mb in (1024, 256, 128, 64, 32, 16, 1024):
    with _lib.knob("TBK_CHAIN_WS_MB", mb):
        res = {"ws_mb": mb}
        for d in (2, 0):
            w.berry_phase(range(8), d, contin=False); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3): ph = w.berry_phase(range(8), d, contin=False)
            res["dir%d_call_ms" % d] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            ctx.prof_enable(1); ctx.prof_reset()
            w.berry_phase(range(8), d, contin=False)
            ctx.prof_enable(0)
            rep = ctx.prof_report()
            res["dir%d_links_ms" % d] = round(rep["chain_links"]["total_ms"], 3); res["dir%d_lu_ms" % d] = round(rep["chain_lu"]["total_ms"], 3)
            res["dir%d_launches" % d] = rep["chain_links"]["launches"]; res["dir%d_sum" % d] = float(np.sum(ph))
        print(json.dumps(res), flush=True)
