import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for n, occ, side in ((8, [0, 1, 2, 3], 65), (8, [0, 1, 2], 65), (6, [0, 1, 2], 129), (16, list(range(8)), 65), (16, list(range(5)), 65), (16, list(range(8)), 129)):
    m = hp.random_model(tb.tb_model, n, 3, 1, seed=5 + n, nhop=4 * n, rmax=1)
    w = tb.wf_array(m, [side] * 3)
    w.solve_on_grid([0.0, 0.0, 0.0])
    w.berry_phase(occ, 2, contin=False, berry_evals=True)
    ctx.prof_enable(1); ctx.prof_reset()
    ctx.timer_begin(); w.berry_phase(occ, 2, contin=False, berry_evals=True); t = ctx.timer_end()
    r = ctx.prof_report(); ctx.prof_enable(0)
    print(json.dumps({"n": n, "nocc": len(occ), "side": side, "call_ms": round(t, 3), "kernels": {k: [v["launches"], round(v["total_ms"], 3)] for k, v in r.items()}}))
