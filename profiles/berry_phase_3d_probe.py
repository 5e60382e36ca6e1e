import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for mesh in ([1449, 1449], [129, 129, 129], [513, 65, 65], [65, 65, 513]):
    d = len(mesh)
    m = hp.random_model(tb.tb_model, 2, d, 1, seed=5 + d, nhop=6, rmax=1)
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.0] * d)
    for dr in range(d):
        for _ in range(3):
            w.berry_phase([0], dr, contin=False)
        ctx.prof_enable(1); ctx.prof_reset()
        w.berry_phase([0], dr, contin=False)
        r = ctx.prof_report(); ctx.prof_enable(0)
        print(json.dumps({"mesh": mesh, "dir": dr, "kernels_us": {k: round(1e3 * v["total_ms"], 1) for k, v in r.items()}}))
