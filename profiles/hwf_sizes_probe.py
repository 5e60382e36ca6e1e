import sys, os, json, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for nl, nocc in ((16, 8), (16, 12), (16, 16), (32, 16), (32, 32)):
    rng = np.random.default_rng(nl)
    m = hp.quiet(tb.tb_model, 2, 3, np.identity(3), [[0.0, 0.0, float(i)] for i in range(nl)], per=[0, 1])
    m.set_onsite(list(0.1 * rng.standard_normal(nl)))
    for i in range(nl):
        m.set_hop(-1.0, i, i, [1, 0, 0]); m.set_hop(-1.0, i, i, [0, 1, 0])
        if i + 1 < nl: m.set_hop(-0.7 + 0.1j, i, i + 1, [0, 0, 0])
    w = tb.wf_array(m, [257, 257])
    w.solve_on_grid([0.0, 0.0])
    w.position_hwf_mesh(list(range(nocc)), 2)
    ctx.prof_enable(1); ctx.prof_reset()
    t0 = time.perf_counter(); w.position_hwf_mesh(list(range(nocc)), 2); t = time.perf_counter() - t0
    r = ctx.prof_report(); ctx.prof_enable(0)
    print(json.dumps({"layers": nl, "nocc": nocc, "call_ms": round(1e3 * t, 3), "kernels_ms": {k: round(v["total_ms"], 3) for k, v in r.items()}}))
