import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
m = hp.random_model(tb.tb_model, 16, 3, 1, seed=21, nhop=64, rmax=1)
w = tb.wf_array(m, [129] * 3)
w.solve_on_grid([0.0, 0.0, 0.0])
for nocc in (1, 2, 3, 4, 8):
    occ = list(range(nocc))
    row = {"nocc": nocc}
    for d in (0, 2):
        for ev in (False, True):
            if nocc == 1 and ev:
                continue
            w.berry_phase(occ, d, contin=False, berry_evals=ev)
            ctx.timer_begin(); w.berry_phase(occ, d, contin=False, berry_evals=ev); t = ctx.timer_end()
            row["dir%d_%s_ms" % (d, "wilson" if ev else "det")] = round(t, 3)
    print(json.dumps(row))
