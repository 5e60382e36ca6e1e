import sys, os, json, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
for name, m, occs in (("haldane", hp.haldane(tb.tb_model), ([0], [1], [0, 1])), ("kane_mele", hp.kane_mele(tb.tb_model), ([0, 1], [2, 3], [1, 2], [0], [0, 1, 2], [0, 1, 2, 3]))):
    w = tb.wf_array(m, [1025, 1025])
    w.solve_on_grid([0.0, 0.0])
    for occ in occs:
        for ind in (False, True):
            w.berry_flux(occ, individual_phases=ind)
            t0 = time.perf_counter()
            for _ in range(5): w.berry_flux(occ, individual_phases=ind)
            t = (time.perf_counter() - t0) / 5
            print(json.dumps({"model": name, "occ": occ, "individual_phases": ind, "call_us": round(1e6 * t, 1)}))
