import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
import importlib
tr = importlib.import_module("test_regimes")
m = tr._chain(tb, 2, 0.3)
k = np.random.default_rng(2 * 1000 + 255).uniform(-1.0, 2.0, 255)
res = {}
for poll in (0, 1):
    for kpt in (1, 2):
        with _lib.knob("TBK_POLL_DONE", poll), _lib.knob("TBK_SMALL_KPT", kpt):
            outs = [m.solve_all(k, eig_vectors=True) for _ in range(20)]
        same = all(np.array_equal(outs[0][1], o[1]) for o in outs)
        res[(poll, kpt)] = outs[0]
        print("poll", poll, "kpt", kpt, "20 repeats identical:", same)
for a in res:
    for b in res:
        if a < b:
            d = res[a][1] != res[b][1]
            print(a, b, "vec mismatches:", int(d.sum()), "max abs diff", float(np.abs(res[a][1] - res[b][1]).max()), "first at", np.argwhere(d)[:3].tolist())
