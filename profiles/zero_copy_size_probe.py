import sys, os, json, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
rng = np.random.default_rng(0)
for n in (2, 4, 8, 16, 32):
    m = hp.random_model(tb.tb_model, n, 2, 1, seed=n, nhop=4 * n, rmax=1)
    row = {"n": n}
    for nk in (500, 1000, 3000, 8000):
        k = rng.random((nk, 2))
        for vec in (False, True):
            for zc in (64, 1024, 4096):
                with _lib.knob("TBK_ZERO_COPY_KB", zc):
                    for _ in range(3): m.solve_all(k, eig_vectors=vec)
                    t0 = time.perf_counter()
                    for _ in range(20): m.solve_all(k, eig_vectors=vec)
                    row["%d%s_zc%d" % (nk, "v" if vec else "e", zc)] = round(1e6 * (time.perf_counter() - t0) / 20, 1)
    print(json.dumps(row))
