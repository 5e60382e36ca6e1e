import sys, os, json, cProfile, pstats, io, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
m = hp.random_model(tb.tb_model, 2, 3, 1, seed=8, nhop=6, rmax=1)
w = tb.wf_array(m, [129, 129, 129])
w.solve_on_grid([0.0] * 3)
for _ in range(3): w.berry_phase([0], 2, contin=False)
t0 = time.perf_counter()
for _ in range(20): w.berry_phase([0], 2, contin=False)
print("per call us", 1e6 * (time.perf_counter() - t0) / 20)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): w.berry_phase([0], 2, contin=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:1800])
