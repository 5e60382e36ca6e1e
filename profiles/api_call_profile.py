import cProfile, pstats, os, sys, io, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
import pythtb_amd as tb
m = bench.haldane(tb)
w = tb.wf_array(m, [2049, 2049])
w.solve_on_grid([-0.5, -0.5]); w.berry_flux([0])
def loop(n):
    for _ in range(n):
        w.solve_on_grid([-0.5, -0.5]); w.berry_flux([0])
t0 = time.perf_counter(); loop(300); t = (time.perf_counter() - t0) / 300
print("per pair of calls: %.1f us" % (t * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(300); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:2600])
