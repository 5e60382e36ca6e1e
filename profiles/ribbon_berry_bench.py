import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb, helpers as hp
def wall(fn, reps=3):
    fn(); best=1e9
    for _ in range(reps):
        t0=time.perf_counter(); fn(); best=min(best,time.perf_counter()-t0)
    return best
for width, nk in ((20, 101), (70, 101), (150, 101)):
    rib = hp.haldane(tb.tb_model, 1.2).cut_piece(width, 1)
    n = rib._nsta
    w = tb.wf_array(rib, [nk])
    t_s = wall(lambda: w.solve_on_grid([0.0]))
    occ = list(range(n // 2))
    t_b = wall(lambda: w.berry_phase(occ, contin=False))
    print("ribbon n=%d nocc=%d, %d k: solve_on_grid %.1f ms, berry_phase %.2f ms (value %.6f)" % (n, n // 2, nk, t_s*1e3, t_b*1e3, w.berry_phase(occ, contin=False)))
slab = hp.haldane(tb.tb_model, 1.2).cut_piece(20, 1)
