import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pythtb_amd as tb, helpers as hp
h = hp.haldane(tb.tb_model, 0.0)
def wall(fn, reps=5):
    fn(); best=1e9
    for _ in range(reps):
        t0=time.perf_counter(); fn(); best=min(best,time.perf_counter()-t0)
    return best
for nx, ny in ((5,5),(7,7),(10,10),(14,14),(20,20)):
    m = h.cut_piece(nx, 0).cut_piece(ny, 1)
    n = m._nsta
    H = np.asarray(m._gen_ham())
    t_gpu = wall(lambda: m.solve_all(eig_vectors=True))
    t_gpu_e = wall(lambda: m.solve_all())
    t_np = wall(lambda: np.linalg.eigh(H), 3)
    print("n=%4d  gpu evec %.2f ms  eval %.2f ms   numpy eigh %.2f ms" % (n, t_gpu*1e3, t_gpu_e*1e3, t_np*1e3))
