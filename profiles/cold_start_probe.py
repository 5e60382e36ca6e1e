import time
t0 = time.perf_counter()
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
t1 = time.perf_counter()
import pythtb_amd as tb
t2 = time.perf_counter()
import helpers as hp
m = hp.haldane(tb.tb_model)
t3 = time.perf_counter()
w = tb.wf_array(m, [31, 31])
w.solve_on_grid([0.0, 0.0])
t4 = time.perf_counter()
f = w.berry_flux([0])
t5 = time.perf_counter()
ev = m.solve_all(np.random.rand(100, 2))
t6 = time.perf_counter()
print("numpy import %.2f s | pythtb_amd import %.2f s | model %.3f s | first solve_on_grid %.3f s | first berry_flux %.3f s | first solve_all %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
