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

# the same pair as bare C-ABI calls with prebuilt arguments: what is left of the difference is Python (argument checks, numpy
# temporaries, ctypes pointer objects)
from pythtb_amd import _lib
start = np.array([-0.5, -0.5]); pbc = w._pbc_phases(); gaps = np.zeros(1); totals = np.zeros(1); occ32 = np.zeros(1, dtype=np.int32)
h, mh = w._ensure_dev(), m._device_model()
a1 = (h, mh, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, 2049, _lib.dptr(gaps))
a2 = (h, _lib.iptr(occ32), 1, 0, 1, _lib.dptr(totals), None)
f1, f2 = _lib.lib.tbk_wfs_solve_grid, _lib.lib.tbk_berry_flux
def cloop(n):
    for _ in range(n):
        f1(*a1); f2(*a2)
cloop(20)
t0 = time.perf_counter(); cloop(300); t = (time.perf_counter() - t0) / 300
print("per pair of bare C-ABI calls: %.1f us" % (t * 1e6))
