import sys, os, time, json, contextlib, io, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
from bench_configs import grid_handle
lib=_lib.lib; ctx=_lib.default_context()
with contextlib.redirect_stdout(io.StringIO()):
    model = hp.cubic16(tb.tb_model)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 257
mesh=[N,N,N]
t0=time.perf_counter()
hw,pbc=grid_handle(_lib, lib, ctx, model, mesh)
print("alloc GB", N**3*256*16/1e9, "s", time.perf_counter()-t0, flush=True)
hm=model._device_model(); start=np.zeros(3)
for rep in range(2):
    ctx.timer_begin()
    _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0]))
    ms=ctx.timer_end()
    print("solve ms", ms, "k/s", (N-1)**3/ms*1e3, flush=True)
gaps=np.zeros(15); _lib.check(lib.tbk_wfs_solve_grid_result(hw,_lib.dptr(gaps))); print("gap78", gaps[7])
occ=np.arange(8,dtype=np.int32); ph=np.zeros(N*N)
for rep in range(2):
    t0=time.perf_counter(); _lib.check(lib.tbk_berry_phase(hw,_lib.iptr(occ),8,2,0,_lib.dptr(ph))); dt=time.perf_counter()-t0
    print("berry_phase s", dt, "links/s", N*N*(N-1)/dt, "sum", ph.sum(), flush=True)
fl=np.zeros(N); t0=time.perf_counter(); _lib.check(lib.tbk_berry_flux(hw,_lib.iptr(occ),8,0,1,_lib.dptr(fl),None)); print("flux s", time.perf_counter()-t0, fl[:3]/(2*np.pi))
