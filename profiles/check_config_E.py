import sys, os, contextlib, io
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pythtb_amd as tb
import helpers as hp
from oracle import tb_oracle as orc
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
mesh=[9,11,65]; start=[0.0,0.0,0.0]
w=tb.wf_array(m, mesh); gaps=w.solve_on_grid(start)
owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
print("gaps diff", np.max(np.abs(gaps-ogaps)))
wrap=lambda d:(np.asarray(d)+np.pi)%(2*np.pi)-np.pi
for occ in (list(range(8)), [0,1,2,3,4], list(range(12))):
    for d in (2,0):
        got=w.berry_phase(occ, d, contin=False)
        ref=orc.berry_phase(owfs, 3, occ, d, contin=False)
        print(len(occ), d, "phase err", np.max(np.abs(wrap(got-ref))))
    got=w.berry_flux(occ, dirs=[0,1])
    ref=orc.berry_flux(owfs,3,occ,[0,1],vectorised=True)
    print(len(occ), "flux err", np.max(np.abs(got-ref)))
# orthonormality of device eigenvectors
h=w._wfs.reshape(-1,16,16)
print("orth err", max(np.max(np.abs(v.conj()@v.T-np.eye(16))) for v in h[::7]))
