import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import pythtb_amd as tb
from pythtb_amd import _lib
import test_regimes as tr
ctx=_lib.default_context()
for n in (5,6,7,8):
    h=tr._special_hermitian(n, np.random.default_rng(700+n)); nk=len(h)
    ref=np.linalg.eigvalsh(h); scale=np.maximum(np.abs(ref).max(axis=1),1e-300)
    for knob in (1,0):
        with _lib.knob("TBK_REG_DIRECT", knob):
            ev=np.zeros((n,nk)); vec=np.zeros((n,nk,n),dtype=complex); hc=np.ascontiguousarray(h)
            _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle,n,_lib.dptr(hc.view(float)),nk,_lib.dptr(ev),_lib.dptr(vec.view(float))))
        V=vec.transpose(1,0,2)
        res=np.abs(np.einsum("kij,kbj->kbi",h,V)-V*ev.T[:,:,None]).reshape(nk,-1).max(axis=1)/scale
        orth=np.abs(np.einsum("kbi,kci->kbc",V.conj(),V)-np.eye(n)).reshape(nk,-1).max(axis=1)
        eve=(np.abs(ev.T-ref)/scale[:,None]).max(axis=1)
        bad=[i for i in range(nk) if res[i]>1e-14 or orth[i]>1e-14 or eve[i]>4e-15]
        print(n,'direct' if knob else 'jacobi','bad idx',bad,[ (float(res[i]),float(orth[i]),float(eve[i])) for i in bad], 'nk',nk)
