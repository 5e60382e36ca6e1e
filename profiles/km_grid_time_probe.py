import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import pythtb_amd as tb, helpers as hp
from pythtb_amd import _lib
ctx=_lib.default_context()
m=hp.kane_mele(tb.tb_model,"odd")
w=tb.wf_array(m,[4097,513])
g=w.solve_on_grid([-0.5,-0.5]); print('gaps',g)
ctx.prof_enable(1); ctx.prof_reset()
for _ in range(20): w.solve_on_grid([-0.5,-0.5])
rep=ctx.prof_report(); ctx.prof_enable(0)
print({k:round(v['total_ms']/v['launches'],4) for k,v in rep.items()})
m3=hp.chain3(tb.tb_model,-1.0,2.0,0.3) if False else None
