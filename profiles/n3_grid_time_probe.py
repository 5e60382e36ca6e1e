"""k_grid_rows<3,PM> on a 3-band model (2049 x 1025): HIP-event bracket per launch."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pythtb_amd as tb, helpers as hp
from pythtb_amd import _lib
ctx = _lib.default_context()
m = hp.random_model(tb.tb_model, 3, 2, 1, seed=43, nhop=12, rmax=1)
w = tb.wf_array(m, [2049, 1025])
g = w.solve_on_grid([0.1, 0.2])
ctx.prof_enable(1); ctx.prof_reset()
for _ in range(20):
    w.solve_on_grid([0.1, 0.2])
rep = ctx.prof_report(); ctx.prof_enable(0)
print('gaps', g, {k: round(v['total_ms'] / v['launches'], 4) for k, v in rep.items()})
