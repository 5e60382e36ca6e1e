"""k_flux_rows right after solve_on_grid rewrote the array (the state the bench's loop puts it in): Haldane 2048^2 and 4096^2, per-launch
HIP-event brackets of the pair, for both tile orders (TBK_FLUX_ORDER)."""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pythtb_amd as tb, helpers as hp
from pythtb_amd import _lib
ctx = _lib.default_context()
for n in (2048, 4096):
    m = hp.haldane(tb.tb_model, 0.0)
    w = tb.wf_array(m, [n + 1, n + 1])
    for order in (0, 1):
        with _lib.knob("TBK_FLUX_ORDER", order):
            w.solve_on_grid([-0.5, -0.5]); w.berry_flux([0])
            ctx.prof_enable(1); ctx.prof_reset()
            for _ in range(10):
                w.solve_on_grid([-0.5, -0.5]); f = w.berry_flux([0])
            rep = ctx.prof_report(); ctx.prof_enable(0)
        print(n, 'order', order, f / (2 * np.pi), {k: round(v['total_ms'] / v['launches'], 4) for k, v in rep.items() if k in ('solve_grid', 'berry_flux')})
