import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
w = tb.wf_array(m, [129] * 3)
w.solve_on_grid([0, 0, 0])
wrap = lambda d: (np.asarray(d) + np.pi) % (2 * np.pi) - np.pi
for nocc in (1, 2, 3, 4):
    for d in (2, 0):
        res = {}
        for frm in (1, 5):
            with _lib.knob("TBK_CHAIN_WAVE_FROM", frm):
                w.berry_phase(range(nocc), d, contin=False); ctx.sync()
                t0 = time.perf_counter(); ph = w.berry_phase(range(nocc), d, contin=False); t = time.perf_counter() - t0
                w.berry_flux(range(nocc), dirs=[0, 1]); ctx.sync()
                t0 = time.perf_counter(); fl = w.berry_flux(range(nocc), dirs=[0, 1]); tf = time.perf_counter() - t0
                res[frm] = (t, ph, tf, fl)
        print("nocc %d dir %d: phase wave %.2f ms thread %.2f ms (diff %.1e) | flux wave %.2f ms thread %.2f ms (diff %.1e)" % (
            nocc, d, res[1][0]*1e3, res[5][0]*1e3, np.abs(wrap(res[1][1]-res[5][1])).max(), res[1][2]*1e3, res[5][2]*1e3, np.abs(res[1][3]-res[5][3]).max()))
