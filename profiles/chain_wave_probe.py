#!/usr/bin/env python3
"""det-type berry_phase of 5..8 wide bands: wave-per-string kernel (k_chain_det_wave) against the thread-per-string one
(TBK_CHAIN_WAVE=0) on config E's 64^3 sub-mesh, all three directions; results must agree to rounding."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 65
w = tb.wf_array(m, [side] * 3)
w.solve_on_grid([0, 0, 0])
wrap = lambda d: (np.asarray(d) + np.pi) % (2 * np.pi) - np.pi
for nocc in (8, 5, 6, 7):
    for d in (2, 1, 0):
        res = {}
        for knob in (1, 0):
            with _lib.knob("TBK_CHAIN_WAVE", knob):
                w.berry_phase(range(nocc), d, contin=False)
                ctx.sync()
                t0 = time.perf_counter()
                ph = w.berry_phase(range(nocc), d, contin=False)
                res[knob] = (time.perf_counter() - t0, ph)
        if nocc == 8:
            ctx.prof_enable(1); ctx.prof_reset()
            w.berry_phase(range(nocc), d, contin=False)
            ctx.prof_enable(0)
            print("   kernels:", {k: round(v["total_ms"], 3) for k, v in ctx.prof_report().items()})
        print("nocc %d dir %d: wave %.2f ms   thread %.2f ms   max diff %.1e" % (
            nocc, d, res[1][0] * 1e3, res[0][0] * 1e3, np.abs(wrap(res[1][1] - res[0][1])).max()))
print("berry_flux(range(8)) on the three plane orientations of the same array:")
for dirs in ([0, 1], [1, 2], [2, 0]):
    res = {}
    for knob in (1, 0):
        with _lib.knob("TBK_CHAIN_WAVE", knob):
            w.berry_flux(range(8), dirs=dirs)
            ctx.sync()
            t0 = time.perf_counter()
            fl = w.berry_flux(range(8), dirs=dirs)
            res[knob] = (time.perf_counter() - t0, fl)
    print("dirs %s: wave-link path %.2f ms   thread-per-plaquette %.2f ms   max diff %.1e" % (
        dirs, res[1][0] * 1e3, res[0][0] * 1e3, np.abs(res[1][1] - res[0][1]).max()))
