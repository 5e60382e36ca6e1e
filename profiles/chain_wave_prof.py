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
with _lib.knob("TBK_CHAIN_WAVE_FROM", 1):
    for nocc in (1, 2, 4, 8):
        w.berry_phase(range(nocc), 2, contin=False); ctx.sync()
        ctx.prof_enable(1); ctx.prof_reset()
        w.berry_phase(range(nocc), 2, contin=False)
        ctx.prof_enable(0)
        print(nocc, {k: (v["launches"], round(v["total_ms"], 3)) for k, v in ctx.prof_report().items()})
