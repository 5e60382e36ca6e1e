#!/usr/bin/env python3
"""configs[4]'s solve_all leg on one GPU: cubic16 eigenvalues on a side^3 uniform mesh generated on the device (solve_all_mesh), per-kernel
brackets, against solve_on_grid (eigenvectors too) on the same number of points.   python profiles/evals16_probe.py [side = 128]"""
import contextlib, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = _lib.default_context()
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
m.solve_all_mesh([side] * 3)
ctx.prof_enable(1); ctx.prof_reset()
t0 = time.perf_counter(); ev = m.solve_all_mesh([side] * 3); t = time.perf_counter() - t0
rep = ctx.prof_report(); ctx.prof_enable(0)
kern = {k: round(v["total_ms"], 3) for k, v in rep.items()}
print(json.dumps({"side": side, "points": side ** 3, "call_ms_incl_download": t * 1e3, "kernels_ms": kern,
                  "ns_per_point_kernels": sum(v["total_ms"] for v in rep.values()) * 1e6 / side ** 3, "checksum": float(ev.sum())}))
