#!/usr/bin/env python3
"""A/B of two builds of the library on the fused step, alternating in one session (box-to-box spread is +-10 %, so only
numbers of the same call compare):  python profiles/ab_fused.py pythtb_amd/libtbk_prev.so pythtb_amd/libtbk.so [env=val ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
import pythtb_amd as tb
from pythtb_amd import _lib
import bench
lib, ctx = _lib.lib, _lib.default_context()
out = {}
for side in (2049, 4097):
    g = bench.Grid(lib, _lib, ctx, bench.haldane(tb), [side, side])
    occ = np.array([0], dtype=np.int32); start = [-0.5, -0.5]
    n = 3000 if side < 3000 else 600
    for _ in range(n // 3): g.solve_flux(start, occ)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): g.solve_flux(start, occ)
    ctx.sync(); out["us_%%d" %% side] = round((time.perf_counter() - t0) / n * 1e6, 2)
    out["chern_%%d" %% side] = float(g.flux_total()[0] / (2 * np.pi))
    g.free()
print(json.dumps(out))
''' % ROOT
libs = [a for a in sys.argv[1:] if "=" not in a]
extra = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
for rep in range(3):
    for lib in libs:
        env = dict(os.environ, TBK_LIBRARY=os.path.abspath(lib), **extra)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
