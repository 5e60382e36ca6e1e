#!/usr/bin/env python3
"""A/B of two builds of the library over the eigen-solver regimes (supplied matrices, device-resident, sum of the HIP-event brackets
of the call's kernels): python profiles/ab_solvers.py libA.so libB.so.   Prints ms per call for every (n, count, vectors) and the ratio B / A."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
from pythtb_amd import _lib
import ctypes as C
lib, ctx = _lib.lib, _lib.default_context()
cases = [(3, 262144), (4, 262144), (6, 131072), (8, 131072), (12, 65536), (16, 65536), (16, 1000), (24, 16384), (32, 16384), (32, 500),
         (48, 8192), (64, 8192), (64, 100), (100, 512), (128, 512), (128, 32), (300, 101), (512, 8), (800, 2)]
rng = np.random.default_rng(5)
out = {}
for n, nk in cases:
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = np.ascontiguousarray(h + h.conj().transpose(0, 2, 1))
    ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
    for wv in (0, 1):
        def call():
            _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float)) if wv else None))
        call()
        ctx.prof_enable(1); ctx.prof_reset()
        reps = 2 if n >= 100 else 3
        for _ in range(reps): call()
        ctx.sync(); ctx.prof_enable(0)
        rep = ctx.prof_report()
        out["n%%d_x%%d_%%s" %% (n, nk, "vec" if wv else "val")] = round(sum(r["total_ms"] for r in rep.values()) / reps, 4)
print(json.dumps(out))
''' % ROOT
libs = sys.argv[1:3]
res = []
for lib in libs:
    env = dict(os.environ, TBK_LIBRARY=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    if not r.stdout.strip():
        print(lib, "failed:", r.stderr[-800:]); sys.exit(1)
    res.append(json.loads(r.stdout.strip().splitlines()[-1]))
for k in res[0]:
    a, b = res[0][k], res[1].get(k)
    print("%-22s %10.4f %10.4f  %5.2f" % (k, a, b, b / a if a else 0))
