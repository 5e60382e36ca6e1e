#!/usr/bin/env python3
"""A/B of two builds on the wave-per-string link kernels: cubic16 (16 orbitals) on a side^3 mesh, berry_phase(range(nocc), dir)
for 8 / 4 / 2 bands and the three directions, kernel brackets per call.  python profiles/ab_chain.py libA.so libB.so [side]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import contextlib, io, json, sys, time
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
side = %d
w = tb.wf_array(m, [side] * 3)
w.solve_on_grid([0, 0, 0])
out = {}
for nocc in (8, 4, 2):
    for d in (2, 1, 0):
        w.berry_phase(range(nocc), d, contin=False); ctx.sync()
        ctx.prof_enable(1); ctx.prof_reset()
        ph = w.berry_phase(range(nocc), d, contin=False)
        ctx.prof_enable(0)
        rep = ctx.prof_report()
        out["nocc%%d_dir%%d" %% (nocc, d)] = {"links_ms": round(rep.get("chain_links", {"total_ms": 0})["total_ms"], 3),
                                               "lu_ms": round(rep.get("chain_lu", {"total_ms": 0})["total_ms"], 3), "sum": float(np.sum(ph))}
print(json.dumps(out))
'''
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
side = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 129
res = []
for lib in libs:
    env = dict(os.environ, TBK_LIBRARY=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT, side)], env=env, capture_output=True, text=True)
    if not r.stdout.strip():
        print(lib, "failed:", r.stderr[-800:]); sys.exit(1)
    res.append(json.loads(r.stdout.strip().splitlines()[-1]))
for k in res[0]:
    print("%-14s" % k, "  ".join("links %8.3f lu %7.3f (sum %.12f)" % (r[k]["links_ms"], r[k]["lu_ms"], r[k]["sum"]) for r in res))
