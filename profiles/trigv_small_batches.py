#!/usr/bin/env python3
"""Batches of n = 17..64 with eigenvectors that are too small for the three-kernel QL path (< 8 matrices per CU): the
workgroup Jacobi kernels (default until now) against the direct path of n >= 65 (TBK_TRIGV_FROM=17).  ms per call (outer bracket)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pythtb_amd import _lib
lib, ctx = _lib.lib, _lib.default_context()
rng = np.random.default_rng(3)
for n in (17, 20, 24, 28, 32, 40, 48, 56, 64):
    row = {"n": n}
    for nk in (16, 64, 256, 1024, 2000):
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = np.ascontiguousarray(h + h.conj().transpose(0, 2, 1))
        ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
        t = {}
        for name, frm in (("jac", -1), ("dir", 17)):
            with _lib.knob("TBK_TRIGV_FROM", frm):
                def call():
                    _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
                call()
                ctx.prof_enable(1); ctx.prof_reset()
                call(); call(); call()
                ctx.sync(); ctx.prof_enable(0)
                t[name] = max(v["total_ms"] for v in ctx.prof_report().values()) / 3
        row["x%d" % nk] = "%.3f|%.3f" % (t["jac"], t["dir"])
    print(json.dumps(row), flush=True)
