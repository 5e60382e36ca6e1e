#!/usr/bin/env python3
"""n = 16 (and 12) on SMALL batches (k paths, a few hundred points): the direct solver (TBK_QL16_MIN=0) against the
workgroup-per-matrix Jacobi it leaves them to by default; kernel times from HIP events."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
ctx = _lib.default_context()
lib = _lib.lib
rng = np.random.default_rng(1)
for n in (int(a) for a in (sys.argv[1:] or ["16", "12"])):
    for nk in (8, 32, 128, 512, 2048):
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = np.ascontiguousarray(h + h.conj().transpose(0, 2, 1))
        ref = np.linalg.eigvalsh(h).T
        line = "n %2d nk %5d:" % (n, nk)
        for vec in (False, True):
            for knob in (0, None):
                with _lib.knob("TBK_QL16_MIN", knob):
                    ev = np.zeros((n, nk)); V = np.zeros((n, nk, n), dtype=complex)
                    args = (ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(V.view(float)) if vec else None)
                    _lib.check(lib.tbk_eigh_batch(*args))
                    ctx.prof_enable(1); ctx.prof_reset()
                    for _ in range(3): _lib.check(lib.tbk_eigh_batch(*args))
                    ctx.prof_enable(0)
                    t = ctx.prof_report()["eigh_batch"]; ms = t["total_ms"] / t["launches"]
                    line += "  %s %s %.3f ms (err %.0e)" % ("vec" if vec else "val", "direct" if knob == 0 else "default", ms, np.abs(ev - ref).max())
        print(line)
