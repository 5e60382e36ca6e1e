#!/usr/bin/env python3
"""n = 5..8 with eigenvectors: the register Jacobi kernel (k_solve_reg, the dispatch's choice) on supplied matrices --
time per batch from HIP-event brackets.  (The fused n <= 16 kernel on the same sizes: profiles/microbench/e16_bench <nk> <n>.)
    python profiles/reg_vs_e16_probe.py [nk = 137312]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pythtb_amd import _lib
ctx = _lib.default_context()
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 137312
rng = np.random.default_rng(0)
for n in (5, 6, 8, 9, 12, 16):
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = np.ascontiguousarray(h + h.conj().transpose(0, 2, 1))
    ev, vec = np.zeros((n, nk)), np.zeros((n, nk, n), dtype=complex)
    for vecs in (True, False):
        best = None
        for _ in range(3):
            ctx.prof_enable(1); ctx.prof_reset()
            _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float)) if vecs else None))
            r = ctx.prof_report(); ctx.prof_enable(0)
            t = sum(v["total_ms"] for v in r.values())
            best = t if best is None else min(best, t)
        print(json.dumps({"n": n, "nk": nk, "vectors": vecs, "kernels_ms": round(best, 4), "kernels": list(r.keys())}))
