#!/usr/bin/env python3
"""n = 24..64 with eigenvectors: the three-kernel QL-replay path (default) against the workgroup-scale direct path of n >= 65
(TBK_TRIGV_FROM=17), supplied matrices, outer HIP-event bracket of the solve, accuracy against LAPACK."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pythtb_amd import _lib
lib, ctx = _lib.lib, _lib.default_context()
rng = np.random.default_rng(3)
for n, nk in ((24, 16384), (32, 16384), (48, 8192), (64, 8192), (64, 1024), (40, 512)):
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = np.ascontiguousarray(h + h.conj().transpose(0, 2, 1))
    ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
    ref = np.linalg.eigvalsh(h[:64])
    res = {"n": n, "nk": nk}
    for name, frm in (("qlw", -1), ("trigv", 17)):
        with _lib.knob("TBK_TRIGV_FROM", frm):
            def call():
                _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
            call()
            ctx.prof_enable(1); ctx.prof_reset()
            call(); call()
            ctx.sync(); ctx.prof_enable(0)
            rep = ctx.prof_report()
            outer = max(v["total_ms"] for v in rep.values()) / 2
            res[name + "_ms"] = round(outer, 3)
            res[name + "_kernels"] = {k: round(v["total_ms"] / 2, 3) for k, v in rep.items()}
            V = vec.transpose(1, 0, 2)[:64]
            res[name + "_eval_err"] = float(np.max(np.abs(ev.T[:64] - ref)))
            r = np.einsum("kij,kbj->kbi", h[:64], V) - V * ev.T[:64, :, None]
            res[name + "_resid"] = float(np.abs(r).max())
            res[name + "_orth"] = float(np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).max())
    print(json.dumps(res), flush=True)
