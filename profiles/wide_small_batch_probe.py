import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pythtb_amd as tb
from pythtb_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(0)
for n, nk in ((300, 101), (512, 16), (300, 512)):
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n)); h = h + h.conj().transpose(0, 2, 1)
    hc = np.ascontiguousarray(h)
    for vec in (True, False):
        ev = np.zeros((n, nk)); v = np.zeros((n, nk, n), dtype=complex) if vec else None
        def call():
            _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev), _lib.dptr(v.view(float)) if vec else None))
        call(); ctx.sync(); ctx.prof_enable(1); ctx.prof_reset(); call(); rep = ctx.prof_report(); ctx.prof_enable(0)
        print(n, nk, "vec" if vec else "eval", {k: round(x["total_ms"], 2) for k, x in rep.items()}, "max err", float(np.max(np.abs(ev.T - np.linalg.eigvalsh(h)))), flush=True)
