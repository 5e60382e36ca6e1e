#!/usr/bin/env python3
"""Stress of the eigenVECTOR paths against their own matrices: residual |H v - lambda v| / |H| and orthonormality |V^+ V - 1| for the hard
spectra of evals16_stress.py, sizes from STRESS_SIZES (default 3..8, 9..16, 17, 24).   python profiles/evecs_stress.py [matrices = 1500]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pythtb_amd import _lib
ctx = _lib.default_context()
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(77)
sizes = [int(x) for x in os.environ.get("STRESS_SIZES", "3,4,5,6,8,9,12,16,17,24").split(",")]
for n in sizes:
    def conj_unitary(lev):
        a = rng.standard_normal((len(lev), n, n)) + 1j * rng.standard_normal((len(lev), n, n))
        q = np.linalg.qr(a)[0]
        h = (q * lev[:, None, :]) @ q.conj().transpose(0, 2, 1)
        return 0.5 * (h + h.conj().transpose(0, 2, 1))
    a = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    base = np.sort(rng.standard_normal((nk, (n + 1) // 2)), axis=1)
    split = 10.0 ** rng.uniform(-15, -3, size=(nk, 1))
    fam = {"random": a + a.conj().transpose(0, 2, 1),
           "pairs": conj_unitary(np.repeat(base, 2, axis=1)[:, :n] + np.tile([0.0, 1.0], (n + 1) // 2)[:n] * split),
           "graded": conj_unitary(np.sort(10.0 ** rng.uniform(-6, 6, size=(nk, n)), axis=1)),
           "rank2": conj_unitary(np.concatenate([np.ones((nk, n - 2)), 1.0 + rng.standard_normal((nk, 2))], axis=1)) if n > 2 else None}
    out = []
    for name, h in fam.items():
        if h is None: continue
        ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
        hc = np.ascontiguousarray(h)
        _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
        V = vec.transpose(1, 0, 2)
        nrm = np.abs(np.linalg.eigvalsh(h)).max(axis=1)
        res = (np.abs(np.einsum("kij,kbj->kbi", h, V) - V * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / nrm).max()
        orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).reshape(nk, -1).max(axis=1).max()
        out.append("%s res %.1e orth %.1e" % (name, res, orth))
    print("n = %2d: " % n + " | ".join(out))
