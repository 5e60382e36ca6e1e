#!/usr/bin/env python3
"""Stress of the eigenvalue-only fused kernel (k_e16<2, false>) against LAPACK: random, clustered, pair-split (1e-3 .. 1e-15), graded and
low-rank-perturbed-identity spectra for every size 9..16; prints the worst eigenvalue error / |T| per family.
    python profiles/evals16_stress.py [matrices per family and size = 4000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pythtb_amd import _lib
ctx = _lib.default_context()
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(2025)

def conj_unitary(lev):
    n = lev.shape[-1]
    a = rng.standard_normal((len(lev), n, n)) + 1j * rng.standard_normal((len(lev), n, n))
    q = np.linalg.qr(a)[0]
    h = (q * lev[:, None, :]) @ q.conj().transpose(0, 2, 1)
    return 0.5 * (h + h.conj().transpose(0, 2, 1))

worst = {}
for n in [int(x) for x in os.environ.get("STRESS_SIZES", "9,10,11,12,13,14,15,16").split(",")]:
    fam = {}
    a = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    fam["random"] = a + a.conj().transpose(0, 2, 1)
    lev = np.where(np.arange(n) < n // 2, -2.0, 2.0) + 0.2 * rng.standard_normal((nk, n))
    fam["two clusters"] = conj_unitary(lev)
    base = np.sort(rng.standard_normal((nk, (n + 1) // 2)), axis=1)
    split = 10.0 ** rng.uniform(-15, -3, size=(nk, 1))
    lev = np.repeat(base, 2, axis=1)[:, :n] + np.tile([0.0, 1.0], (n + 1) // 2)[:n] * split
    fam["pairs split 1e-15..1e-3"] = conj_unitary(lev)
    fam["graded 1e-6..1e6"] = conj_unitary(np.sort(10.0 ** rng.uniform(-6, 6, size=(nk, n)), axis=1))
    fam["identity + rank 2"] = conj_unitary(np.concatenate([np.ones((nk, n - 2)), 1.0 + rng.standard_normal((nk, 2))], axis=1))
    fam["triple within 1e-6"] = conj_unitary(np.sort(rng.standard_normal((nk, n)), axis=1) * np.r_[np.ones(n - 3), 0, 0, 0] +
                                             np.r_[np.zeros(n - 3), 0.5, 0.5 + 1e-6, 0.5 + 2e-6])
    for name, h in fam.items():
        ev = np.zeros((n, nk))
        hc = np.ascontiguousarray(h)
        _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev), None))
        ref = np.linalg.eigvalsh(h)
        err = (np.abs(ev.T - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
        srt = bool(np.all(np.diff(ev, axis=0) >= 0))
        worst[name] = max(worst.get(name, 0.0), err)
        if not srt or err > 2e-14:
            print("n = %d, %s: error %.3g, sorted %s" % (n, name, err, srt))
for name, e in worst.items():
    print("%-26s worst eigenvalue error / |T| over the sizes: %.3g" % (name, e))
