#!/usr/bin/env python3
"""Bands degenerate on the whole mesh (H_0 x 1_2) against a generic model of the same size, through every eigenvector regime:
time of solve_on_grid and quality (residual, orthonormality against the model's own H(k)) of a sample.
    python profiles/degenerate_regimes_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()


def model(nspin, norb, seed):
    rng = np.random.default_rng(seed)
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((norb, 2)), nspin=nspin)
    m.set_onsite(list(rng.standard_normal(norb)))
    for i in range(norb):
        for j in range(i + 1, min(norb, i + 6)):
            m.set_hop(0.3 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0])
    for R in ([1, 0], [0, 1]):
        for i in range(norb):
            for j in range(max(0, i - 2), min(norb, i + 3)):
                m.set_hop(0.2 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    return m


for n, mesh in ((8, [257, 257]), (16, [257, 257]), (32, [129, 129]), (64, [65, 65]), (96, [33, 33]), (160, [17, 17])):
    out = {"n": n, "mesh": mesh}
    for name, m in (("degenerate", model(2, n // 2, n)), ("generic", model(1, n, n))):
        w = tb.wf_array(m, mesh)
        w.solve_on_grid([0.05, -0.1])
        ctx.timer_begin(); w.solve_on_grid([0.05, -0.1]); t = ctx.timer_end()
        host = w.to_host().reshape(-1, n, n)
        idx = np.linspace(0, host.shape[0] - 1, 40).astype(int)
        res = orth = 0.0
        for p in idx:
            i, j = divmod(int(p), mesh[1])
            k = [0.05 + (i % (mesh[0] - 1)) / (mesh[0] - 1), -0.1 + (j % (mesh[1] - 1)) / (mesh[1] - 1)]
            H = m._gen_ham(k).reshape(n, n)
            V = host[p]
            if i == mesh[0] - 1 or j == mesh[1] - 1:
                continue                      # (periodic images carry the boundary phase)
            ev = np.einsum("bi,ij,bj->b", V.conj(), H, V).real
            res = max(res, np.abs(V @ H.T - ev[:, None] * V).max() / np.abs(ev).max())
            orth = max(orth, np.abs(V.conj() @ V.T - np.identity(n)).max())
        out[name] = {"ms": round(t, 3), "residual": float("%.2e" % res), "orth": float("%.2e" % orth)}
    print(json.dumps(out))
