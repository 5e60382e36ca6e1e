#!/usr/bin/env python3
"""Eigenvectors of 65..1024 states: the direct path (tbk_solve_trigv.inl) against the Jacobi solvers (TBK_TRIGV=0) -- Haldane ribbons
on a k path (solve_all with eigenvectors) and random supplied matrices; eigenvalue error, residual and orthonormality against
numpy, wall-clock of the call and the kernels' own brackets.    python profiles/trigv_probe.py [width x nk ...]"""
import contextlib, json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
from oracle import tb_oracle as orc
ctx = _lib.default_context()

def quality(ham, ev, V):
    nk, n = ham.shape[0], ham.shape[1]
    ref = np.linalg.eigvalsh(ham)
    nrm = np.abs(ref).max(axis=1)
    Vk = V.transpose(1, 0, 2)
    res = np.abs(np.einsum("kij,kbj->kbi", ham, Vk) - Vk * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / nrm
    orth = np.abs(np.einsum("kbi,kci->kbc", Vk.conj(), Vk) - np.eye(n)).reshape(nk, -1).max(axis=1)
    return float((np.abs(ev.T - ref).max(axis=1) / nrm).max()), float(res.max()), float(orth.max())

cases = [(35, 256), (64, 512), (100, 101), (150, 101), (150, 16), (400, 8), (512, 6)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for width, nk in cases:
    rib = hp.quiet(hp.haldane(tb.tb_model, 0.3).cut_piece, width, 1)
    n = 2 * width
    k = np.linspace(0.0, 1.0, nk, endpoint=False)[:, None] + 0.013
    ham = orc.ham_batch(rib, k)
    out = {"n": n, "nk": nk}
    for tag, env in (("trigv", {}), ("jacobi", {"TBK_TRIGV": "0"})):
        with contextlib.ExitStack() as st:
            for kk, v in env.items():
                st.enter_context(_lib.knob(kk, v))
            rib.solve_all(k[:2], eig_vectors=True)
            ev, V = rib.solve_all(k, eig_vectors=True)
            t0 = time.perf_counter(); ev, V = rib.solve_all(k, eig_vectors=True); t = time.perf_counter() - t0
            ctx.prof_enable(1); ctx.prof_reset(); rib.solve_all(k, eig_vectors=True); rep = ctx.prof_report(); ctx.prof_enable(0)
            e, r, o = quality(ham, ev, V)
            out[tag] = {"call_ms": 1e3 * t, "eval_err": e, "resid": r, "orth": o,
                        "kernels_ms": {kn: round(v["total_ms"], 3) for kn, v in rep.items()}}
    print(json.dumps(out), flush=True)
