#!/usr/bin/env python3
"""Large-n solver regimes (workgroup n=65..256, whole-chip n=257..2048): wall-clock of the Python call
next to numpy.linalg.eigh/eigvalsh on the host cores of the same box (the reference's own path)."""
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
import helpers as hp  # noqa: E402
from oracle import tb_oracle as orc  # noqa: E402


def wall(fn, reps=3):
    fn()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


h = hp.haldane(tb.tb_model, 0.0)
cases = [("ribbon n=40, 1024 k", h.cut_piece(20, 1), np.linspace(0, 1, 1024)),
         ("ribbon n=128, 32 k", h.cut_piece(64, 1), np.linspace(0, 1, 32)),
         ("ribbon n=128, 128 k", h.cut_piece(64, 1), np.linspace(0, 1, 128)),
         ("ribbon n=128, 512 k", h.cut_piece(64, 1), np.linspace(0, 1, 512)),
         ("ribbon n=200, 128 k", h.cut_piece(100, 1), np.linspace(0, 1, 128)),
         ("ribbon n=400, 128 k", h.cut_piece(200, 1), np.linspace(0, 1, 128)),
         ("flake n=800 (haldane_fin), 1 matrix", h.cut_piece(20, 0).cut_piece(20, 1), None),
         ("flake n=1800, 1 matrix", h.cut_piece(30, 0).cut_piece(30, 1), None)]
for tag, m, k in cases:
    om = orc.Model.from_tables(orc.model_tables(m))
    kk = None if k is None else k.reshape(-1, 1)
    t_val = wall(lambda: m.solve_all(k))
    t_vec = wall(lambda: m.solve_all(k, eig_vectors=True))
    hams = orc.ham_batch(om, kk) if kk is not None else np.array([orc.gen_ham(om)])
    skip_np = bool(os.environ.get("SKIP_NUMPY"))
    t_np_val = 0.0 if skip_np else wall(lambda: [np.linalg.eigvalsh(x) for x in hams], reps=1)
    t_np_vec = 0.0 if skip_np else wall(lambda: [np.linalg.eigh(x) for x in hams], reps=1)
    ev = m.solve_all(k)
    ref = np.array([np.linalg.eigvalsh(x) for x in hams])          # (nk, n)
    err = np.max(np.abs(ev.reshape(m._nsta, -1).T - ref))
    print(json.dumps({"case": tag, "nsta": m._nsta, "gpu_eval_s": t_val, "gpu_evec_s": t_vec, "numpy_eigvalsh_s": t_np_val,
                      "numpy_eigh_s": t_np_vec, "max_abs_eval_diff": err}))
