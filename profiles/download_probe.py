#!/usr/bin/env python3
"""What reading a resident array back costs: wf_array.to_host() of the 2049^2 Haldane array (268 MB) and solve_all with
eigenvectors on 40 000 points of a 16-state model (164 MB).  (Round 4 tried 8 MB pieces through two pinned buffers with the host
copying piece i under the DMA of piece i + 1 -- TBK_D2H_PIPELINE, not shipped: 12 GB/s against the 23.5 GB/s of one plain
hipMemcpyAsync into the fresh numpy array; the host copy into untouched pages is the slow part.  microbench/pcie_d2h.hip has the
raw rates: pinned 57 GB/s, hipHostMalloc 0.2 ms per MB.)
    python profiles/download_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
m = hp.haldane(tb.tb_model)
m16 = hp.random_model(tb.tb_model, 16, 2, 1, seed=16, nhop=64, rmax=1)
k = np.random.default_rng(0).random((40000, 2))
for pipe in (0, 0):
    with _lib.knob("TBK_D2H_PIPELINE", pipe):
        w = tb.wf_array(m, [2049, 2049])
        w.solve_on_grid([-0.5, -0.5])
        t0 = time.perf_counter(); a = w.to_host(); t1 = time.perf_counter()
        ev, vec = m16.solve_all(k, eig_vectors=True)
        t2 = time.perf_counter(); ev, vec = m16.solve_all(k, eig_vectors=True); t3 = time.perf_counter()
        print(json.dumps({"pipeline": pipe, "to_host_268MB_ms": round(1e3 * (t1 - t0), 2), "GBs": round(a.nbytes / (t1 - t0) / 1e9, 1),
                          "solve_all_vectors_164MB_ms": round(1e3 * (t3 - t2), 2)}))
        del w, a
