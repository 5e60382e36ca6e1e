"""Periodic-image column of k_grid_rows (n = 3, 4): written by the lanes that hold column 0 (TBK_GRID_IMG=1) against the column solved on its own (TBK_GRID_IMG=0)."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pythtb_amd as tb, helpers as hp
from pythtb_amd import _lib
for n in (3, 4):
    m = hp.kane_mele(tb.tb_model) if n == 4 else hp.random_model(tb.tb_model, n, 2, 1, seed=40 + n, nhop=4 * n, rmax=1)
    for mesh in ([65, 65], [130, 9], [5, 129], [4, 66]):
        out = {}
        for img in (0, 1):
            with _lib.knob("TBK_GRID_IMG", img):
                w = tb.wf_array(m, mesh)
                w.solve_on_grid([0.13, -0.21])
                out[img] = w.to_host().copy()
        d = np.abs(out[0] - out[1])
        print(n, mesh, 'max diff all', d.max(), 'image col', d[:, -1].max(), 'others', d[:, :-1].max())
