#!/usr/bin/env python3
"""solve_on_grid + berry_flux on meshes of the same size but different shapes (the row kernels map 64-point chunks of the LAST
axis to a wavefront): time per point.   python profiles/mesh_shape_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
ctx = _lib.default_context()
for name, m in (("haldane", hp.haldane(tb.tb_model)), ("kane_mele", hp.kane_mele(tb.tb_model))):
    for mesh in ([1025, 1025], [16385, 65], [65, 16385], [131073, 9], [9, 131073], [349526, 4], [4, 349526]):
        w = tb.wf_array(m, mesh)
        w.solve_on_grid([0.0, 0.0]); w.berry_flux([0])
        ctx.timer_begin(); w.solve_on_grid([0.0, 0.0]); t1 = ctx.timer_end()
        ctx.timer_begin(); w.berry_flux([0]); t2 = ctx.timer_end()
        npt = mesh[0] * mesh[1]
        print(json.dumps({"model": name, "mesh": mesh, "solve_ms": round(t1, 4), "flux_ms": round(t2, 4),
                          "solve_ns_per_point": round(1e6 * t1 / npt, 3), "flux_ns_per_point": round(1e6 * t2 / npt, 3)}))
