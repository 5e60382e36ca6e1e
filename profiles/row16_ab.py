#!/usr/bin/env python3
"""A/B of the n = 9..16 solvers (TBK_ROW16=0: wavefront-per-matrix LDS kernel with warm start; default:
one DPP row per matrix, registers, cold start) on list workloads: mesh-ordered and random k."""
import ctypes as C
import json
import os
import sys

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import pythtb_amd as tb  # noqa: E402
from pythtb_amd import _lib  # noqa: E402
import helpers as hp  # noqa: E402

lib, ctx = _lib.lib, _lib.default_context()
rng = np.random.default_rng(0)
for tag, m in (("cubic16", hp.cubic16(tb.tb_model)), ("random n=12", hp.random_model(tb.tb_model, 12, 3, 1, 4, nhop=60, rmax=1)),
               ("random n=9", hp.random_model(tb.tb_model, 9, 2, 1, 5, nhop=40, rmax=1))):
    n = m._nsta
    d = m._dim_k
    nk = 1 << 18
    side = int(round(nk ** (1.0 / d)))
    mesh_k = m.k_uniform_mesh([side] * d)
    nk = len(mesh_k)
    rand_k = rng.uniform(-0.5, 0.5, size=(nk, d))
    hm = m._device_model()
    kd, ed, vd = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * d * 8, C.byref(kd)))
    _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * n * 8, C.byref(ed)))
    _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * n * n * 16, C.byref(vd)))
    out = {"model": tag, "n": n, "nk": nk}
    for kname, k in (("mesh", mesh_k), ("random", rand_k)):
        kk = np.ascontiguousarray(k)
        _lib.check(lib.tbk_dev_upload(ctx.handle, kd, kk.ctypes.data_as(C.c_void_p), kk.nbytes))
        for vname, vp in (("eval", None), ("evec", vd)):
            best = 1e30
            for rep in range(3):
                ctx.timer_begin()
                _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, vp))
                t = ctx.timer_end()
                if rep:
                    best = min(best, t)
            out["%s_%s_ms" % (kname, vname)] = round(best, 3)
    print(json.dumps(out))
    for p in (kd, ed, vd):
        _lib.check(lib.tbk_dev_free(ctx.handle, p))
