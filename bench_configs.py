#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs on ONE GPU (device-resident
timings from HIP events on the library's stream; not the driver's headline line --
that is bench.py).  Prints one JSON object per config.

    python bench_configs.py [B] [C] [D] [E] [--reps 5]
"""
import contextlib
import ctypes as C
import io
import json
import sys
import time

import os

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))

import pythtb_amd as tb  # noqa: E402
from pythtb_amd import _lib  # noqa: E402

import helpers as hp  # noqa: E402  (model builders written against the public API)

lib = _lib.lib


def timed(ctx, fn, reps):
    fn()
    ctx.sync()
    best = 1e30
    for _ in range(reps):
        ctx.timer_begin()
        fn()
        best = min(best, ctx.timer_end())
    return best


def grid_handle(ctx, model, mesh):
    n = model._nsta
    h = C.c_void_p()
    m32 = np.ascontiguousarray(mesh, dtype=np.int32)
    _lib.check(lib.tbk_wfs_create(ctx.handle, len(mesh), _lib.iptr(m32), n, n, C.byref(h)))
    pbc = np.ascontiguousarray(np.array([np.repeat(np.exp(-2j * np.pi * model._orb[:, model._per[d]]), model._nspin)
                                         for d in range(len(mesh))]))
    return h, pbc


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
    which = args or ["B", "C", "D", "E"]
    ctx = _lib.default_context()
    out = []
    if "B" in which:
        m = hp.haldane(tb.tb_model, 0.2)
        k = m.k_uniform_mesh([1024, 1024])
        nk = len(k)
        hm = m._device_model()
        kd, ed, vd = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, k.nbytes, C.byref(kd)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 2 * 8, C.byref(ed)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 4 * 16, C.byref(vd)))
        _lib.check(lib.tbk_dev_upload(ctx.handle, kd, k.ctypes.data_as(C.c_void_p), k.nbytes))
        t_val = timed(ctx, lambda: _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None)), reps)
        t_vec = timed(ctx, lambda: _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, vd)), reps)
        def wall(fn, reps=3):
            fn()
            best = 1e30
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                best = min(best, time.perf_counter() - t0)
            return best
        t_host = wall(lambda: m.solve_all(k))                          # k uploaded, eigenvalues downloaded
        t_mesh = wall(lambda: m.solve_all_mesh([1024, 1024]))          # k generated on the device
        t_dos = wall(lambda: m.dos_mesh([1024, 1024], 50, range=(-4.0, 4.0)))   # nothing bulky crosses PCIe
        out.append({"config": "B: Haldane solve_all 1024^2", "nk": nk, "eval_only_ms": t_val, "with_vectors_ms": t_vec,
                    "kpts_per_s_eval": nk / t_val * 1e3, "kpts_per_s_vec": nk / t_vec * 1e3,
                    "hbm_GBs_eval": 32 * nk / t_val / 1e6, "hbm_GBs_vec": 96 * nk / t_vec / 1e6,
                    "python_call_incl_pcie_s": t_host, "kpts_per_s_python_call": nk / t_host,
                    "solve_all_mesh_call_s": t_mesh, "dos_mesh_call_s": t_dos, "kpts_per_s_dos_call": nk / t_dos})
    if "C" in which or "D" in which:
        for tag, model, mesh, occ in (("C: Haldane 2048^2", hp.haldane(tb.tb_model, 0.0), [2049, 2049], [0]),
                                      ("D: Kane-Mele 4096x512", hp.kane_mele(tb.tb_model, "odd"), [4097, 513], [0, 1])):
            if tag[0] not in which:
                continue
            hw, pbc = grid_handle(ctx, model, mesh)
            hm = model._device_model()
            start = np.array([-0.5, -0.5])
            n = model._nsta
            nk = (mesh[0] - 1) * (mesh[1] - 1)
            occ32 = np.array(occ, dtype=np.int32)
            t_solve = timed(ctx, lambda: _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0])), reps)
            t_flux = timed(ctx, lambda: _lib.check(lib.tbk_berry_flux_async(hw, _lib.iptr(occ32), len(occ), 0, 1, 0)), reps)
            res = {"config": tag, "nk": nk, "solve_grid_ms": t_solve, "kpts_per_s": nk / t_solve * 1e3,
                   "solve_hbm_GBs": 16 * n * n * nk / t_solve / 1e6, "berry_flux_ms": t_flux,
                   "plaq_per_s": nk / t_flux * 1e3, "flux_hbm_GBs": 16 * len(occ) * n * nk / t_flux / 1e6}
            if tag[0] == "D":
                phases = np.zeros(mesh[1] * 2)
                t0 = time.perf_counter()
                _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 2, 0, 1, _lib.dptr(phases)))
                res["wilson_loop_call_ms"] = (time.perf_counter() - t0) * 1e3
                ctx.prof_enable(True)
                ctx.prof_reset()
                _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 2, 0, 1, _lib.dptr(phases)))
                res["wilson_loop_kernels_ms"] = {k: v["total_ms"] for k, v in ctx.prof_report().items()}
                ctx.prof_enable(False)
                nlinks = (mesh[0] - 1) * mesh[1]
                res["links_per_s"] = nlinks / (sum(res["wilson_loop_kernels_ms"].values()) * 1e-3)
            _lib.check(lib.tbk_wfs_free(hw))
            out.append(res)
    if "E" in which:
        with contextlib.redirect_stdout(io.StringIO()):
            model = hp.cubic16(tb.tb_model)
        mesh = [65, 65, 65]                                   # 64^3 sub-mesh of the 256^3 config (4.4 GB of _wfs)
        hw, pbc = grid_handle(ctx, model, mesh)
        hm = model._device_model()
        start = np.zeros(3)
        nk = 64 ** 3
        t_solve = timed(ctx, lambda: _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0])), max(1, reps // 2))
        occ32 = np.arange(8, dtype=np.int32)
        phases = np.zeros(65 * 65)
        t0 = time.perf_counter()
        _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 8, 2, 0, _lib.dptr(phases)))
        t_phase = time.perf_counter() - t0
        out.append({"config": "E: cubic16 (888 hops) 64^3 sub-mesh", "nk": nk, "solve_grid_ms": t_solve,
                    "kpts_per_s": nk / t_solve * 1e3, "berry_phase_8band_call_ms": t_phase * 1e3,
                    "links_per_s": 65 * 65 * 64 / t_phase})
        _lib.check(lib.tbk_wfs_free(hw))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
