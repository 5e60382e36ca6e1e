#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs (device-resident timings from HIP events on the
library's stream; not the driver's headline line -- that is bench.py).  One JSON object per config.

    python bench_configs.py [B] [C] [D] [E] [R] [W] [H] [P] [L] [--reps 5] one GPU (not in BASELINE.json: R ribbons / mid-size meshes, P late Berry kernels, L the k-list
                                                                        kernels on an 8192^2 list, W the
                                                                        Wannier90 silicon model, H hybrid Wannier centres (8f-1))
    python bench_configs.py --gpus N [B] [D] [E] [--side 257]           N GPUs, one process each (starts its own ranks; the
                                                                        same under torch.distributed.run --nproc-per-node N)

Multi-GPU legs (BASELINE configs[1], [3] and [4]; drivers in pythtb_amd/multi.py):
  B  Haldane solve_all on the 1024^2 k list: contiguous chunks of the list, ONE all-gather-v of the eigenvalues into the
     band-major (2, nk) array (also E's solve_all leg: (16, 256^3), 268 MB contributed per rank).
  D  Kane-Mele wf_array([4097,513]): the 513 Wilson loops along axis 0 are sharded along axis 1 (65 + 7 x 64 for
     8 ranks); every rank solves its own window of the global mesh and the (513, 2) eigenphase array is assembled by
     ONE all-gather-v.
  E  cubic16 wf_array([257]*3): slabs along axis 0 with a recomputed halo plane, berry_phase(range(8), dir=2) per slab,
     the (257, 257) phase array and the min gaps assembled by the gather.
The RCCL communicator is brought up first and the gathers of the reported runs are tbk_comm_allgatherv[_rows]_f64 (RCCL over
xGMI); every RCCL call runs under a time limit, the rendezvous socket (launch.Rendezvous; TBK_RENDEZVOUS=gloo: torch) is the
labelled fallback (exit status 4).  Time per config = max over
ranks of one pass including its gather.
"""
import contextlib
import ctypes as C
import io
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))


def timed(ctx, fn, reps):
    fn()
    ctx.sync()
    best = 1e30
    for _ in range(reps):
        ctx.timer_begin()
        fn()
        best = min(best, ctx.timer_end())
    return best


def grid_handle(_lib, lib, ctx, model, mesh):
    n = model._nsta
    h = C.c_void_p()
    m32 = np.ascontiguousarray(mesh, dtype=np.int32)
    _lib.check(lib.tbk_wfs_create(ctx.handle, len(mesh), _lib.iptr(m32), n, n, C.byref(h)))
    pbc = np.ascontiguousarray(np.array([np.repeat(np.exp(-2j * np.pi * model._orb[:, model._per[d]]), model._nspin)
                                         for d in range(len(mesh))]))
    return h, pbc


def single_gpu(which, reps):
    import pythtb_amd as tb
    from pythtb_amd import _lib
    import helpers as hp
    lib = _lib.lib
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
        t_host = wall(lambda: m.solve_all(k))                          # the drop-in call: the list is uploaded (16 B per k-point) and solved as a list
        ctx.prof_enable(1)
        ctx.prof_reset()
        m.solve_all_mesh([1024, 1024])
        prof_mesh = {kk: v["total_ms"] / max(v["launches"], 1) for kk, v in ctx.prof_report().items()}
        ctx.prof_enable(0)
        t_list = t_host
        t_mesh = wall(lambda: m.solve_all_mesh([1024, 1024]))          # k generated on the device
        t_dos = wall(lambda: m.dos_mesh([1024, 1024], 50, range=(-4.0, 4.0)))   # nothing bulky crosses PCIe
        out.append({"config": "B: Haldane solve_all 1024^2", "nk": nk, "eval_only_ms": t_val, "with_vectors_ms": t_vec,
                    "kpts_per_s_eval": nk / t_val * 1e3, "kpts_per_s_vec": nk / t_vec * 1e3,
                    "hbm_GBs_eval": 32 * nk / t_val / 1e6, "hbm_GBs_vec": 96 * nk / t_vec / 1e6,
                    "python_call_incl_pcie_s": t_host, "kpts_per_s_python_call": nk / t_host,
                    "python_call_list_path_s": t_list, "mesh_path_kernels_ms": prof_mesh,
                    "mesh_evals_hbm_GBs": 32 * nk / prof_mesh["mesh_evals"] / 1e6 if "mesh_evals" in prof_mesh else None,
                    "solve_all_mesh_call_s": t_mesh, "dos_mesh_call_s": t_dos, "kpts_per_s_dos_call": nk / t_dos})
    if "L" in which:
        # (not a BASELINE config) the k-list kernels PAST the last-level cache: configs[1]'s leg moves
        # 33.5 MB -- 4 us at HBM peak, so its 17 us mostly measure launch + completion latency.  Here the same kernels on an 8192^2
        # list: 1.07 GB of k read, 1.07 GB (2 states) / 2.15 GB (4 states) of eigenvalues written, everything resident, the list made
        # on the device (tbk_k_uniform_mesh_dev).  Algorithmic bytes 8 (d + n) per k-point.
        HBM = 8000.0
        side = 8192
        nk = side * side
        mesh32 = np.array([side, side], dtype=np.int32)
        kd = C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 16, C.byref(kd)))
        _lib.check(lib.tbk_k_uniform_mesh_dev(ctx.handle, 2, _lib.iptr(mesh32), kd))
        legs = []
        for tag, model in (("haldane_2states", hp.haldane(tb.tb_model, 0.2)), ("kane_mele_4states", hp.kane_mele(tb.tb_model, "odd"))):
            n = model._nsta
            hm = model._device_model()
            ed = C.c_void_p()
            _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * n * 8, C.byref(ed)))
            t = timed(ctx, lambda: _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None)), reps)
            ctx.prof_enable(1)
            ctx.prof_reset()
            _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None))
            kern = {kk: v["total_ms"] / max(v["launches"], 1) for kk, v in ctx.prof_report().items()}
            ctx.prof_enable(0)
            # spot check against the same kernel on a short list (the first and the last 4096 points)
            probe = np.zeros((n, 4096))
            chk = []
            for first in (0, nk - 4096):
                for b in range(n):
                    _lib.check(lib.tbk_dev_download(ctx.handle, probe[b].ctypes.data_as(C.c_void_p),
                                                    C.c_void_p(ed.value + 8 * (b * nk + first)), 8 * 4096))
                kk = np.array([[(i // side) / side, (i % side) / side] for i in range(first, first + 4096)])
                chk.append(float(np.max(np.abs(probe - model.solve_all(kk)))))
            nbytes = 8 * (2 + n) * nk
            legs.append({"leg": "list_%s_%dsq" % (tag, side), "nk": nk, "n": n, "eval_only_ms": t, "kernels_ms": kern,
                         "kpts_per_s": nk / t * 1e3, "max_abs_diff_vs_short_list": max(chk),
                         "roofline": {"bound": "hbm", "algorithmic_bytes": nbytes, "achieved": nbytes / (t * 1e-3) / 1e9, "peak": HBM,
                                      "unit": "GB/s", "frac": nbytes / (t * 1e-3) / 1e9 / HBM}})
            _lib.check(lib.tbk_dev_free(ctx.handle, ed))
        _lib.check(lib.tbk_dev_free(ctx.handle, kd))
        out.append({"config": "L: k-list kernels past the last-level cache (8192^2 k list, eigenvalues only, resident)", "legs": legs})
    if "C" in which or "D" in which:
        for tag, model, mesh, occ in (("C: Haldane 2048^2", hp.haldane(tb.tb_model, 0.0), [2049, 2049], [0]),
                                      ("D: Kane-Mele 4096x512", hp.kane_mele(tb.tb_model, "odd"), [4097, 513], [0, 1])):
            if tag[0] not in which:
                continue
            hw, pbc = grid_handle(_lib, lib, ctx, model, mesh)
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
                _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 2, 0, 1, _lib.dptr(phases)))   # (first call: scratch allocation)
                best = 1e9
                for _ in range(max(1, reps)):
                    t0 = time.perf_counter()
                    _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 2, 0, 1, _lib.dptr(phases)))
                    best = min(best, time.perf_counter() - t0)
                res["wilson_loop_call_ms"] = best * 1e3
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
        hw, pbc = grid_handle(_lib, lib, ctx, model, mesh)
        hm = model._device_model()
        start = np.zeros(3)
        nk = 64 ** 3
        t_solve = timed(ctx, lambda: _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, mesh[0])), max(1, reps // 2))
        occ32 = np.arange(8, dtype=np.int32)
        phases = np.zeros(65 * 65)
        _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 8, 2, 0, _lib.dptr(phases)))   # (first call: workspace allocation)
        t_phase = 1e9
        for _ in range(max(1, reps // 2)):
            t0 = time.perf_counter()
            _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 8, 2, 0, _lib.dptr(phases)))
            t_phase = min(t_phase, time.perf_counter() - t0)
        # (beside the config: the Wilson-loop eigenphases of the same 8 bands, berry_evals=True -- link polar factors on the matrix cores)
        wil = np.zeros(65 * 65 * 8)
        _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 8, 2, 1, _lib.dptr(wil)))
        t_wil = 1e9
        for _ in range(max(1, reps // 2)):
            t0 = time.perf_counter()
            _lib.check(lib.tbk_berry_phase(hw, _lib.iptr(occ32), 8, 2, 1, _lib.dptr(wil)))
            t_wil = min(t_wil, time.perf_counter() - t0)
        wil = wil.reshape(-1, 8)
        # configs[4]'s solve_all leg: eigenvalues of the k_uniform_mesh list, list and results resident (k_e16<0, false>)
        mesh64 = np.array([64, 64, 64], dtype=np.int32)
        kd, ed = C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 24, C.byref(kd)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 16 * 8, C.byref(ed)))
        _lib.check(lib.tbk_k_uniform_mesh_dev(ctx.handle, 3, _lib.iptr(mesh64), kd))
        t_all = timed(ctx, lambda: _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None)), max(1, reps // 2))
        ev0 = np.zeros(16)
        for b in range(16):
            _lib.check(lib.tbk_dev_download(ctx.handle, ev0[b:b + 1].ctypes.data_as(C.c_void_p), C.c_void_p(ed.value + 8 * b * nk), 8))
        ev_ref = model.solve_all(np.zeros((1, 3)), eig_vectors=True)[0][:, 0]      # (another kernel: the counters of k_e16<0, false> stay those of the leg)
        _lib.check(lib.tbk_dev_free(ctx.handle, kd))
        _lib.check(lib.tbk_dev_free(ctx.handle, ed))
        out.append({"config": "E: cubic16 (888 hops) 64^3 sub-mesh", "nk": nk, "solve_grid_ms": t_solve,
                    "solve_all_eigenvalues_ms": t_all, "solve_all_ns_per_point": t_all * 1e6 / nk,
                    "solve_all_first_point_max_abs_diff": float(np.max(np.abs(ev0 - ev_ref))),
                    "kpts_per_s": nk / t_solve * 1e3, "berry_phase_8band_call_ms": t_phase * 1e3,
                    "links_per_s": 65 * 65 * 64 / t_phase, "wilson_loop_8band_call_ms": t_wil * 1e3,
                    "wilson_links_per_s": 65 * 65 * 64 / t_wil,
                    "wilson_sum_vs_det_phase": float(np.max(np.abs(np.angle(np.exp(1j * (wil.sum(axis=1) - phases))))))})
        _lib.check(lib.tbk_wfs_free(hw))
    if "W" in which:
        # (not a BASELINE config) SURVEY.md 8f-4: the Wannier90 importer's regime -- silicon, 8 Wannier functions, 2972 hopping terms
        # (website/local/w90_example/example_a, committed under tests/golden/w90_silicon): the long hopping table is where H(k)
        # assembly dominates.  solve_all eigenvalues on a 48^3 uniform mesh (k generated on the device) and solve_on_grid on 65^3.
        HBM, VALU = 8000.0, 256 * 4 * 2.4e9 / 4.0
        with contextlib.redirect_stdout(io.StringIO()):
            si = tb.w90(os.path.join(_ROOT, "tests", "golden", "w90_silicon"), "silicon")
            m = si.model()
        hm = m._device_model()
        nterm = len(m._hoppings)
        mesh = np.array([48, 48, 48], dtype=np.int32)
        nk = int(np.prod(mesh))
        kd, ed = C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 3 * 8, C.byref(kd)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 8 * 8, C.byref(ed)))
        _lib.check(lib.tbk_k_uniform_mesh_dev(ctx.handle, 3, _lib.iptr(mesh), kd))
        t_list = timed(ctx, lambda: _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None)), reps)
        G = 65
        hw, pbc = grid_handle(_lib, lib, ctx, m, [G, G, G])
        start = np.zeros(3)
        t_grid = timed(ctx, lambda: _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, G)), max(1, reps // 2))
        ctx.prof_enable(1)
        ctx.prof_reset()
        _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None))
        _lib.check(lib.tbk_wfs_solve_grid_async(hw, hm, _lib.dptr(start), _lib.dptr(pbc.view(float)), 0, G))
        ctx.sync()
        kern = {kk: v["total_ms"] for kk, v in ctx.prof_report().items()}
        ctx.prof_enable(0)
        # the assembly: one complex multiply-add (8 flops) per merged (slot, R) term and k-point; the R-grouped table holds
        # nR x 36 slots of 16 bytes and is read once per k-point through scalar loads (L2)
        out.append({"config": "W: w90 silicon (8 Wannier functions, %d hopping terms): solve_all 48^3 (eigenvalues) + solve_on_grid 65^3" % nterm,
                    "nterm": nterm, "nk_list": nk, "solve_list_eval_ms": t_list, "kpts_per_s_list": nk / t_list * 1e3,
                    "term_evals_per_s_list": nk / t_list * 1e3 * nterm,
                    "grid": G, "solve_on_grid_ms": t_grid, "kpts_per_s_grid": G ** 3 / t_grid * 1e3,
                    "term_evals_per_s_grid": G ** 3 / t_grid * 1e3 * nterm, "kernels_ms": kern,
                    "roofline": {"solve_list": {"bound": "fp64 VALU issue: the assembly's multiply-adds (~12 k instructions per k-point; the table streams from L2 by scalar loads), then Householder + QL on (d, e) (~3.5 k; cyclic Jacobi, ~20 k, until round 5)",
                                                "hbm_frac": 8 * (3 + 8) * nk / (t_list * 1e-3) / 1e9 / HBM, "algorithmic_bytes_per_k": 8 * (3 + 8),
                                                "assembly_useful_tflops": 8.0 * nterm * nk / (t_list * 1e-3) / 1e12},
                                 "solve_grid": {"bound": "fp64 VALU issue at one wavefront per SIMD (the reflectors' LDS): H(k) from the rows' coefficient cells (reg_assemble_cells, round 5: ~3.5 k instead of 14.5 k instructions per wavefront), then Householder + QL with Q and the bands' back-transformation (~5 k); assembly_useful_tflops counts the merged terms as if each were still evaluated per point",
                                                "hbm_frac": 16 * 64 * G ** 3 / (t_grid * 1e-3) / 1e9 / HBM,
                                                "algorithmic_bytes_per_k": 16 * 64,
                                                "assembly_useful_tflops": 8.0 * nterm * G ** 3 / (t_grid * 1e-3) / 1e12}}})
        _lib.check(lib.tbk_wfs_free(hw))
        _lib.check(lib.tbk_dev_free(ctx.handle, kd))
        _lib.check(lib.tbk_dev_free(ctx.handle, ed))
    if "H" in which:
        # (not a BASELINE config) SURVEY.md 8f-1: hybrid Wannier centres of a cubic slab on a 2-D mesh (examples/cubic_slab_hwf.py's loop as
        # ONE batched call on the resident array): position matrix of the occupied block + its small eigen-solve per mesh point
        HBM = 8000.0
        with contextlib.redirect_stdout(io.StringIO()):
            m3 = tb.tb_model(3, 3, np.identity(3), [[0, 0, 0]])
            for Rv in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
                m3.set_hop(-1.0, 0, 0, Rv)
            slab = m3.cut_piece(16, 2, glue_edgs=False)
        mesh2 = [513, 513]
        w = tb.wf_array(slab, mesh2)
        w.solve_on_grid([0.0, 0.0])
        nocc = 8
        w.position_hwf_mesh(range(nocc), 2)
        ctx.sync()
        best = 1e9
        for _ in range(max(1, reps)):
            t0 = time.perf_counter()
            hwfc = w.position_hwf_mesh(range(nocc), 2)
            best = min(best, time.perf_counter() - t0)
        ctx.prof_enable(1)
        ctx.prof_reset()
        w.position_hwf_mesh(range(nocc), 2)
        kern = {kk: v["total_ms"] for kk, v in ctx.prof_report().items()}
        ctx.prof_enable(0)
        npt = mesh2[0] * mesh2[1]
        dev_ms = sum(kern.values())
        out.append({"config": "H: hybrid Wannier centres, 16-layer cubic slab, wf_array([513,513]), position_hwf_mesh(range(8), dir=2)",
                    "points": npt, "nocc": nocc, "n": 16, "call_ms_incl_download": best * 1e3, "points_per_s_call": npt / best,
                    "kernels_ms": kern, "device_ms": dev_ms, "points_per_s_device": npt / (dev_ms * 1e-3) if dev_ms > 0 else None,
                    "roofline": {"bound": "position matrices: HBM; then the batched 8 x 8 eigen-solve (k_solve_reg: the direct solver since round 5 when no vectors are asked for)",
                                 "algorithmic_bytes_per_point": 16 * nocc * 16 + 8 * nocc,
                                 "hbm_frac": (16 * nocc * 16 + 8 * nocc) * npt / (dev_ms * 1e-3) / 1e9 / HBM if dev_ms > 0 else None},
                    "check": {"centres_in_slab": bool(np.all((hwfc > -0.5) & (hwfc < 16.5))), "mean_centre": float(hwfc.mean())}})
    if "P" in which:
        # (not a BASELINE config) the Berry / position kernels that landed at the end of round 4 without rocprofv3 evidence (VERDICT r4
        # missing #5): Wilson-loop eigenphases of 3 and 4 bands (k_wilson_seg_reg), of 8 wide bands by polar factors on the matrix
        # cores (k_chain_prod_tile<..., POLAR>), berry_flux in planes that do not hold the fastest axis (k_flux_slices), the position
        # matrix of 16 of 16 states (k_position_matrix_tile).  Algorithmic bytes: the occupied vectors read once, 16 nocc n per point.
        HBM = 8000.0

        def leg(name, fn, nbytes, npts, unit):
            fn()
            ctx.sync()
            ctx.prof_enable(1)
            ctx.prof_reset()
            for _ in range(max(2, reps)):
                fn()
            rep = ctx.prof_report()
            ctx.prof_enable(0)
            kern = {kk: v["total_ms"] / max(v["launches"], 1) for kk, v in rep.items()}
            dev_ms = sum(v["total_ms"] for v in rep.values()) / max(2, reps)
            return {"leg": name, "kernels_ms_per_launch": kern, "device_ms_per_call": dev_ms, unit + "_per_s": npts / (dev_ms * 1e-3),
                    "roofline": {"bound": "hbm", "algorithmic_bytes": nbytes, "achieved": nbytes / (dev_ms * 1e-3) / 1e9, "peak": HBM, "unit": "GB/s",
                                 "frac": nbytes / (dev_ms * 1e-3) / 1e9 / HBM}}
        legs = []
        for nb in (3, 4):                                  # Wilson loops of 3 / 4 bands of a 2 nb-state model, 257 strings of 1024 links
            mw = hp.random_model(tb.tb_model, 2 * nb, 2, 1, 7 + nb)
            ww = tb.wf_array(mw, [1025, 257])
            ww.solve_on_grid([0.0, 0.0])
            occ = list(range(nb))
            legs.append(leg("wilson_%dbands_1024x257" % nb, lambda: ww.berry_phase(occ, 0, contin=False, berry_evals=True),
                            16 * nb * 2 * nb * 1025 * 257, 1024 * 257, "links"))
            del ww
        m16 = hp.cubic16(tb.tb_model)
        w16 = tb.wf_array(m16, [65, 65, 65])
        w16.solve_on_grid([0.0, 0.0, 0.0])
        legs.append(leg("wilson_polar_8of16_65cubed_dir2", lambda: w16.berry_phase(range(8), 2, contin=False, berry_evals=True),
                        16 * 8 * 16 * 65 ** 3, 64 * 65 * 65, "links"))
        legs.append(leg("flux_slices_8of16_65cubed_dirs01", lambda: w16.berry_flux(range(8), dirs=[0, 1]), 16 * 8 * 16 * 65 ** 3, 64 * 64 * 65, "plaquettes"))
        del w16
        m2 = hp.random_model(tb.tb_model, 2, 3, 1, seed=7, nhop=8, rmax=1)
        w2 = tb.wf_array(m2, [129, 129, 129])
        w2.solve_on_grid([0.0, 0.0, 0.0])
        legs.append(leg("flux_slices_1of2_129cubed_dirs01", lambda: w2.berry_flux([0], dirs=[0, 1]), 16 * 1 * 2 * 129 ** 3, 128 * 128 * 129, "plaquettes"))
        del w2
        with contextlib.redirect_stdout(io.StringIO()):
            m3 = tb.tb_model(3, 3, np.identity(3), [[0, 0, 0]])
            for Rv in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
                m3.set_hop(-1.0, 0, 0, Rv)
            slab = m3.cut_piece(16, 2, glue_edgs=False)
        ws = tb.wf_array(slab, [257, 257])
        ws.solve_on_grid([0.0, 0.0])
        legs.append(leg("position_16of16_257sq", lambda: ws.position_hwf_mesh(range(16), 2), (16 * 16 * 16 + 8 * 16) * 257 * 257, 257 * 257, "points"))
        out.append({"config": "P: Berry / position kernels of round 4's last hours (rocprofv3 evidence: profiles/r05fcfg)", "legs": legs})
    if "R" in which:
        # (not a BASELINE config) the widening rows of SURVEY.md 8f-2: ribbon band structures and a mid-size mesh solve, to keep
        # the direct paths for 17..1024 states under measurement.  Wall-clock of the Python calls, PCIe included.
        def wall(fn, r=3):
            fn()
            best = 1e9
            for _ in range(r):
                t0 = time.perf_counter()
                fn()
                best = min(best, time.perf_counter() - t0)
            return best * 1e3
        res = {"config": "R: ribbons / mid-size meshes (not in BASELINE.json)"}
        with contextlib.redirect_stdout(io.StringIO()):
            hal = hp.haldane(tb.tb_model, 0.2)
            rib = {w: hal.cut_piece(w, 1, glue_edgs=False) for w in (20, 64, 150)}
        k = np.linspace(0.0, 1.0, 512, endpoint=False)
        for w, m in rib.items():
            res["ribbon_n%d_512k_eigenvalues_ms" % (2 * w)] = wall(lambda m=m: m.solve_all(k))
        res["ribbon_n40_512k_with_vectors_ms"] = wall(lambda: rib[20].solve_all(k, eig_vectors=True))
        rng = np.random.default_rng(5)
        norb = 24
        with contextlib.redirect_stdout(io.StringIO()):
            m24 = tb.tb_model(3, 3, np.identity(3), rng.random((norb, 3)))
        m24.set_onsite(np.where(np.arange(norb) < norb // 2, -2.5, 2.5))
        for Rv in ([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]):
            for i in range(norb):
                for j in range(norb):
                    if Rv == [0, 0, 0] and j <= i:
                        continue
                    m24.set_hop(0.08 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, Rv)
        w24 = tb.wf_array(m24, [33, 33, 33])
        res["dense24_32cubed_solve_on_grid_ms"] = wall(lambda: w24.solve_on_grid([0.0, 0.0, 0.0]))
        res["dense24_kpts_per_s"] = 32 ** 3 / res["dense24_32cubed_solve_on_grid_ms"] * 1e3
        # (round 6) the step between 16 and 17 states: random sparse models on 33^3 points, with eigenvectors and without
        # (k_e16 up to 16 states; k_hh32 + k_ql32_lanes + k_tw32_vectors from 17: profiles/HISTORY.md)
        for nn in (16, 17, 24, 32):
            mm = hp.random_model(tb.tb_model, nn, 3, 1, seed=5, nhop=6 * nn, rmax=1)
            ww = tb.wf_array(mm, [33, 33, 33])
            ms = wall(lambda ww=ww: ww.solve_on_grid([0.0, 0.0, 0.0]))
            res["random%d_33cubed_solve_on_grid_ms" % nn] = ms
            res["random%d_ns_per_point" % nn] = ms * 1e6 / 33 ** 3
            kk = rng.uniform(-0.5, 0.5, (33 ** 3, 3))
            res["random%d_33cubed_solve_all_eigenvalues_ms_incl_pcie" % nn] = wall(lambda mm=mm, kk=kk: mm.solve_all(kk))
        out.append(res)
    for o in out:
        print(json.dumps(o))


def multi_gpu(which, side):
    """One process per GPU (started by main() through pythtb_amd/launch.py, or by torch.distributed.run): gloo rendezvous
    BEFORE anything touches the GPU, then the RCCL communicator, then the legs.  The gathers inside the drivers of
    pythtb_amd/multi.py -- the path's one collective each -- run through RCCL (tbk_comm_allgatherv[_rows]_f64); gloo is the
    labelled fallback, and then the exit status is 4."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import pythtb_amd as tb
    from pythtb_amd import _lib, multi, launch
    if os.environ.get("TBK_RENDEZVOUS", "socket") == "gloo":  # (optional: torch.distributed on the host, as rounds 2-5 did)
        import datetime
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))
        gloo = multi.GlooComm(dist)
    else:                                                     # the launcher's own TCP rendezvous: no torch in the run
        dist = launch.Rendezvous(rank=rank, world=world, timeout=900.0)
        gloo = multi.SocketComm(dist)
    import helpers as hp
    ctx = _lib.default_context()                              # device = LOCAL_RANK
    limit = float(os.environ.get("TBK_BENCH_RCCL_TIMEOUT", "120"))
    comm, err, hung = multi.rccl_bring_up(ctx, dist, rank, world, limit)
    status = 0
    if comm is None:
        sys.stderr.write("[bench_configs] rank %d: RCCL communicator unavailable (%s); gathers go through %s\n" % (rank, err, gloo.name))
        comm, gather, status = gloo, "%s (FALLBACK: rccl %s)" % (gloo.name, err), 4
    else:
        gather = "rccl gather-v / all-gather-v (tbk_comm_gatherv_rows_f64, tbk_comm_allgatherv_f64, RCCL over xGMI)"
    lines = []

    def per_rank_stats(st):
        """[{solve_ms, gather_ms, sent_bytes, recv_bytes}] of the last timed pass, one entry per rank (gloo, control plane)."""
        keys = ("solve_ms", "gather_ms", "sent_bytes", "recv_bytes")
        allv = gloo.allgatherv(np.array([float(st.get(k, 0.0)) for k in keys]), [len(keys)] * world).reshape(world, len(keys))
        return [dict(zip(keys, [float(x) for x in row])) for row in allv]

    def finish(code):
        if rank == 0:
            for ln in lines:
                print(json.dumps(ln), flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)

    def run(fn):
        """fn(comm) -> result.  One warm-up (allocations, table builds), one timed pass between barriers; every pass under a
        time limit, and the ranks agree on the outcome before going on (a rank that failed must not leave the others in a
        collective).  Returns (result, max-over-ranks seconds) or None."""
        out = None
        for timed_pass in (False, True):
            ctx.sync()
            dist.barrier()
            t0 = time.perf_counter()
            ok, val, h = multi.call_with_timeout(lambda: fn(comm), 10 * limit)
            ctx.sync() if ok else None
            t1 = time.perf_counter()
            if h:                                             # stuck inside a collective: report what is known and leave
                lines.append({"error": "a leg did not return within %g s" % (10 * limit), "gather": gather})
                finish(3)
            if not multi.agree(dist, ok):
                if not ok:
                    sys.stderr.write("[bench_configs] rank %d: %s\n" % (rank, val))
                return None
            out = (val, t1 - t0)
        tmax = float(gloo.allgatherv(np.array([out[1]]), [1] * world).max())
        return out[0], tmax

    if "B" in which:
        m = hp.haldane(tb.tb_model, 0.2)
        k = m.k_uniform_mesh([1024, 1024])
        nk = len(k)
        r = run(lambda c: multi.solve_all_sharded(m, k, c, rank, world))
        if r is None:
            lines.append({"config": "configs[1]", "error": "solve_all_sharded failed", "gather": gather})
            status = status or 6
        else:
            ev, t = r
            samp = np.arange(0, nk, 4099)
            same = bool(np.array_equal(ev[:, samp], m.solve_all(k[samp])))
            ln = {"config": "configs[1]: Haldane (delta 0.2) solve_all(k_uniform_mesh([1024,1024])), k list cut into %d chunks, ONE "
                            "all-gather-v of eval (2, nk) band-major" % world,
                  "n_gpus": world, "kpts": nk, "seconds_max_over_ranks_incl_upload_gather_download": t, "kpts_per_s": nk / t,
                  "chunks": [e - b for b, e in multi.plan_list(nk, world)], "gathered_bytes_per_rank": 8 * 2 * nk,
                  "check": {"sum": float(ev.sum()), "min": float(ev.min()), "max": float(ev.max()),
                            "sampled_columns_equal_unsharded": same}, "gather": gather}
            if hasattr(comm, "gatherv_rows_dev"):
                # the ROOTED gather (ret_eval is one array on one caller): ranks != 0 allocate no (2, nk) buffer
                st = {}
                r2 = run(lambda c: multi.solve_all_mesh_sharded(m, [1024, 1024], c, rank, world, root=0, stats=st))
                if r2 is not None:
                    ln["k_generated_on_device_rooted_gather_seconds"] = r2[1]
                    ln["per_rank"] = per_rank_stats(st)
                    if rank == 0:
                        ln["k_generated_on_device_equal"] = bool(np.array_equal(r2[0], ev))
            lines.append(ln)
    if "D" in which:
        m = hp.kane_mele(tb.tb_model, "odd")
        mesh = [4097, 513]
        r = run(lambda c: multi.wilson_loops_sharded(tb.wf_array, m, mesh, [-0.5, -0.5], [0, 1], c, rank, world))
        if r is None:
            lines.append({"config": "configs[3]", "error": "wilson_loops_sharded failed", "gather": gather})
            status = status or 6
        else:
            (arr, gaps), t = r
            lines.append({"config": "configs[3]: Kane-Mele wf_array([4097,513]), 513 Wilson loops (2 bands) sharded over %d GPUs" % world,
                          "n_gpus": world, "seconds_max_over_ranks_incl_gather": t, "kpts_per_s": 4096 * 512 / t,
                          "strings_per_rank": [p[2] - p[1] for p in multi.plan_strings(mesh, 0, world)],
                          "checksum": float(np.sum(np.cos(arr))), "gather": gather})
    if "E" in which:
        with contextlib.redirect_stdout(io.StringIO()):
            m = hp.cubic16(tb.tb_model)
        mesh = [side, side, side]
        r = run(lambda c: multi.mesh_phases_sharded(tb.wf_array, m, mesh, [0.0, 0.0, 0.0], list(range(8)), c, rank, world))
        if r is None:
            lines.append({"config": "configs[4]", "error": "mesh_phases_sharded failed", "gather": gather})
            status = status or 6
        else:
            (arr, gaps), t = r
            lines.append({"config": "configs[4]: cubic16 wf_array([%d]*3), solve_on_grid + berry_phase(range(8), dir=2), axis-0 slabs over %d GPUs" % (side, world),
                          "n_gpus": world, "seconds_max_over_ranks_incl_gather": t, "kpts_per_s": (side - 1) ** 3 / t,
                          "planes_per_rank": [p[2] for p in multi.plan_slabs(side, world)],
                          "gap78": float(gaps[7]), "checksum": float(np.sum(np.cos(arr))), "gather": gather})
        if hasattr(comm, "gatherv_rows_dev"):
            # the solve_all leg of configs[4]: eigenvalues of the (side-1)^3 uniform mesh, k generated per rank on the device,
            # ONE rooted rows gather-v of eval (16, nk) to rank 0 -- 268 MB sent per rank at 256^3 over 8 GPUs;
            # the other ranks hold no (16, nk) array
            msz = [side - 1] * 3
            nk = (side - 1) ** 3
            st4 = {}
            r = run(lambda c: multi.solve_all_mesh_sharded(m, msz, c, rank, world, download=False, root=0, stats=st4))
            if r is None:
                lines.append({"config": "configs[4] solve_all", "error": "solve_all_mesh_sharded failed", "gather": gather})
                status = status or 6
            else:
                ends, t = r
                ref = m.solve_all(np.array([[0.0, 0.0, 0.0], [(side - 2.0) / (side - 1.0)] * 3]))
                lines.append({"config": "configs[4] solve_all leg: cubic16 eigenvalues on k_uniform_mesh([%d]*3), k chunks over %d GPUs, ONE "
                                        "rooted rows gather-v of eval (16, nk) to rank 0" % (side - 1, world),
                              "n_gpus": world, "kpts": nk, "seconds_max_over_ranks_incl_gather": t, "kpts_per_s": nk / t,
                              "per_rank": per_rank_stats(st4),
                              "first_and_last_columns_equal_unsharded": bool(rank != 0 or np.array_equal(ends, ref)),
                              "gather": gather})
    dist.barrier()
    if hasattr(comm, "close") and not hung:
        comm.close()
    if rank == 0:
        for ln in lines:
            print(json.dumps(ln), flush=True)
    if hung or status:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(status or 3)
    if hasattr(dist, "close"):
        dist.close()
    else:
        dist.destroy_process_group()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = sys.argv[1:]

    def opt(name, default):
        return int(flags[flags.index(name) + 1]) if name in flags else default
    skip = set()
    for name in ("--reps", "--gpus", "--side"):
        if name in flags:
            skip.add(flags[flags.index(name) + 1])
    which = [a for a in args if a not in skip] or None
    gpus = opt("--gpus", int(os.environ.get("WORLD_SIZE", "1")))
    if "WORLD_SIZE" not in os.environ and gpus > 1:
        # plain `python bench_configs.py --gpus N ...`: start the N ranks (this process has made no GPU call)
        import importlib.util
        spec = importlib.util.spec_from_file_location("_tbk_launch", os.path.join(_ROOT, "pythtb_amd", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        multi_gpu(which or ["B", "D", "E"], opt("--side", 257))
    else:
        single_gpu(which or ["B", "C", "D", "E"], opt("--reps", 5))


if __name__ == "__main__":
    main()
