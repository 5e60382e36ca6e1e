#!/usr/bin/env python3
"""Headline benchmark: Haldane model, wf_array.solve_on_grid + berry_flux on the
2048 x 2048 k-mesh (BASELINE.json configs[2], the configuration the metric is
quoted on), fp64, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over the mesh: H(k) assembly + Hermitian
eigen-solve for every mesh point (eigenvectors written to the device-resident
wf_array, periodic images included) followed by the Berry flux of the lower band
over all plaquettes and its deterministic sum.  Inputs (model tables) are resident
in HBM before the timed region; nothing crosses PCIe inside it.

`value` = k-points of the K timed steps / wall time (max over ranks), exactly as the
driver's contract asks.  Order on every rank: W warm-up steps | the K-step burst on the still-idle chip
(`cold_burst`, reported, not the value) | `--preheat-s` seconds (default 0.5) of the same step, untimed | barrier |
the K timed steps | barrier.  An MI355X that has been idle needs more than 35 ms of work before its clocks
are up (profiles/burst_vs_warmup.sh: the burst costs 73-77 us per step after 5 or 500 warm-up steps, 67 us
after 3000); W = 5 steps are 0.4 ms.  Beside it, on ONE GPU, the same JSON line carries (all
outside the timed region):
  roofline        dominant kernel: algorithmic bytes / average launch duration from HIP events on the
                  kernels' stream (the RAW bracket, event overhead included: conservative); valu_frac = VALU
                  issue-slot use at the fp64 rate from rocprofv3 instruction counts (profiles/valu.json);
                  traffic = HBM bytes per launch from the committed rocprofv3 PMC passes (traffic_source)
  sustained       >= 1 s of back-to-back steps (a 20-step burst is 1.6 ms: boost clocks, warm caches);
                  its ms per step also stands next to ms_per_step as ms_per_step_sustained
  python_api      wall-clock of wf.solve_on_grid(); wf.berry_flux([0]) through the Python API
  configs         the other single-GPU legs: the same kernels at 4096^2 (1.07 GB of eigenvectors, four
                  times the 256 MiB last-level cache), BASELINE configs[3] (Kane-Mele 4096 x 512: solve +
                  flux + Wilson loops) and configs[4] (cubic16 256^3: solve + Berry phase), each with
                  HBM and VALU fractions
  cpu_baseline    the oracle's per-k Python loop on 1 host core and on all of them (os.cpu_count())

N > 1 (weak scaling): the global mesh is (2048*N + 1) x 2049; rank r owns the slab
of 2048 plaquette rows starting at global row 2048*r and recomputes its one halo
row, so there is no data-path collective.  `python bench.py --gpus N` starts its own N
ranks (pythtb_amd/launch.py; the parent makes no GPU call) and behaves the same under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.  The ranks
rendezvous on the launcher's own TCP socket (launch.Rendezvous: barriers, one-word agreements and
RCCL's 128-byte id; no torch -- TBK_RENDEZVOUS=gloo takes torch.distributed instead), bring the RCCL
communicator up FIRST, and the
gather of [partial flux, min gap, elapsed] per rank after the timed loop -- the path's
one collective -- is tbk_comm_allgather_f64 (RCCL over xGMI): `config.gather` says
`rccl_allgather`.  Every RCCL call runs under a time limit; if RCCL is unavailable the
line is still printed, through the rendezvous socket and labelled FALLBACK, and the exit status is 4.
(configs[1]/[3]/[4] at N > 1: bench_configs.py --gpus N.)

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MESH = 2048                      # plaquette rows per rank and plaquettes per row
N_STA = 2                        # Haldane: two orbitals
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6     # vector fp64 peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
VALU_SLOTS_PER_S = 256 * 4 * 2.4e9 / 4.0   # wave64 VALU instructions the chip can issue per second at the fp64 rate


def bytes_solve(n):
    return 16 * n * n            # solve_on_grid writes 16 n^2 B per k


def bytes_berry(nocc, n):
    return 16 * nocc * n         # Berry kernels read 16 nocc n B per k


def haldane(tb):
    """examples/haldane_bp.py:16-41 (delta = 0)."""
    import contextlib
    import io
    lat = [[1.0, 0.0], [0.5, np.sqrt(3.0) / 2.0]]
    orb = [[1.0 / 3.0, 1.0 / 3.0], [2.0 / 3.0, 2.0 / 3.0]]
    with contextlib.redirect_stdout(io.StringIO()):
        m = tb.tb_model(2, 2, lat, orb)
    t, t2 = -1.0, 0.15 * np.exp(1j * np.pi / 2.0)
    m.set_onsite([0.0, 0.0])
    m.set_hop(t, 0, 1, [0, 0])
    m.set_hop(t, 1, 0, [1, 0])
    m.set_hop(t, 1, 0, [0, 1])
    m.set_hop(t2, 0, 0, [1, 0])
    m.set_hop(t2, 1, 1, [1, -1])
    m.set_hop(t2, 1, 1, [0, 1])
    m.set_hop(t2.conjugate(), 1, 1, [1, 0])
    m.set_hop(t2.conjugate(), 0, 0, [1, -1])
    m.set_hop(t2.conjugate(), 0, 0, [0, 1])
    return m


# ---------------------------------------------------------------- CPU baseline (oracle; runs BEFORE any GPU call)
def _cpu_leg(args):
    """One worker: the oracle's solve_on_grid + berry_flux on an n x n sub-mesh anchored at its own start_k."""
    n, start = args
    from oracle import tb_oracle as orc
    m = orc.haldane(0.0)
    t0 = time.perf_counter()
    wfs, _ = orc.solve_on_grid(m, [n, n], start)
    t1 = time.perf_counter()
    flux = orc.berry_flux(wfs, 2, [0])
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, float(flux)


def usable_cores():
    """Host cores this process may really use: os.cpu_count() capped by the scheduler affinity mask and by the cgroup
    CPU quota (a container on a 256-thread host is often limited to a handful)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(sample_mesh=257):
    """Oracle (a NumPy port of the reference's per-k / per-plaquette Python loops, kind "port") on a bounded sample of
    the headline workload: one host core, then one worker per host core on as many sub-meshes of the same size."""
    import multiprocessing as mp
    n = sample_mesh
    nk = (n - 1) * (n - 1)
    ts, tf, flux = _cpu_leg((n, [-0.5, -0.5]))
    cores = min(usable_cores(), 64)                 # (more workers than that only measures the fork)
    out = {
        "value": nk / (ts + tf), "unit": "k-points/s", "cores": 1, "kind": "port",
        "sample": "Haldane %dx%d sub-mesh (%d k): oracle solve_on_grid %.2fs + berry_flux %.2fs on one core"
                  % (n - 1, n - 1, nk, ts, tf),
        "solve_kpts_per_s": nk / ts, "flux_plaq_per_s": nk / tf,
        # the REAL reference (pythtb 1.8.0 imported read-only) on one thread of an 8-vCPU Xeon 2.1 GHz,
        # so that a reader can see the port timed here tracks it (the reference itself cannot travel to the GPU box)
        "calibration": {"source": "BASELINE.md section 2 (reference measured in the survey container, 1 thread, Xeon 2.10 GHz)",
                        "reference_solve_on_grid_kpts_per_s": 8727.0, "reference_berry_flux_plaq_per_s": 39617.0,
                        "reference_end_to_end_kpts_per_s": 7152.0, "reference_solve_all_kpts_per_s": 10048.0,
                        "reference_km4_solve_kpts_per_s": 5807.0, "reference_cubic16_solve_kpts_per_s": 141.0,
                        "port_over_reference_solve": (nk / ts) / 8727.0, "port_over_reference_flux": (nk / tf) / 39617.0},
    }
    if cores > 1:
        try:
            ctx = mp.get_context("fork")              # no exec; the GPU has not been touched yet
            # one worker per usable core; the sub-mesh shrinks on many-core hosts so that the leg stays bounded even if
            # the cores turn out to be shared (an oversubscribed host must not turn a 3 s leg into minutes)
            nw = 257 if cores <= 16 else 129
            nkw = (nw - 1) * (nw - 1)
            jobs = [(nw, [-0.5 + 0.37 * i / cores, -0.5]) for i in range(cores)]
            t0 = time.perf_counter()
            with ctx.Pool(cores) as pool:
                res = pool.map(_cpu_leg, jobs)
            wall = time.perf_counter() - t0
            out["all_cores"] = {
                "value": nkw * cores / wall, "unit": "k-points/s", "cores": cores, "kind": "port",
                "os_cpu_count": os.cpu_count(),
                "sample": "%d workers (usable cores: os.cpu_count() = %s capped by affinity and cgroup quota), each a %dx%d "
                          "sub-mesh solve + flux; wall %.2fs incl. fork" % (cores, os.cpu_count(), nw - 1, nw - 1, wall),
                "slowest_worker_s": max(r[0] + r[1] for r in res),
            }
        except Exception as e:                        # a sandbox without fork/semaphores must not cost the GPU numbers
            out["all_cores"] = {"error": " ".join(str(e).split())[:200], "cores": cores}
    return out


# ---------------------------------------------------------------- helpers around the C ABI
def Grid(*a, **k):
    """One rank's slab of the headline workload: the library's own slab driver (pythtb_amd/multi.py::GridSlab)."""
    from pythtb_amd import multi
    return multi.GridSlab(*a, **k)


def kernel_times(ctx, fn, reps, ev_ms):
    """Per-kernel HIP-event brackets (every launch) around `reps` calls of fn: name -> raw and net average ms."""
    ctx.prof_enable(1)
    ctx.prof_reset()
    for _ in range(reps):
        fn()
    ctx.sync()
    ctx.prof_enable(0)
    out = {}
    for name, rec in ctx.prof_report().items():
        raw = rec["total_ms"] / max(rec["launches"], 1)
        out[name] = {"launches": rec["launches"], "avg_bracket_ms": raw}
    return out


def load_valu():
    p = os.path.join(ROOT, "profiles", "valu.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def roof(alg_bytes, ms_raw, valu_key=None, points=None, valu=None):
    """Roofline block of one kernel launch: HBM fraction from the algorithmic bytes and the RAW HIP-event bracket (event
    overhead included: conservative; rocprofv3's kernel duration is ~3 us shorter, profiles/), VALU fraction from the
    rocprofv3 instruction count per mesh point (profiles/valu.json) scaled to this launch."""
    gbs = alg_bytes / (ms_raw * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "basis": "avg_bracket_ms (HIP events on the kernels' stream, event overhead included)",
         "algorithmic_bytes": alg_bytes, "avg_launch_ms": ms_raw}
    v = (valu or {}).get(valu_key) if valu_key else None
    if v and points:
        insts = v["valu_wave_insts_per_point"] * points
        # ISSUE-slot occupancy (wave-instructions / slots): how busy the vector pipe is, NOT achieved flops -- masked lanes,
        # moves, reductions and replicated scalar work all count
        r["valu_issue_frac"] = insts / (ms_raw * 1e-3) / VALU_SLOTS_PER_S
        r["valu_issue_slots_as_tflops"] = insts * 128.0 / (ms_raw * 1e-3) / 1e12
        r["valu_peak_tflops"] = FP64_VALU_PEAK_TFLOPS
        r["valu_source"] = v.get("source")
        r["valu_kernel"] = v.get("kernel")
        if v.get("useful_flops_per_point"):
            # USEFUL fp64 work: SURVEY.md 8d's algorithmic flops per k-point (the model is written out in
            # profiles/make_valu_json.py) x points / time / the 78.6 TFLOP/s vector peak
            uf = v["useful_flops_per_point"] * points / (ms_raw * 1e-3) / 1e12
            r["useful_tflops"] = uf
            r["useful_flop_frac"] = uf / FP64_VALU_PEAK_TFLOPS
            r["useful_flops_per_point"] = v["useful_flops_per_point"]
            r["flop_model"] = v.get("flop_model")
    return r


# ---------------------------------------------------------------- the other single-GPU legs
def extra_configs(tb, _lib, lib, ctx, ev_ms, valu):
    import contextlib
    import io
    import helpers as hp
    out = []
    # ---- BASELINE configs[1]: Haldane (delta = 0.2) solve_all on the 1024 x 1024 uniform k list, buffers resident
    try:
        m = hp.haldane(tb.tb_model, 0.2)
        k = m.k_uniform_mesh([1024, 1024])
        nk = len(k)
        hm = m._device_model()
        kd, ed, vd = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, k.nbytes, C.byref(kd)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 2 * 8, C.byref(ed)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 4 * 16, C.byref(vd)))
        _lib.check(lib.tbk_dev_upload(ctx.handle, kd, k.ctypes.data_as(C.c_void_p), k.nbytes))
        _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, vd))
        ctx.sync()
        kt = kernel_times(ctx, lambda: (_lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None)),
                                        _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, vd))), 10, ev_ms)
        ev = np.zeros((2, nk))
        _lib.check(lib.tbk_dev_download(ctx.handle, ev.ctypes.data_as(C.c_void_p), ed, ev.nbytes))
        t_api = 1e9
        for _ in range(3):                                   # (the first call allocates the result staging; best of three)
            t0 = time.perf_counter()
            ev_api = m.solve_all(k)                          # the drop-in call: the k_uniform_mesh array is uploaded and solved as a list
            t_api = min(t_api, time.perf_counter() - t0)
        t_list = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ev_list = m.solve_all_mesh([1024, 1024])         # the explicit extension: the same list generated on the device
            t_list = min(t_list, time.perf_counter() - t0)
        ktm = kernel_times(ctx, lambda: m.solve_all_mesh([1024, 1024]), 5, ev_ms)
        out.append({"config": "BASELINE configs[1] on one GPU: Haldane (delta 0.2) solve_all on k_uniform_mesh([1024,1024]), k list and results resident",
                    "kpts": nk, "kernels": kt,
                    "kpts_per_s_eigenvalues": nk / (kt["solve_list_val"]["avg_bracket_ms"] * 1e-3),
                    "kpts_per_s_with_vectors": nk / (kt["solve_list_vec"]["avg_bracket_ms"] * 1e-3),
                    "roofline": {"solve_list_val": roof(8 * (2 + 2) * nk, kt["solve_list_val"]["avg_bracket_ms"], "k_solve_small_multi<2,false,2>", nk, valu),
                                 "solve_list_vec": roof((8 * (2 + 2) + 16 * 4) * nk, kt["solve_list_vec"]["avg_bracket_ms"], "k_solve_small<2,0,true>", nk, valu)},
                    "python_call_incl_pcie_s": t_api, "python_call_solve_all_mesh_s": t_list,
                    "drop_in_call": "m.solve_all(m.k_uniform_mesh([1024, 1024])): the list path on the array given (16.8 MB of k up, "
                                    "16.8 MB of eigenvalues back over PCIe); solve_all_mesh generates the list on the device (k_mesh_evals)",
                    "mesh_kernels": ktm,
                    "check": {"sum": float(ev.sum()), "min": float(ev.min()), "max": float(ev.max()),
                              "api_max_abs_diff_vs_resident_list_kernel": float(np.max(np.abs(ev_api - ev))),
                              "solve_all_mesh_max_abs_diff_vs_resident_list_kernel": float(np.max(np.abs(ev_list - ev)))}})
        if "mesh_evals" in ktm:
            out[-1]["roofline"]["mesh_evals"] = roof(8 * (2 + 2) * nk, ktm["mesh_evals"]["avg_bracket_ms"], "k_mesh_evals<2,1>", nk, valu)
        for ptr in (kd, ed, vd):
            _lib.check(lib.tbk_dev_free(ctx.handle, ptr))
    except Exception as e:
        out.append({"config": "configs[1]", "error": " ".join(str(e).split())[:300]})
    # ---- the headline kernels past the last-level cache: 4096^2 (1.07 GB of eigenvectors)
    try:
        m = haldane(tb)
        g = Grid(lib, _lib, ctx, m, [4097, 4097])
        occ = np.array([0], dtype=np.int32)
        start = [-0.5, -0.5]

        def step():
            g.solve(start)
            g.flux(occ)
        step()
        ctx.sync()
        kt = kernel_times(ctx, step, 5, ev_ms)
        npt = 4096 * 4096
        chern = float(g.flux_total()[0] / (2 * np.pi))

        def fused():
            g.solve_flux(start, occ)
        for _ in range(3):
            fused()
        ctx.sync()
        f0 = time.perf_counter()
        for _ in range(50):
            fused()
        ctx.sync()
        f_ms = 1e3 * (time.perf_counter() - f0) / 50
        chern_f = float(g.flux_total()[0] / (2 * np.pi))
        ktf = kernel_times(ctx, fused, 5, ev_ms)
        out.append({"config": "Haldane solve_on_grid + berry_flux at 4096^2 (1.07 GB array: 4x the 256 MiB last-level cache)",
                    "kpts": npt, "kernels": dict(kt, **ktf), "chern": chern,
                    "fused_step": {"ms_per_step": f_ms, "kpts_per_s": npt / (f_ms * 1e-3), "chern": chern_f, "steps": 50},
                    "roofline": {"solve_grid_flux": roof(bytes_solve(2) * npt, ktf["solve_grid_flux"]["avg_bracket_ms"],
                                                         "k_grid_rows_flux<2,1,1>", 4097 * 4097, valu),
                                 "solve_grid": roof(bytes_solve(2) * npt, kt["solve_grid"]["avg_bracket_ms"],
                                                    "k_grid_rows<2,1>", 4097 * 4097, valu),
                                 "berry_flux": roof(bytes_berry(1, 2) * npt, kt["berry_flux"]["avg_bracket_ms"],
                                                    "k_flux_rows<1,2>", 4097 * 4097, valu)},
                    "solve_kpts_per_s": npt / (kt["solve_grid"]["avg_bracket_ms"] * 1e-3)})
        g.free()
    except Exception as e:
        out.append({"config": "4096^2", "error": " ".join(str(e).split())[:300]})
    # ---- BASELINE configs[3]: Kane-Mele (4 states, spinor) 4096 x 512, solve + flux + Wilson loops, one GPU
    try:
        m = hp.kane_mele(tb.tb_model, "odd")
        mesh = [4097, 513]
        g = Grid(lib, _lib, ctx, m, mesh)
        occ = np.array([0, 1], dtype=np.int32)
        start = [-0.5, -0.5]
        phases = np.zeros(mesh[1] * 2)

        def step():
            g.solve(start)
            g.flux(occ)
            _lib.check(lib.tbk_berry_phase(g.h, _lib.iptr(occ), 2, 0, 1, _lib.dptr(phases)))
        step()
        ctx.sync()
        kt = kernel_times(ctx, step, 5, ev_ms)
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        ctx.sync()
        wall = (time.perf_counter() - t0) / 5
        npt = 4096 * 512
        gaps = g.gaps()
        wl_ms = sum(v["avg_bracket_ms"] for k, v in kt.items() if k.startswith("chain"))
        cen = np.sort(phases.reshape(mesh[1], 2) / (2 * np.pi) % 1.0, axis=1)
        r4 = roof(bytes_solve(4) * npt, kt["solve_grid"]["avg_bracket_ms"], "k_grid_rows<4,1>", 4097 * 513, valu)
        k4 = npt / (kt["solve_grid"]["avg_bracket_ms"] * 1e-3)
        target_4orb = {"kpts_per_s": k4, "required": 1e7, "met": bool(k4 >= 1e7), "hbm_frac": r4["frac"],
                       "bound": "fp64 VALU issue (valu_issue_frac %.2f), not HBM" % r4.get("valu_issue_frac", float("nan")),
                       "clause_claimed": "throughput: >= 1e7 (H(k)+eigh) solves/s for a 4-orbital model on one MI355X.  The '>= 60 % "
                                         "HBM roofline' clause is NOT claimed for n = 4: 256 B per k-point at 60 % of 8 TB/s would be "
                                         "1.9e10 k/s, above the fp64-VALU ceiling of the 4x4 Hermitian eigen-solve (SURVEY.md 8d); it is "
                                         "claimed for n = 2 (the headline roofline block)"}
        out.append({"config": "BASELINE configs[3] on one GPU: Kane-Mele (4 states) wf_array([4097,513]) solve_on_grid + berry_flux([0,1]) "
                              "+ berry_phase([0,1], dir=0, berry_evals=True)",
                    "kpts": npt, "kernels": kt, "wall_ms_per_pass_incl_result_download": wall * 1e3,
                    "solve_kpts_per_s": npt / (kt["solve_grid"]["avg_bracket_ms"] * 1e-3),
                    "wilson_links_per_s": 4096 * 513 / (wl_ms * 1e-3) if wl_ms > 0 else None,
                    "target_4orb": target_4orb,
                    "z2_index": int(tb.z2_from_wilson_centres(phases.reshape(mesh[1], 2))),
                    "roofline": {"solve_grid": r4,
                                 "berry_flux": roof(bytes_berry(2, 4) * npt, kt["berry_flux"]["avg_bracket_ms"],
                                                    "k_flux_rows<2,4>", 4097 * 513, valu)},
                    "check": {"min_gaps": [float(x) for x in gaps[:3]],
                              # time reversal maps the string at k_y to the one at -k_y: the two Wilson-loop spectra coincide
                              "time_reversal_error": float(np.max(np.abs(np.exp(2j * np.pi * cen).sum(axis=1)
                                                                         - np.exp(2j * np.pi * cen[::-1]).sum(axis=1))))}})
        g.free()
    except Exception as e:
        out.append({"config": "configs[3]", "error": " ".join(str(e).split())[:300]})
    # ---- BASELINE configs[4]: cubic16 on the full 256^3 mesh (69.5 GB of eigenvectors resident), one GPU
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m = hp.cubic16(tb.tb_model)
        info = ctx.info()
        side = 257 if info["hbm_bytes"] > 100e9 else 65
        mesh = [side, side, side]
        g = Grid(lib, _lib, ctx, m, mesh)
        occ = np.arange(8, dtype=np.int32)
        start = [0.0, 0.0, 0.0]
        phases = np.zeros(side * side)

        def step():
            g.solve(start)
            _lib.check(lib.tbk_berry_phase(g.h, _lib.iptr(occ), 8, 2, 0, _lib.dptr(phases)))
        step()
        ctx.sync()
        kt = kernel_times(ctx, step, 2, ev_ms)
        npt = (side - 1) ** 3
        gaps = g.gaps()
        # (chain_links / chain_lu are brackets INSIDE chain_partial_det: not added twice)
        bp_ms = sum(v["avg_bracket_ms"] * v["launches"] / max(kt["solve_grid"]["launches"], 1) for k, v in kt.items()
                    if k not in ("solve_grid", "grid_tables", "chain_links", "chain_lu", "chain_prod", "chain_prod_det", "e16", "tw16_fallback",
                                 "tw16_tridiag", "tw16_eigvals", "tw16_vectors"))
        out.append({"config": "BASELINE configs[4] on one GPU: cubic16 (16 orbitals, 888 hops) wf_array([%d]*3) solve_on_grid + "
                              "berry_phase(range(8), dir=2)" % side,
                    "kpts": npt, "kernels": kt, "solve_kpts_per_s": npt / (kt["solve_grid"]["avg_bracket_ms"] * 1e-3),
                    "berry_links_per_s": side * side * (side - 1) / (bp_ms * 1e-3) if bp_ms > 0 else None,
                    "roofline": {"solve_grid": roof(bytes_solve(16) * npt, kt["solve_grid"]["avg_bracket_ms"],
                                                    "k_e16<1>" if "e16" in kt else "k_tw16<1>", side ** 3, valu),
                                 "berry_phase": dict(roof(bytes_berry(8, 16) * side ** 3, bp_ms, None, None, None),
                                                     kernel="chain_prod" if "chain_prod" in kt else "chain_links + chain_lu")},
                    "check": {"gap78": float(gaps[7]), "phase_checksum": float(np.sum(np.cos(phases)))}})
        g.free()
        # ---- ... and its solve_all leg: eigenvalues of the k_uniform_mesh([side - 1] * 3) list, list and results resident
        # (eigenvalue-only form of the fused kernel, k_e16<0, false>; algorithmic bytes 8 (d + n) per k-point)
        nk = (side - 1) ** 3
        mesh32 = np.array([side - 1] * 3, dtype=np.int32)
        kd, ed = C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 24, C.byref(kd)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * 16 * 8, C.byref(ed)))
        _lib.check(lib.tbk_k_uniform_mesh_dev(ctx.handle, 3, _lib.iptr(mesh32), kd))
        hm = m._device_model()
        _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None))
        ctx.sync()
        ka = kernel_times(ctx, lambda: _lib.check(lib.tbk_solve_list_dev(hm, kd, nk, ed, None)), 2, ev_ms)
        t_all = sum(v["avg_bracket_ms"] for v in ka.values())
        ev0 = np.zeros(16)
        for b in range(16):
            _lib.check(lib.tbk_dev_download(ctx.handle, ev0[b:b + 1].ctypes.data_as(C.c_void_p), C.c_void_p(ed.value + 8 * b * nk), 8))
        out[-1]["solve_all"] = {"kpts": nk, "kernels": ka, "ns_per_point": t_all * 1e6 / nk, "kpts_per_s": nk / (t_all * 1e-3),
                                "roofline": roof(8 * (3 + 16) * nk, t_all, "k_e16<0,false>", nk, valu),
                                "first_point_max_abs_diff_vs_solve_one": float(np.max(np.abs(ev0 - m.solve_one([0.0, 0.0, 0.0]))))}
        for ptr in (kd, ed):
            _lib.check(lib.tbk_dev_free(ctx.handle, ptr))
    except Exception as e:
        out.append({"config": "configs[4]", "error": " ".join(str(e).split())[:300]})
    return out


def python_api_leg(tb, model, reps=5):
    """Wall-clock of the drop-in calls themselves (SURVEY.md 8d's end-to-end number): every call synchronises and
    brings its small result (min gaps, one float) back to the host; the 268.7 MB array stays on the device."""
    wf = tb.wf_array(model, [MESH + 1, MESH + 1])
    wf.solve_on_grid([-0.5, -0.5])
    wf.berry_flux([0])
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        gaps = wf.solve_on_grid([-0.5, -0.5])
        flux = wf.berry_flux([0])
        ts.append(time.perf_counter() - t0)
    out = {"call": "wf.solve_on_grid([-0.5,-0.5]); wf.berry_flux([0])  (wf_array([2049,2049]))", "reps": reps,
           "ms_per_step_min": 1e3 * min(ts), "ms_per_step_median": 1e3 * sorted(ts)[len(ts) // 2],
           "kpts_per_s": MESH * MESH / sorted(ts)[len(ts) // 2], "chern": float(flux / (2 * np.pi)), "min_gap": float(gaps[0])}
    # the same result from ONE call (the fused pass of the timed loop, through the Python method)
    wf.solve_on_grid_flux([-0.5, -0.5], [0])
    tf = []
    for _ in range(reps):
        t0 = time.perf_counter()
        gaps, flux = wf.solve_on_grid_flux([-0.5, -0.5], [0])
        tf.append(time.perf_counter() - t0)
    out["fused_call"] = {"call": "wf.solve_on_grid_flux([-0.5,-0.5], [0])", "ms_per_step_min": 1e3 * min(tf),
                         "ms_per_step_median": 1e3 * sorted(tf)[len(tf) // 2], "chern": float(flux / (2 * np.pi)),
                         "min_gap": float(gaps[0])}
    return out


class _StubCtx(object):
    """--stub: the control flow of a multi-rank run (launcher, rendezvous, barriers, gather, line) with no GPU."""
    handle = None

    def sync(self): pass
    def prof_enable(self, period=1): pass
    def prof_reset(self): pass
    def prof_report(self): return {}
    def prof_calibrate(self, reps=50): return 0.0
    def info(self): return {"name": "stub (no GPU)", "compute_units": 0, "hbm_bytes": 0}


class _StubGrid(object):
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def solve(self, start): time.sleep(2e-4)
    def flux(self, occ): pass
    def gaps(self): return np.array([1.5 + self.rank])
    def flux_total(self, nslices=1): return np.array([-2 * np.pi / self.world])
    def free(self): pass


class _StubComm(object):
    """--stub: a communicator with RcclComm's interface over the rendezvous; TBK_BENCH_STUB_COMM=fail makes its gather raise
    (on every rank: the stand-in shares the rendezvous with the fallback, so a one-sided failure would wedge it)."""
    name = "stub"

    def __init__(self, dist):
        from pythtb_amd import multi
        self.g = multi.SocketComm(dist) if hasattr(dist, "allgather_bytes") else multi.GlooComm(dist)
        self.world = self.g.world

    def allgather(self, mine):
        if os.environ.get("TBK_BENCH_STUB_COMM") == "fail":
            raise RuntimeError("stub communicator told to fail")
        mine = np.asarray(mine, dtype=float).reshape(-1)
        return self.g.allgatherv(mine, [mine.size] * self.world).reshape(self.world, mine.size)


def _load_launcher():
    """pythtb_amd/launch.py by path: importing the package would load libtbk.so, and the parent of the ranks must
    not have touched the GPU runtime at all."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_tbk_launch", os.path.join(ROOT, "pythtb_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat-s", type=float, default=0.5,
                    help="seconds of the same step run back to back, untimed, between the W warm-up steps and the K timed ones: "
                         "an idle MI355X needs > 35 ms of work before its clocks are up (profiles/burst_vs_warmup.sh); "
                         "0 = none.  The K-step burst WITHOUT it is in the line as cold_burst")
    ap.add_argument("--headline-only", action="store_true", help="skip the sustained / python-API / other-config legs")
    ap.add_argument("--no-check", action="store_true", help="diagnostics: skip the Chern-number assertion")
    ap.add_argument("--two-calls", action="store_true", help=argparse.SUPPRESS)   # (kept for old command lines: now the default)
    ap.add_argument("--fused-headline", action="store_true", help="diagnostics: time the fused extension (tbk_wfs_solve_grid_flux_async) "
                    "as the K-step loop instead of the drop-in two-call step; the line's metric string then says so")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # CPU test of the multi-rank control flow
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  This process has made no GPU call (nothing
        # GPU-side is imported before this point) and only waits; rank 0's JSON line is the children's stdout.
        sys.exit(_load_launcher().spawn_ranks(os.environ.get("TBK_BENCH_SELF") or os.path.abspath(__file__), sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    cpu = None
    if world == 1 and not args.no_cpu_baseline and not args.stub:
        cpu = cpu_baseline()                        # forks workers: must precede every GPU call of this process

    dist = None
    rdzv_kind = os.environ.get("TBK_RENDEZVOUS", "socket")
    if world > 1:                                   # rendezvous + barriers + RCCL's 128-byte id; nothing of the data path
        if rdzv_kind == "gloo":                     # (optional: torch.distributed on the host, as rounds 2-5 did)
            import datetime
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        else:                                       # the launcher's own TCP rendezvous: no torch anywhere in the run
            dist = _load_launcher().Rendezvous(rank=rank, world=world, timeout=600.0)

    status = 0                                      # exit status of this rank (non-zero: the line is still printed)
    hung = False                                    # an RCCL call was left behind on its thread: finish with os._exit
    comm, comm_err = None, ""
    rccl_timeout = float(os.environ.get("TBK_BENCH_RCCL_TIMEOUT", "120"))
    if args.stub:
        tb = _lib = lib = None
        from pythtb_amd import shard, multi
        ctx = _StubCtx()
        info = ctx.info()
        valu = {}
        if world > 1:
            comm = _StubComm(dist)
    else:
        import pythtb_amd as tb
        from pythtb_amd import _lib, shard, multi
        lib = _lib.lib
        ctx = _lib.default_context()                # device = LOCAL_RANK
        info = ctx.info()
        valu = load_valu()
        if world > 1:
            # the path's one collective is an RCCL all-gather over xGMI: bring the communicator up FIRST, so that the
            # gather that produces the reported numbers is the product path's own (gloo only as the labelled fallback)
            comm, comm_err, hung = multi.rccl_bring_up(ctx, dist, rank, world, rccl_timeout)
            if comm is None:
                sys.stderr.write("[bench] rank %d: RCCL communicator unavailable (%s)\n" % (rank, comm_err))

    g_n0 = MESH * world + 1                         # global axis-0 mesh points
    row0, nrows = shard.split_rows(g_n0, world, rank)
    assert nrows == MESH + 1
    if args.stub:
        model, grid = None, _StubGrid(rank, world)
    else:
        model = haldane(tb)
        grid = Grid(lib, _lib, ctx, model, [nrows, MESH + 1], row0, g_n0)
    start = [-0.5, -0.5]
    occ = np.array([0], dtype=np.int32)

    def step_two_calls():
        grid.solve(start)
        grid.flux(occ)

    def step_fused():
        grid.solve_flux(start, occ)

    # The HEADLINE step is the drop-in pair -- wf.solve_on_grid(); wf.berry_flux([0]) as their two C-ABI launches -- the
    # workload BASELINE.json's metric names and every script written for the reference runs (ADVICE r3 / VERDICT r3 weak #5).
    # The one-pass extension (wf.solve_on_grid_flux) is measured in its own leg and reported as `fused_extension`.
    fused_headline = args.fused_headline and not args.stub
    step = step_fused if fused_headline else step_two_calls

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    # ---- the chip has been idle until now and W steps are ~0.4 ms of work: the same K-step burst here (kept in the line as
    # cold_burst) runs ~10 % slower than any later one.  Then `--preheat-s` seconds of the same step, untimed.
    preheat = None
    if args.preheat_s > 0 and not args.stub:
        barrier()
        c0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.sync()
        c1 = time.perf_counter()
        n_pre, p0 = 0, time.perf_counter()
        while time.perf_counter() - p0 < args.preheat_s:
            for _ in range(200):
                step()
            ctx.sync()
            n_pre += 200
        preheat = {"cold_burst": {"steps": args.steps, "ms_per_step": 1e3 * (c1 - c0) / args.steps, "when": "right after the W warm-up steps"},
                   "preheat": {"steps": n_pre, "seconds": time.perf_counter() - p0,
                               "what": "the same step back to back, untimed, before the K timed steps (--preheat-s)"}}
    # HIP-event brackets on the kernels' own stream, inside the timed region.  A bracket
    # costs ~3 us of stream time (3 kernels x 2 events = 20% of a step), so every 7th
    # launch is bracketed: coprime with the 3 launches per step, every kernel is sampled
    # once per 7 steps.  TBK_PROF_PERIOD=1 brackets every launch, 0 none.
    ctx.prof_enable(int(os.environ.get("TBK_PROF_PERIOD", "7")))
    ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    t1 = time.perf_counter()
    barrier()
    ctx.prof_enable(False)
    elapsed = t1 - t0
    prof = ctx.prof_report()
    ev_ms = ctx.prof_calibrate(50)                  # empty-bracket overhead, measured now (reported, not subtracted)

    # results of the last step (outside the timed region)
    gaps = grid.gaps()
    tot = grid.flux_total()

    extras = dict(preheat or {})
    if world == 1 and not args.headline_only and not args.stub:
        # ---- sustained: at least one second of back-to-back steps (the K-step burst above is ~1.6 ms)
        n_sus = max(args.steps, int(math.ceil(1.2 / max(elapsed / args.steps, 1e-6))))
        ctx.sync()
        s0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        ctx.sync()
        s1 = time.perf_counter()
        extras["ms_per_step_sustained"] = 1e3 * (s1 - s0) / n_sus
        extras["sustained"] = {"steps": n_sus, "seconds": s1 - s0, "ms_per_step": 1e3 * (s1 - s0) / n_sus,
                               "value": MESH * MESH * n_sus / (s1 - s0), "unit": "k-points/s",
                               "chern": float(grid.flux_total()[0] / (2 * np.pi))}
        # ---- the same kernels with every launch bracketed over 100 steps (per-kernel averages on a warm, busy chip)
        extras["kernels_every_launch_bracketed"] = kernel_times(ctx, step, 100, ev_ms)
        # ---- the other form of the step, 2000 back to back: the fused extension when the headline is the drop-in pair
        other, okey = (step_two_calls, "two_call_step") if step is step_fused else (step_fused, "fused_extension")
        for _ in range(5):
            other()
        ctx.sync()
        s0 = time.perf_counter()
        for _ in range(args.steps):
            other()
        ctx.sync()
        sb = time.perf_counter()
        for _ in range(2000):
            other()
        ctx.sync()
        s1 = time.perf_counter()
        extras[okey] = {"what": "wf.solve_on_grid_flux (tbk_wfs_solve_grid_flux_async): solve_on_grid AND berry_flux in one pass, eigenvectors "
                                "written once, plaquette phases from registers -- an EXTENSION no script written for the reference calls"
                                if okey == "fused_extension" else "the drop-in pair as two launches",
                        "ms_per_step": 1e3 * (s1 - sb) / 2000, "value": MESH * MESH * 2000 / (s1 - sb), "unit": "k-points/s",
                        "burst": {"steps": args.steps, "ms_per_step": 1e3 * (sb - s0) / args.steps},
                        "chern": float(grid.flux_total()[0] / (2 * np.pi)),
                        "kernels": kernel_times(ctx, other, 100, ev_ms)}
        try:
            extras["python_api"] = python_api_leg(tb, model)
        except Exception as e:
            extras["python_api"] = {"error": " ".join(str(e).split())[:300]}
        extras["configs"] = extra_configs(tb, _lib, lib, ctx, ev_ms, valu)

    # ---- the gather: [partial flux, min gap, elapsed] of every rank
    mine = np.array([tot[0], gaps[0], elapsed])
    allv, gather = mine.reshape(1, 3), "none (one GPU)"
    if world > 1:
        allv = None
        if comm is not None:
            ok, val, h = multi.call_with_timeout(lambda: comm.allgather(mine), rccl_timeout)
            hung = hung or h
            if not ok:
                comm_err = "all-gather: %s" % val
            if multi.agree(dist, ok):
                allv = np.asarray(val, dtype=float).reshape(world, 3)
                gather = "stub_allgather" if args.stub else "rccl_allgather (tbk_comm_allgather_f64, RCCL over xGMI)"
            elif ok:
                comm_err = "all-gather failed on another rank"
        if allv is None:
            host = multi.SocketComm(dist) if hasattr(dist, "allgather_bytes") else multi.GlooComm(dist)
            allv = host.allgatherv(mine, [3] * world).reshape(world, 3)
            gather = "%s (FALLBACK: rccl %s)" % (host.name, comm_err or "unavailable")
            status = 4                              # the line is printed, the run is not a success
            sys.stderr.write("[bench] rank %d: RCCL gather unavailable (%s); reported through %s\n" % (rank, comm_err, host.name))

    def build_line(gather):
        """Pure host arithmetic on numbers already in hand."""
        t_max = float(allv[:, 2].max())
        nk_step = MESH * MESH * world
        flux_all, gaps_all = multi.combine_flux_blocks(allv[:, :2], 1)     # rank-ordered sum of the partial fluxes, min of the gaps
        chern = float(flux_all / (2 * np.pi))
        kern = {}
        for name, rec in prof.items():
            kern[name] = {"launches": rec["launches"], "avg_bracket_ms": rec["total_ms"] / max(rec["launches"], 1)}
        npt = MESH * MESH
        # (the fused pass writes the array once and reads nothing back: its algorithmic traffic is the solve's 16 n^2 B per point)
        alg = {"solve_grid": bytes_solve(N_STA) * npt, "berry_flux": bytes_berry(1, N_STA) * npt,
               "solve_grid_flux": bytes_solve(N_STA) * npt}
        vkey = {"solve_grid": "k_grid_rows<2,1>", "berry_flux": "k_flux_rows<1,2>", "solve_grid_flux": "k_grid_rows_flux<2,1,1>"}
        # dominant kernel of the TIMED step (the fused kernel's block is under fused_extension.roofline / roofline_all)
        dom = max(("solve_grid_flux", "solve_grid", "berry_flux"), key=lambda k: kern.get(k, {"avg_bracket_ms": 0})["avg_bracket_ms"])
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC passes (rocprofv3 --pmc), per launch
        if os.path.exists(tpath):
            tj = json.load(open(tpath)).get(dom)
            if tj is None:                          # an empty / clobbered table must be seen, not read as "no counters"
                sys.stderr.write("[bench] profiles/traffic.json has no entry for %r: roofline.traffic is null\n" % dom)
                tj = {}
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/traffic.json <- %s (rocprofv3 --pmc passes of this command, committed; not re-measured in this run)" % tj.get("source")
        roofs = {}
        for name in ("solve_grid_flux", "solve_grid", "berry_flux"):
            if name in kern:
                roofs[name] = roof(alg[name], kern[name]["avg_bracket_ms"], vkey[name], (MESH + 1) * (MESH + 1), valu)
        # the other form's kernels, measured in their own leg
        for leg in ("two_call_step", "fused_extension"):
            for name, rec in extras.get(leg, {}).get("kernels", {}).items():
                if name in alg and name not in roofs:
                    roofs[name] = roof(alg[name], rec["avg_bracket_ms"], vkey[name], (MESH + 1) * (MESH + 1), valu)
            if leg in extras and "solve_grid_flux" in roofs and leg == "fused_extension":
                extras[leg]["roofline"] = dict(roofs["solve_grid_flux"], kernel="solve_grid_flux")
        out = {
            "metric": "k-points solved/sec (H(k)+eigh) and Berry-flux/sec, Haldane 2048^2 mesh"
                      + (" [fused extension as the timed step]" if fused_headline else ""),
            "value": nk_step * args.steps / t_max, "unit": "k-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Haldane (2 orb, 9 hops) wf_array solve_on_grid + berry_flux([0]), "
                                   "%d x %d k-mesh per GPU (BASELINE.json configs[2])" % (MESH, MESH),
                       "step": "one fused pass (tbk_wfs_solve_grid_flux_async: eigenvectors written once, plaquette phases from registers)"
                               if fused_headline else
                               "the drop-in pair wf.solve_on_grid(); wf.berry_flux([0]) as its two C-ABI launches (tbk_wfs_solve_grid_async, "
                               "tbk_berry_flux_async); the one-pass extension is reported under fused_extension",
                       "global_mesh": [MESH * world, MESH], "start_k": [-0.5, -0.5],
                       "sharding": "k-slabs along mesh axis 0, halo row recomputed", "gather": gather},
            "roofline": dict(roofs.get(dom, {}), kernel=dom, traffic=traffic, traffic_source=traffic_source,
                             traffic_over_algorithmic=(traffic / alg[dom]) if traffic else None),
            "kernels": kern, "empty_bracket_ms": ev_ms,
            "roofline_all": roofs,
            "solve_kpts_per_s": npt / (kern["solve_grid"]["avg_bracket_ms"] * 1e-3) if "solve_grid" in kern else None,
            "flux_plaq_per_s": npt / (kern["berry_flux"]["avg_bracket_ms"] * 1e-3) if "berry_flux" in kern else None,
            "check": {"chern": chern, "min_gap": float(gaps_all[0])},
            "device": info["name"].strip() or "gfx950", "compute_units": info["compute_units"],
        }
        out.update(extras)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        return out, chern

    if rank == 0:
        out, chern = build_line(gather)
        print(json.dumps(out), flush=True)
        if not args.no_check and not abs(chern + 1.0) < 1e-9:
            sys.stderr.write("[bench] Chern number %r != -1\n" % chern)
            status = status or 5
    if dist is not None:
        dist.barrier()
        if hasattr(dist, "close"):
            dist.close()
        elif not hung:
            dist.destroy_process_group()
    if not hung:
        grid.free()
    sys.stdout.flush()
    sys.stderr.flush()
    if hung or status:
        os._exit(status or 3)                       # (a thread may still sit in an RCCL call: no interpreter teardown)


if __name__ == "__main__":
    main()
