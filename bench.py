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

N > 1 (weak scaling): the global mesh is (2048*N + 1) x 2049; rank r owns the slab
of 2048 plaquette rows starting at global row 2048*r and recomputes its one halo
row, so there is no data-path collective.  The only exchange is the gather of
[partial flux, min gap, elapsed] per rank after the timed loop: through gloo for the
reported line, and once more through the RCCL all-gather (tbk_comm_*) as a check of
that path, under a watchdog so that a communicator that never comes up cannot cost
the measurement.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MESH = 2048                      # plaquette rows per rank and plaquettes per row
N_STA = 2                        # Haldane: two orbitals
BYTES_SOLVE_PER_K = 16 * N_STA * N_STA        # solve_on_grid writes 16 n^2 B per k
BYTES_FLUX_PER_K = 16 * 1 * N_STA             # berry_flux reads 16 nocc n B per k (nocc = 1)
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s spec


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


def cpu_baseline(sample_mesh=513):
    """Oracle (a NumPy port of the reference's per-k / per-plaquette Python loops)
    timed on one host core on a bounded sample of the same workload."""
    from oracle import tb_oracle as orc
    m = orc.haldane(0.0)
    n = sample_mesh
    t0 = time.perf_counter()
    wfs, _ = orc.solve_on_grid(m, [n, n], [-0.5, -0.5])
    t1 = time.perf_counter()
    flux = orc.berry_flux(wfs, 2, [0])
    t2 = time.perf_counter()
    nk = (n - 1) * (n - 1)
    return {
        "value": nk / (t2 - t0), "unit": "k-points/s", "cores": 1, "kind": "port",
        "sample": "Haldane %dx%d sub-mesh (%d k): oracle solve_on_grid %.2fs + berry_flux %.2fs, Chern %.6f"
                  % (n - 1, n - 1, nk, t1 - t0, t2 - t1, flux / (2 * np.pi)),
        "solve_kpts_per_s": nk / (t1 - t0), "flux_plaq_per_s": nk / (t2 - t1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="diagnostics: skip the Chern-number assertion")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    dist = None
    if world > 1:                                   # rendezvous + barriers only; no tensors on the GPU
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import pythtb_amd as tb
    from pythtb_amd import _lib, shard
    lib = _lib.lib
    ctx = _lib.default_context()                    # device = LOCAL_RANK
    info = ctx.info()

    model = haldane(tb)
    hmodel = model._device_model()
    g_n0 = MESH * world + 1                         # global axis-0 mesh points
    row0, nrows = shard.split_rows(g_n0, world, rank)
    assert nrows == MESH + 1
    mesh = np.array([nrows, MESH + 1], dtype=np.int32)
    hw = C.c_void_p()
    _lib.check(lib.tbk_wfs_create(ctx.handle, 2, _lib.iptr(mesh), N_STA, N_STA, C.byref(hw)))
    start = np.array([-0.5, -0.5])
    pbc = np.ascontiguousarray(np.exp(-2j * np.pi * model._orb[:, model._per].T))     # [dim][orb]
    occ = np.array([0], dtype=np.int32)

    def step():
        _lib.check(lib.tbk_wfs_solve_grid_async(hw, hmodel, _lib.dptr(start), _lib.dptr(pbc.view(float)), row0, g_n0))
        _lib.check(lib.tbk_berry_flux_async(hw, _lib.iptr(occ), 1, 0, 1, 0))

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
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

    # results of the last step (outside the timed region)
    gaps = np.zeros(1)
    tot = np.zeros(1)
    _lib.check(lib.tbk_wfs_solve_grid_result(hw, _lib.dptr(gaps)))
    _lib.check(lib.tbk_berry_flux_result(hw, _lib.dptr(tot), None))

    gather = "none"
    allv = np.array([[tot[0], gaps[0], elapsed]])
    if world > 1:
        import torch
        mine = np.array([tot[0], gaps[0], elapsed])

        # results first through gloo (3 doubles per rank, outside the timed region): the line below must not
        # depend on anything that can hang
        buf = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(buf, torch.from_numpy(mine))
        allv = np.stack([b.numpy() for b in buf])
        gather = "gloo"

    def report(gather):
        t_max = float(allv[:, 2].max())
        nk_step = MESH * MESH * world
        chern = float(allv[:, 0].sum() / (2 * np.pi))
        # a bracket = [event][kernel][event]; an empty bracket measured on the same stream gives
        # the events' own share, which is subtracted (both numbers are reported)
        ev_ms = ctx.prof_calibrate(50)
        kern = {}
        for name, rec in prof.items():
            raw = rec["total_ms"] / max(rec["launches"], 1)
            kern[name] = {"launches": rec["launches"], "avg_ms": max(raw - ev_ms, 0.25 * raw), "avg_bracket_ms": raw}
        alg = {"solve_grid": BYTES_SOLVE_PER_K * MESH * MESH, "berry_flux": BYTES_FLUX_PER_K * MESH * MESH}
        dom = max(("solve_grid", "berry_flux"), key=lambda k: kern.get(k, {"avg_ms": 0})["avg_ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC passes (rocprofv3 --pmc), per launch
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
        roof = {}
        for name in ("solve_grid", "berry_flux"):
            if name in kern:
                gbs = alg[name] / (kern[name]["avg_ms"] * 1e-3) / 1e9
                roof[name] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": alg[name],
                              "avg_launch_ms": kern[name]["avg_ms"]}
        out = {
            "metric": "k-points solved/sec (H(k)+eigh) and Berry-flux/sec, Haldane 2048^2 mesh",
            "value": nk_step * args.steps / t_max, "unit": "k-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Haldane (2 orb, 9 hops) wf_array solve_on_grid + berry_flux([0]), "
                                   "%d x %d k-mesh per GPU (BASELINE.json configs[2])" % (MESH, MESH),
                       "global_mesh": [MESH * world, MESH], "start_k": [-0.5, -0.5],
                       "sharding": "k-slabs along mesh axis 0, halo row recomputed", "gather": gather},
            "roofline": dict(roof.get(dom, {}), kernel=dom, traffic=traffic),
            "kernels": kern, "empty_bracket_ms": ev_ms,
            "roofline_all": roof,
            "solve_kpts_per_s": MESH * MESH / (kern["solve_grid"]["avg_ms"] * 1e-3) if "solve_grid" in kern else None,
            "flux_plaq_per_s": MESH * MESH / (kern["berry_flux"]["avg_ms"] * 1e-3) if "berry_flux" in kern else None,
            "check": {"chern": chern, "min_gap": float(allv[:, 1].min())},
            "device": info["name"].strip() or "gfx950", "compute_units": info["compute_units"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        assert args.no_check or abs(chern + 1.0) < 1e-9, "Chern number %r != -1" % chern
        print(json.dumps(out), flush=True)

    if world > 1:
        # The path's one collective, the RCCL all-gather over xGMI (tbk_comm_*), run on the same numbers and
        # checked against the gloo result.  A communicator that never comes up must not cost the measurement:
        # a watchdog prints the line (rank 0) and ends the process if the RCCL stage is still stuck after 90 s.
        import threading

        def give_up():
            if rank == 0:
                report("gloo (rccl all-gather timed out)")
            os._exit(0)

        dog = threading.Timer(float(os.environ.get("TBK_BENCH_RCCL_TIMEOUT", "90")), give_up)
        dog.daemon = True
        dog.start()

        def all_ok(ok):
            """Every rank takes the same branch: true only if the step succeeded everywhere."""
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        err = ""
        uid_bytes = None
        if rank == 0:
            try:
                uid = (C.c_ubyte * 128)()
                _lib.check(lib.tbk_comm_unique_id(uid))
                uid_bytes = bytes(uid)
            except Exception as e:
                err = "unique_id: %s" % e
        box = [uid_bytes]
        dist.broadcast_object_list(box, src=0)
        ok = box[0] is not None
        if ok:
            try:
                uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
                _lib.check(lib.tbk_comm_init(ctx.handle, uid, world, rank))
            except Exception as e:
                ok, err = False, "init: %s" % e
        ok = all_ok(ok)
        if ok:
            try:
                send, recv = C.c_void_p(), C.c_void_p()
                _lib.check(lib.tbk_dev_alloc(ctx.handle, 24, C.byref(send)))
                _lib.check(lib.tbk_dev_alloc(ctx.handle, 24 * world, C.byref(recv)))
                _lib.check(lib.tbk_dev_upload(ctx.handle, send, mine.ctypes.data_as(C.c_void_p), 24))
                _lib.check(lib.tbk_comm_allgather_f64(ctx.handle, send, recv, 3))
                got = np.zeros((world, 3))
                _lib.check(lib.tbk_dev_download(ctx.handle, got.ctypes.data_as(C.c_void_p), recv, 24 * world))
                if not np.array_equal(got, allv):
                    ok, err = False, "allgather: result differs from the gloo gather"
            except Exception as e:
                ok, err = False, "allgather: %s" % e
            ok = all_ok(ok)
        dog.cancel()
        if ok:
            gather = "rccl_allgather (equal to the gloo gather)"
        else:
            gather = "gloo (rccl all-gather unavailable)"
            if err:
                sys.stderr.write("[bench] rank %d: RCCL gather unavailable (%s)\n" % (rank, " ".join(str(err).split())))
    if rank == 0:
        report(gather)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    _lib.check(lib.tbk_wfs_free(hw))


if __name__ == "__main__":
    main()
