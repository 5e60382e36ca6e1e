#!/usr/bin/env python3
"""profiles/valu.json: VALU wave-instructions per mesh point of the kernels bench.py quotes a VALU fraction for,
from the committed rocprofv3 PMC summaries (SQ_INSTS_VALU per dispatch / mesh points of that dispatch).

    python profiles/make_valu_json.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (kernel as bench.py names it, summary file, key inside it, mesh points per dispatch of that run)
# (a key "a+b+c" sums the per-dispatch counts of several kernels that each cover the same points)
SOURCES = [
    # round 3, final build (+ expi2pi, two k-points per lane on eigenvalue-only lists, four links per step): r03l, r03l2, r03lcfg (r03k* = the same a few commits earlier)
    ("k_grid_rows_flux<2,1,1>", "r03l/pmc_per_dispatch.json", "k_grid_rows_flux<2,1,1>", 2049 * 2049),
    ("k_grid_rows<2,1>", "r03l2/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r03l2/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r03lcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r03lcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_tw16<1>", "r03lcfg/pmc_per_dispatch.json", "k_tw16_tridiag<1>+k_tw16_eigvals<1>+k_tw16_vectors<1>", 65 ** 3),
    ("k_solve_small_multi<2,false,2>", "r03lcfg/pmc_per_dispatch.json", "k_solve_small_multi<2,false,2>", 1024 * 1024),
    ("k_solve_small<2,0,true>", "r03lcfg/pmc_per_dispatch.json", "k_solve_small<2,0,true>", 1024 * 1024),
    # the same before those three changes (tbk_solve / tbk_berry compiled without MachineLICM, seams + total in one kernel): r03j, r03j2, r03jcfg
    ("k_grid_rows_flux<2,1,1>", "r03j/pmc_per_dispatch.json", "k_grid_rows_flux<2,1,1>", 2049 * 2049),
    ("k_grid_rows<2,1>", "r03j2/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r03j2/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r03jcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r03jcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_tw16<1>", "r03jcfg/pmc_per_dispatch.json", "k_tw16_tridiag<1>+k_tw16_eigvals<1>+k_tw16_vectors<1>", 65 ** 3),
    ("k_solve_small<2,0,false>", "r03jcfg/pmc_per_dispatch.json", "k_solve_small<2,0,false>", 1024 * 1024),
    ("k_solve_small<2,0,true>", "r03jcfg/pmc_per_dispatch.json", "k_solve_small<2,0,true>", 1024 * 1024),
    # round 3: the fused headline kernel, the two-launch step's kernels, configs[1] / [3] / [4] (r03hcfg: TBK_TW16_STREAMS=1, so
    # that every kernel's counters belong to it alone)
    ("k_grid_rows_flux<2,1,1>", "r03h/pmc_per_dispatch.json", "k_grid_rows_flux<2,1,1>", 2049 * 2049),
    ("k_grid_rows<2,1>", "r03h2/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r03h2/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r03hcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r03hcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_tw16<1>", "r03hcfg/pmc_per_dispatch.json", "k_tw16_tridiag<1>+k_tw16_eigvals<1>+k_tw16_vectors<1>", 65 ** 3),
    ("k_solve_small<2,0,false>", "r03hcfg/pmc_per_dispatch.json", "k_solve_small<2,0,false>", 1024 * 1024),
    ("k_solve_small<2,0,true>", "r03hcfg/pmc_per_dispatch.json", "k_solve_small<2,0,true>", 1024 * 1024),
    ("k_grid_rows<4,1>", "r02hcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r02hcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    # the 65^3 mesh of the collection goes through the three kernels in two chunks
    ("k_solve_ql16<1,true>", "r02hcfg/pmc_per_dispatch.json", "k_solve_ql16<1,true,2>+k_ql16_lanes<1>+k_ql16_replay<1>", 65 ** 3 // 2),
    ("k_grid_rows<2,1>", "r02i/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r02i/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<2,1>", "r02g/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r02g/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r02gcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r02gcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_solve_ql16<1,true>", "r02gcfg/pmc_per_dispatch.json", "k_solve_ql16<1,true,0>", 65 ** 3),
    ("k_grid_rows<2,1>", "r02f/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r02f/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r02fcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r02fcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_solve_ql16<1,true>", "r02fcfg/pmc_per_dispatch.json", "k_solve_ql16<1,true,0>", 65 ** 3),
    ("k_grid_rows<2,1>", "r02e/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r02e/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r02ecfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r02ecfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_solve_ql16<1,true>", "r02ecfg/pmc_per_dispatch.json", "k_solve_ql16<1,true>", 65 ** 3),
    ("k_grid_rows<2,1>", "r02c/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r02c/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r02ccfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r02ccfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_solve_ql16<1,true>", "r02ccfg/pmc_per_dispatch.json", "k_solve_ql16<1,true>", 65 ** 3),
    # fallbacks from earlier collections (used only while the entry above is missing)
    ("k_grid_rows<2,1>", "r01g/pmc_per_dispatch.json", "k_grid_rows", 2049 * 2049),
    ("k_flux_rows<1,2>", "r01g/pmc_per_dispatch.json", "k_flux_rows", 2049 * 2049),
    ("k_grid_rows<4,1>", "r02a/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r02a/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
]
out = {}
for name, rel, key, points in SOURCES:
    if name in out:
        continue
    path = os.path.join(HERE, rel)
    if not os.path.exists(path):
        continue
    table = json.load(open(path))
    recs = [table.get(k) for k in key.split("+")]
    if any(not r or "SQ_INSTS_VALU" not in r for r in recs):
        continue
    insts = sum(r["SQ_INSTS_VALU"] for r in recs)
    out[name] = {"kernel": key, "source": "profiles/" + rel, "SQ_INSTS_VALU_per_dispatch": insts,
                 "mesh_points_per_dispatch": points, "valu_wave_insts_per_point": insts / points}
json.dump(out, open(os.path.join(HERE, "valu.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
