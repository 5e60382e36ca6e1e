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
    # round 6, final build: r06a (the two-call headline step), r06f (--fused-headline), r06bcfg (bench_configs.py B D E P L W H R),
    # r06ecfg (bench_configs.py E alone: clean per-dispatch counts of k_e16<1,true>, k_e16<0,false>, k_chain_prod_tile<8,2,false>)
    ("k_grid_rows<2,1>", "r06a/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r06a/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows_flux<2,1,1>", "r06f/pmc_per_dispatch.json", "k_grid_rows_flux<2,1,1>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r06bcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r06bcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_e16<1>", "r06ecfg/pmc_per_dispatch.json", "k_e16<1,true>", 65 ** 3),
    ("k_e16<0,false>", "r06ecfg/pmc_per_dispatch.json", "k_e16<0,false>", 64 ** 3),
    ("k_chain_prod_tile<8,2>", "r06ecfg/pmc_per_dispatch.json", "k_chain_prod_tile<8,2,false>", 65 ** 3),
    # round 5, final build: r05c (the two-call headline step), r05ccfg (bench_configs.py B D E), r05p (leg P)
    ("k_grid_rows<2,1>", "r05c/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r05c/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r05dcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),      # (after the sort's short way)
    ("k_flux_rows<2,4>", "r05dcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_e16<1>", "r05ecfg/pmc_per_dispatch.json", "k_e16<1>", 65 ** 3),                        # (after the round-5 work on k_e16)
    ("k_mesh_evals<2,1>", "r05ccfg/pmc_per_dispatch.json", "k_mesh_evals<2,1>", 1024 * 1024),
    ("k_chain_prod_tile<8,2>", "r05ecfg/pmc_per_dispatch.json", "k_chain_prod_tile<8,2,false>", 65 ** 3),
    ("k_solve_small_multi<2,false,2>", "r05ccfg/pmc_per_dispatch.json", "k_solve_small_multi<2,false,2>", 1024 * 1024),
    ("k_solve_small<2,0,true>", "r05ccfg/pmc_per_dispatch.json", "k_solve_small<2,0,true>", 1024 * 1024),
    # round 4, final build: r04j (the two-call step), r04k (--fused-headline), r04jcfg (bench_configs.py B D E under TBK_TW16_STREAMS=1)
    ("k_grid_rows_flux<2,1,1>", "r04k/pmc_per_dispatch.json", "k_grid_rows_flux<2,1,1>", 2049 * 2049),
    ("k_grid_rows<2,1>", "r04j/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r04j/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r04jcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r04jcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    ("k_e16<1>", "r04jcfg/pmc_per_dispatch.json", "k_e16<1>", 65 ** 3),
    ("k_mesh_evals<2,1>", "r04jcfg/pmc_per_dispatch.json", "k_mesh_evals<2,1>", 1024 * 1024),
    ("k_chain_prod_tile<8,2>", "r04jcfg/pmc_per_dispatch.json", "k_chain_prod_tile<8,2>", 65 ** 3),
    ("k_solve_small_multi<2,false,2>", "r04jcfg/pmc_per_dispatch.json", "k_solve_small_multi<2,false,2>", 1024 * 1024),
    ("k_solve_small<2,0,true>", "r04jcfg/pmc_per_dispatch.json", "k_solve_small<2,0,true>", 1024 * 1024),
    # round 3, final build (the three-kernel n = 9..16 path, TBK_E16=0, is still counted from here)
    ("k_tw16<1>", "r03lcfg/pmc_per_dispatch.json", "k_tw16_tridiag<1>+k_tw16_eigvals<1>+k_tw16_vectors<1>", 65 ** 3),
    ("k_grid_rows_flux<2,1,1>", "r03l/pmc_per_dispatch.json", "k_grid_rows_flux<2,1,1>", 2049 * 2049),
    ("k_grid_rows<2,1>", "r03l2/pmc_per_dispatch.json", "k_grid_rows<2,1>", 2049 * 2049),
    ("k_flux_rows<1,2>", "r03l2/pmc_per_dispatch.json", "k_flux_rows<1,2>", 2049 * 2049),
    ("k_grid_rows<4,1>", "r03lcfg/pmc_per_dispatch.json", "k_grid_rows<4,1>", 4097 * 513),
    ("k_flux_rows<2,4>", "r03lcfg/pmc_per_dispatch.json", "k_flux_rows<2,4>", 4097 * 513),
    # round 2: the single-kernel and first three-kernel forms of the n = 16 solve (history; bench.py does not name them any more)
    ("k_solve_ql16<1,true>", "r02hcfg/pmc_per_dispatch.json", "k_solve_ql16<1,true,2>+k_ql16_lanes<1>+k_ql16_replay<1>", 65 ** 3 // 2),
]
# ---- the flop model behind bench.py's `useful_flop_frac` (SURVEY.md 8d "ALGORITHMIC flops per k-point").
# Real fp64 flops per mesh point of the ALGORITHM (not of the instruction stream: masked lanes, moves, cross-lane reductions and
# work replicated over the lanes of a matrix are what the issue fraction counts and this does not).  Conventions: complex
# multiply-add 8, complex multiply 6, real fma 2, sqrt / reciprocal / atan2 / sincos counted as 1 / 1 / 30 / 40.
#   assembly        8 per merged (slot, R) term of S(k); on a regular mesh the d + norb phases come from
#                   per-axis tables (0 per point), on a k list they are d + norb sincos
#   eigen-solve     n = 2: closed form (one sqrt, one rsqrt, ~30) + vectors (~30);
#                   n = 3, 4: Householder (16/3) n^3 + implicit QL with accumulated vectors ~ 12 n^3 (real rotations on complex rows)
#                   n = 9..16: Householder (16/3) n^3 + QL on (d, e) ~ 30 n^2 + twisted factorisation 10 n^2 + one Newton-Schulz
#                   step 4 n^3 + back-transformation by the reflectors 8 n * n(n-1)/2 ~ 4 n^3 (pythtb.py:939,944 = zheevd's work)
#   orbital phases  6 n^2 (eigenvector components x conj e_a)
#   plaquette       two new link overlaps (nocc^2 n complex multiply-adds each), nocc x nocc determinants, the product of four
#                   link variables (18) and one atan2 (30)
def _eig16(n):
    return (16.0 / 3.0) * n ** 3 + 30 * n ** 2 + 10 * n ** 2 + 4 * n ** 3 + 4 * n ** 3


FLOPS = {
    # Haldane: 15 merged terms (two diagonal slots of 6, one off-diagonal of 3)
    "k_grid_rows<2,1>": (15 * 8 + 60 + 6 * 4, "mesh solve n=2: 15 terms x 8 + closed-form 2x2 with vectors 60 + orbital phases 24"),
    "k_flux_rows<1,2>": (2 * 2 * 8 + 18 + 30, "plaquette nocc=1 n=2: two links x 2 cmadd + product of four link variables 18 + atan2 30"),
    "k_grid_rows_flux<2,1,1>": (15 * 8 + 60 + 24 + 2 * 2 * 8 + 18 + 30, "the two rows above in one pass"),
    # Kane-Mele: 10 slots, 9 hops of 2x2 blocks + onsite -> ~60 merged scalar terms
    "k_grid_rows<4,1>": (60 * 8 + (16.0 / 3.0) * 64 + 12 * 64 + 6 * 16, "mesh solve n=4: 60 terms x 8 + Householder 341 + QL with vectors 768 + phases 96 (the model of rounds 2-4, kept so that the fraction stays comparable; round 5's guided sweeps do ~25 % less QL work than it assumes)"),
    "k_flux_rows<2,4>": (2 * 4 * 4 * 8 + 2 * 14 + 18 + 30, "plaquette nocc=2 n=4: two links x 4 entries x 4 cmadd + two 2x2 dets + product + atan2"),
    # cubic16: 136 slots x 4 lattice vectors (R-grouped) = 544 complex multiply-adds
    "k_tw16<1>": (544 * 8 + _eig16(16) + 6 * 256, "mesh solve n=16: 544 cmadd assembly + tridiagonalise 21.8k + QL 7.7k + twisted 2.6k + Newton-Schulz 16.4k + back-transform 16.4k + phases 1.5k"),
    "k_e16<1>": (544 * 8 + _eig16(16) + 6 * 256, "mesh solve n=16 (one fused kernel): same algorithm as k_tw16<1>"),
    "k_mesh_evals<2,1>": (15 * 8 + 30, "mesh eigenvalues n=2: 15 terms x 8 + closed form 30 (phases from the per-axis tables)"),
    "k_chain_prod_tile<8,2>": (8 * 8 * 16 * 8 + 8 * 8 * 8 * 8, "link of 8 of 16 bands: overlap matrix 8x8x16 cmadd + running product 8^3 cmadd (the determinant is once per string)"),
    "k_solve_small_multi<2,false,2>": (4 * 40 + 15 * 8 + 30, "k list n=2 eigenvalues: 4 sincos + 15 terms x 8 + closed form 30"),
    "k_solve_small<2,0,false>": (4 * 40 + 15 * 8 + 30, "k list n=2 eigenvalues: 4 sincos + 15 terms x 8 + closed form 30"),
    "k_solve_small<2,0,true>": (4 * 40 + 15 * 8 + 60 + 24, "k list n=2 with vectors: 4 sincos + 15 terms x 8 + closed form 60 + phases 24"),
}

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
    if name in FLOPS:
        out[name]["useful_flops_per_point"] = float(FLOPS[name][0])
        out[name]["flop_model"] = FLOPS[name][1] + " (profiles/make_valu_json.py)"
json.dump(out, open(os.path.join(HERE, "valu.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
