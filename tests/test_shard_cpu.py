"""The N>1 path on CPU: pure index partitioning, and a world_size-2 gloo run in which
each rank computes its k-slab (here with the oracle standing in for the kernels, since
this suite has no GPU), recomputes its halo row, and the ranks exchange one small
gather -- the same structure bench.py runs on N GPUs with RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import tb_oracle as orc
from pythtb_amd import shard


def test_split_list_covers_exactly_once():
    for n in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                b, e = shard.split_list(n, w, r)
                assert 0 <= b <= e <= n
                seen.extend(range(b, e) if n < 100 else [b, e])
            if n < 100:
                assert seen == list(range(n))
            else:
                assert seen[0] == 0 and seen[-1] == n and all(seen[i] == seen[i + 1] for i in range(1, 2 * w - 1, 2))
    with pytest.raises(ValueError):
        shard.split_list(5, 2, 2)


def test_split_rows_halo_and_pbc():
    for mesh0 in (3, 17, 2049, 16385):
        for w in (1, 2, 3, 8):
            if mesh0 - 1 < w:
                continue
            nxt = 0
            for r in range(w):
                row0, nrows = shard.split_rows(mesh0, w, r)
                assert row0 == nxt and nrows >= 2
                nxt = row0 + nrows - 1                       # last stored row = next slab's first
            assert nxt == mesh0 - 1                          # the last slab ends on the periodic image
    with pytest.raises(ValueError):
        shard.split_rows(3, 4, 3)


def test_split_strings_keeps_strings_local():
    axis, b, e = shard.split_strings([4097, 513], 0, 8, 3)
    assert axis == 1 and (b, e) == shard.split_list(513, 8, 3)
    axis, b, e = shard.split_strings([257, 257, 257], 2, 8, 7)
    assert axis in (0, 1) and e == 257
    with pytest.raises(ValueError):
        shard.split_strings([100], 0, 2, 0)


def test_slab_sum_equals_unsharded_flux():
    """Sharding identity used by bench.py: slabs with a recomputed halo row give the
    same plaquettes, so partial sums add to the unsharded flux (and Chern number)."""
    m = orc.haldane(0.0)
    mesh0, mesh1, start = 25, 13, [-0.5, -0.5]
    wfs, gaps = orc.solve_on_grid(m, [mesh0, mesh1], start)
    ref = orc.berry_flux(wfs, 2, [0], individual_phases=True, vectorised=True)
    for world in (2, 3, 8):
        parts, mins = [], []
        for r in range(world):
            row0, nrows = shard.split_rows(mesh0, world, r)
            slab, g = slab_solve(m, mesh0, mesh1, start, row0, nrows)
            assert np.max(np.abs(slab - wfs[row0:row0 + nrows])) < 1e-13
            p = orc.berry_flux(slab, 2, [0], individual_phases=True, vectorised=True)
            assert np.max(np.abs(p - ref[row0:row0 + nrows - 1])) < 1e-12
            parts.append(p.sum())
            mins.append(g)
        assert abs(sum(parts) - ref.sum()) < 1e-12
        assert abs(round(sum(parts) / (2 * np.pi)) + 1) == 0
        assert abs(min(mins) - gaps[0]) < 1e-12


def slab_solve(m, mesh0, mesh1, start, row0, nrows):
    """Rows [row0,row0+nrows) of the global mesh, each point solved locally; the
    global last row/column are periodic images (what tbk_wfs_solve_grid does per slab)."""
    slab = np.zeros((nrows, mesh1, m._nsta, m._norb), dtype=complex)
    gmin = np.inf
    ph = [np.exp(-2j * np.pi * m._orb[:, m._per[d]]) for d in range(2)]
    for i in range(nrows):
        gi = row0 + i
        wi = gi == mesh0 - 1
        for j in range(mesh1):
            wj = j == mesh1 - 1
            k = [start[0] + float(0 if wi else gi) / float(mesh0 - 1), start[1] + float(0 if wj else j) / float(mesh1 - 1)]
            w, v = orc.solve_one(m, k, True)
            if wi:
                v = v * ph[0]
            if wj:
                v = v * ph[1]
            slab[i, j] = v
            gmin = min(gmin, w[1] - w[0])
    return slab, gmin


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["TBK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["TBK_ROOT"], "tests"))
from oracle import tb_oracle as orc
from pythtb_amd import shard
from test_shard_cpu import slab_solve
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
m = orc.haldane(0.0)
mesh0, mesh1, start = 17, 9, [-0.5, -0.5]
row0, nrows = shard.split_rows(mesh0, world, rank)
slab, gmin = slab_solve(m, mesh0, mesh1, start, row0, nrows)
part = orc.berry_flux(slab, 2, [0], vectorised=True)
mine = torch.tensor([part, gmin], dtype=torch.float64)
buf = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
dist.barrier()
dist.all_gather(buf, mine)                      # the one collective of the path
allv = np.stack([b.numpy() for b in buf])
if rank == 0:
    wfs, gaps = orc.solve_on_grid(m, [mesh0, mesh1], start, vectorised=True)
    ref = orc.berry_flux(wfs, 2, [0], vectorised=True)
    assert abs(allv[:, 0].sum() - ref) < 1e-12, (allv, ref)
    assert abs(allv[:, 1].min() - gaps[0]) < 1e-12
    assert round(allv[:, 0].sum() / (2 * np.pi)) == -1
    print("GLOO_SHARD_OK", world)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_gloo_run(tmp_path):
    pytest.importorskip("torch")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TBK_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "GLOO_SHARD_OK 2" in res.stdout


# ---------------------------------------------------------------- multi-GPU drivers of configs[3] / [4] (pythtb_amd/multi.py)
MULTI_WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["TBK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["TBK_ROOT"], "tests"))
from oracle import tb_oracle as orc
from pythtb_amd import multi
import pythtb_amd as tb
import helpers as hp


class OracleWf(object):
    '''Stand-in for wf_array on a box without a GPU: the oracle solves the GLOBAL mesh and hands out the window
    (numpy's eigh is a function of the matrix alone, so this is what a window solve returns).'''
    def __init__(self, model, mesh):
        self.model, self.mesh = model, [int(x) for x in mesh]
    def solve_on_grid_window(self, start, offset, gmesh):
        wfs, gaps = orc.solve_on_grid(self.model, list(gmesh), start, vectorised=True)
        sl = tuple(slice(o, o + n) for o, n in zip(offset, self.mesh))
        self.wfs = wfs[sl]
        # min gaps over THIS window's solved points
        ev = []
        idx = np.indices(self.mesh).reshape(len(self.mesh), -1).T
        for ii in idx:
            k = [start[d] + ((ii[d] + offset[d]) % (gmesh[d] - 1)) / (gmesh[d] - 1) for d in range(len(self.mesh))]
            ev.append(np.linalg.eigvalsh(orc.gen_ham(self.model, k)))
        return np.diff(np.array(ev), axis=1).min(axis=0)
    def berry_phase(self, occ, dir, contin=True, berry_evals=False):
        return orc.berry_phase(self.wfs, len(self.mesh), list(occ), dir, contin=contin, berry_evals=berry_evals)
    def berry_flux(self, occ, dirs=None, individual_phases=False):
        return orc.berry_flux(self.wfs, len(self.mesh), list(occ), dirs=dirs, individual_phases=individual_phases, vectorised=True)


SOCKET = os.environ.get("TBK_TEST_RDZV") == "socket"      # the launcher's TCP rendezvous instead of torch.distributed (no torch)
Base = multi.SocketComm if SOCKET else multi.GlooComm


class Counting(Base):
    '''counts the collectives a driver issues (north_star: ONE gather per driver)'''
    calls = 0
    def allgatherv(self, mine, counts):
        Counting.calls += 1
        return Base.allgatherv(self, mine, counts)


if SOCKET:
    from pythtb_amd import launch
    dist = launch.Rendezvous(timeout=300.0)
    dist.destroy_process_group = dist.close
    rank, world = dist.rank, dist.world
else:
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
comm = Counting(dist)
# solve_all on a k list (configs[1] / the solve_all leg of configs[4]) in small: 11 k-points do not divide by 2 or 3
hal = hp.haldane(tb.tb_model, 0.2)
k11 = np.random.default_rng(3).random((11, 2))
def per_k(model, n):
    # one k at a time, so that the bits do not depend on how the list is cut
    return lambda kk: np.concatenate([orc.solve_all_vec(model, kk[i:i + 1]) for i in range(len(kk))] + [np.zeros((n, 0))], axis=1)
oracle_chunk = per_k(hal, 2)
ev = multi.solve_all_sharded(hal, k11, comm, rank, world, solve_chunk=oracle_chunk)
assert Counting.calls == 1, Counting.calls
assert ev.shape == (2, 11) and ev.flags["C_CONTIGUOUS"] and np.array_equal(ev, oracle_chunk(k11))
assert np.max(np.abs(ev - orc.solve_all_vec(hal, k11))) < 1e-13
chunks = [e - b for b, e in multi.plan_list(11, world)]
assert sum(chunks) == 11 and len(set(chunks)) > 1, chunks
# the ROOTED gather (SURVEY.md 8e: ret_eval is one array on one caller): only the root returns it
root = world - 1
evr = multi.solve_all_sharded(hal, k11, comm, rank, world, solve_chunk=oracle_chunk, root=root)
assert (evr is None) == (rank != root)
if rank == root:
    assert evr.shape == (2, 11) and np.array_equal(evr, ev)
# a rank whose solve fails: every rank raises BEFORE the gather (nobody is left inside the collective, no unchecked data)
def failing(kk):
    if rank == world - 1:
        raise RuntimeError("eigen-solver did not converge (test)")
    return oracle_chunk(kk)
Counting.calls = 0
try:
    multi.solve_all_sharded(hal, k11, comm, rank, world, solve_chunk=failing)
    raise SystemExit("a failing rank went unnoticed on rank %d" % rank)
except RuntimeError as e:
    assert ("did not converge" in str(e)) == (rank == world - 1) and ("another rank" in str(e)) == (rank != world - 1), str(e)
assert Counting.calls == 0
m3s = hp.random_model(tb.tb_model, 3, 3, 1, 5)                # more ranks than k-points: empty chunks
k2 = np.random.default_rng(4).random((2, 3))
oc3 = per_k(m3s, 3)
assert np.array_equal(multi.solve_all_sharded(m3s, k2, comm, rank, world, solve_chunk=oc3), oc3(k2))
Counting.calls = 0
# configs[3] in small: Kane-Mele, 7 strings over `world` ranks (uneven for 2 and 3)
km = hp.kane_mele(tb.tb_model, "odd")
mesh, start = [9, 7], [-0.5, -0.5]
got, _ = multi.wilson_loops_sharded(OracleWf, km, mesh, start, [0, 1], comm, rank, world)
counts = [p[2] - p[1] for p in multi.plan_strings(mesh, 0, world)]
assert sum(counts) == 7 and (len(set(counts)) > 1 or world == 7), counts
full, _ = orc.solve_on_grid(km, mesh, start, vectorised=True)
ref = orc.berry_phase(full, 2, [0, 1], 0, contin=False, berry_evals=True)
assert got.shape == (7, 2) and np.array_equal(got, ref), np.abs(got - ref).max()
got1, _ = multi.wilson_loops_sharded(OracleWf, km, mesh, start, [2, 3], comm, rank, world, berry_evals=False)
assert np.array_equal(got1, orc.berry_phase(full, 2, [2, 3], 0, contin=False))
# configs[4] in small: a 3-orbital cubic model, slabs along axis 0 (5 plaquette rows over `world` ranks), strings along 2
m3 = hp.random_model(tb.tb_model, 3, 3, 1, 11)
mesh3, start3 = [max(6, world + 2), 4, 5], [0.1, 0.2, 0.3]   # (a slab owns at least one plaquette row)
Counting.calls = 0
ph, gaps = multi.mesh_phases_sharded(OracleWf, m3, mesh3, start3, [0, 1], comm, rank, world, dir=2)
assert Counting.calls == 1, Counting.calls                    # phases and min gaps travel in one buffer
full3, gaps3 = orc.solve_on_grid(m3, mesh3, start3, vectorised=True)
assert ph.shape == (mesh3[0], 4) and np.array_equal(ph, orc.berry_phase(full3, 3, [0, 1], 2, contin=False))
assert np.max(np.abs(gaps - gaps3)) < 1e-12
# configs[2] in small: Haldane, slabs along axis 0 with the halo row recomputed, ONE gather of [partial flux | min gaps]
hal0 = hp.haldane(tb.tb_model, 0.0)
mesh2, start2 = [max(12, world + 3), 9], [-0.5, -0.5]        # 11 plaquette rows over 2 / 3 ranks: uneven
fullw, fgaps = orc.solve_on_grid(hal0, mesh2, start2, vectorised=True)
for occ2 in ([0], [0, 1]):
    Counting.calls = 0
    tot, g2 = multi.berry_flux_sharded(OracleWf, hal0, mesh2, start2, occ2, comm, rank, world)
    assert Counting.calls == 1, Counting.calls
    ref_plaq = orc.berry_flux(fullw, 2, occ2, individual_phases=True, vectorised=True)
    assert abs(tot - ref_plaq.sum()) < 1e-11 and np.max(np.abs(g2 - fgaps)) < 1e-12
    plq, g2b = multi.berry_flux_sharded(OracleWf, hal0, mesh2, start2, occ2, comm, rank, world, individual_phases=True)
    assert plq.shape == ref_plaq.shape and np.max(np.abs(plq - ref_plaq)) < 1e-11 and np.array_equal(g2b, g2)
    plt, _ = multi.berry_flux_sharded(OracleWf, hal0, mesh2, start2, occ2, comm, rank, world, dirs=[1, 0], individual_phases=True)
    ref_t = orc.berry_flux(fullw, 2, occ2, dirs=[1, 0], individual_phases=True, vectorised=True)
    assert plt.shape == ref_t.shape == (mesh2[1] - 1, mesh2[0] - 1) and np.max(np.abs(plt - ref_t)) < 1e-11
assert abs(multi.berry_flux_sharded(OracleWf, hal0, mesh2, start2, [0], comm, rank, world)[0] / (2 * np.pi) + 1.0) < 1e-10   # Chern -1
if rank == 0:
    print("MULTI_DRIVERS_OK", world, counts)
    if os.environ.get("TBK_TEST_OUT"):
        open(os.environ["TBK_TEST_OUT"], "a").write("MULTI_DRIVERS_OK %d\n" % world)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3, 8])
def test_multi_gpu_drivers_on_gloo_with_uneven_counts(tmp_path, world):
    """The drivers bench_configs.py --gpus N runs for BASELINE configs[1] / [3] / [4], here with 2, 3 and 8 gloo ranks
    (8 = the node north_star names; some ranks then own no string or no k-point), an oracle-backed wf_array stand-in and
    counts that do not divide evenly (all-gather-v), plus the rooted gather and a rank whose solve fails."""
    pytest.importorskip("torch")
    script = tmp_path / "multi_worker.py"
    script.write_text(MULTI_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TBK_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "MULTI_DRIVERS_OK %d" % world in res.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_multi_gpu_drivers_on_the_socket_rendezvous(tmp_path, world):
    """The same drivers with multi.SocketComm over launch.Rendezvous (what bench.py / bench_configs.py --gpus N use since round 6
    for the control plane and the labelled fallback): started by our own launcher, no torch.distributed in the ranks."""
    from pythtb_amd import launch
    script = tmp_path / "multi_worker.py"
    script.write_text(MULTI_WORKER.replace("import torch.distributed as dist\n", "dist = None\n"))
    env = dict(os.environ, TBK_ROOT=ROOT, OMP_NUM_THREADS="1", TBK_TEST_RDZV="socket")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = tmp_path / "out.txt"
    rc = launch.spawn_ranks(str(script), [], world, env=dict(env, TBK_TEST_OUT=str(out)), timeout=600)
    assert rc == 0
    assert "MULTI_DRIVERS_OK %d" % world in open(out).read()


def test_multi_plans_cover_every_string_and_plane_once():
    from pythtb_amd import multi
    for world in (1, 2, 3, 5, 8):
        p = multi.plan_strings([4097, 513], 0, world)
        assert [x[0] for x in p] == [1] * world
        assert p[0][1] == 0 and p[-1][2] == 513 and all(p[i][2] == p[i + 1][1] for i in range(world - 1))
        assert all(x[4] - x[3] >= 2 and x[3] <= x[1] and x[2] <= x[4] for x in p)
        s = multi.plan_slabs(257, world)
        assert sum(x[2] for x in s) == 257 and s[0][0] == 0 and s[-1][0] + s[-1][1] == 257
    p = multi.plan_strings([9, 3], 0, 3)                      # one string per rank: windows widened to two points
    assert [(x[1], x[2]) for x in p] == [(0, 1), (1, 2), (2, 3)] and all(x[4] - x[3] == 2 for x in p)
