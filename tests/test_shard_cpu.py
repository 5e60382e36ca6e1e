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
