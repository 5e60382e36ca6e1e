"""One-process-per-GPU drivers for the two multi-GPU configurations of BASELINE.json:

* configs[3]: Kane-Mele wf_array([4097, 513]) -- Wilson-loop eigenphases of the 513 strings along axis 0
  (reference loop: pythtb.py:2987-2996).  Strings shard along axis 1; 513 strings over 8 ranks is an uneven
  split (65 + 7 x 64), so the gather is an all-gather-v.
* configs[4]: cubic16 wf_array([257, 257, 257]) -- solve_on_grid (pythtb.py:2499-2511) and
  berry_phase(range(8), dir=2) (pythtb.py:3002-3025).  Slabs along axis 0, each with its recomputed halo plane;
  a rank reports the planes it owns and the gather assembles the (257, 257) phase array.

Nothing is exchanged while computing: every rank solves its own window of the global mesh
(`wf_array.solve_on_grid_window`; periodic images and halo planes are recomputed, bit-identically).  The one
collective is the gather of the per-rank results at the end, through a `Comm` object:

    RcclComm   tbk_comm_allgatherv_f64 on device buffers (RCCL over xGMI) -- the product path
    GlooComm   torch.distributed (gloo) on host arrays -- rendezvous, CPU tests, and the fallback the bench
               drivers report through while the RCCL leg is being validated

The drivers take the `wf_array` class to use, so the CPU test-suite can run them (2 ranks, gloo, uneven
counts) with an oracle-backed stand-in where no GPU exists; the library itself never does that.
"""
import ctypes as C

import numpy as np

from . import shard

__all__ = ["GlooComm", "RcclComm", "plan_strings", "plan_slabs", "wilson_loops_sharded", "mesh_phases_sharded"]


# ---------------------------------------------------------------------------------------------- plans
def plan_strings(mesh, dir, world):
    """Per rank: (axis, begin, end, w_begin, w_end) -- the rank reports the strings [begin, end) of `axis` and
    solves the window [w_begin, w_end) of it (a wf_array axis needs at least two points, so a one-string
    share is widened by a neighbour that is computed and not reported)."""
    plans = []
    for r in range(world):
        axis, b, e = shard.split_strings(mesh, dir, world, r)
        wb, we = b, e
        if we - wb < 2:
            wb = max(0, min(wb, int(mesh[axis]) - 2))
            we = wb + 2
        plans.append((axis, b, e, wb, we))
    return plans


def plan_slabs(mesh0, world):
    """Per rank: (row0, nrows_stored, own) -- the slab stores global rows [row0, row0 + nrows_stored) of axis 0
    (its last row is the next slab's first, recomputed) and reports the first `own` of them; the last rank also
    reports the final row (the periodic image)."""
    plans = []
    for r in range(world):
        row0, nrows = shard.split_rows(mesh0, world, r)
        plans.append((row0, nrows, nrows if r == world - 1 else nrows - 1))
    return plans


# ---------------------------------------------------------------------------------------------- communicators
class GlooComm(object):
    """torch.distributed process group (gloo) used as an all-gather-v of float64 host arrays."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.name = "gloo"

    def allgatherv(self, mine, counts):
        import torch
        mine = np.ascontiguousarray(mine, dtype=np.float64).reshape(-1)
        assert mine.size == counts[self.rank]
        cap = max(max(counts), 1)
        pad = np.zeros(cap)
        pad[:mine.size] = mine
        buf = [torch.zeros(cap, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(buf, torch.from_numpy(pad))
        return np.concatenate([b.numpy()[:c] for b, c in zip(buf, counts)])


class RcclComm(object):
    """tbk_comm_* (RCCL): all-gather-v of float64 device buffers.  `uid` is the 128-byte id created on rank 0
    (tbk_comm_unique_id) and distributed by the launcher."""

    def __init__(self, ctx, uid, world, rank):
        from . import _lib
        self._lib, self.lib, self.ctx = _lib, _lib.lib, ctx
        self.rank, self.world = rank, world
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        _lib.check(self.lib.tbk_comm_init(ctx.handle, buf, world, rank))
        self.name = "rccl"

    def allgatherv(self, mine, counts):
        _lib, lib, ctx = self._lib, self.lib, self.ctx
        mine = np.ascontiguousarray(mine, dtype=np.float64).reshape(-1)
        assert mine.size == counts[self.rank]
        total = int(sum(counts))
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        dsp = np.ascontiguousarray(np.concatenate([[0], np.cumsum(counts)[:-1]]), dtype=np.int64)
        send, recv = C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 8 * max(mine.size, 1), C.byref(send)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 8 * max(total, 1), C.byref(recv)))
        try:
            if mine.size:
                _lib.check(lib.tbk_dev_upload(ctx.handle, send, mine.ctypes.data_as(C.c_void_p), 8 * mine.size))
            i64p = C.POINTER(C.c_int64)
            _lib.check(lib.tbk_comm_allgatherv_f64(ctx.handle, send, mine.size, recv, cnt.ctypes.data_as(i64p),
                                                   dsp.ctypes.data_as(i64p)))
            out = np.zeros(total)
            if total:
                _lib.check(lib.tbk_dev_download(ctx.handle, out.ctypes.data_as(C.c_void_p), recv, 8 * total))
        finally:
            lib.tbk_dev_free(ctx.handle, send)
            lib.tbk_dev_free(ctx.handle, recv)
        return out

    def close(self):
        self._lib.check(self.lib.tbk_comm_destroy(self.ctx.handle))


# ---------------------------------------------------------------------------------------------- drivers
def wilson_loops_sharded(wf_array_cls, model, mesh, start_k, occ, comm, rank, world, dir=0, berry_evals=True):
    """configs[3]: berry_phase(occ, dir, contin=False, berry_evals) of a 2-D solve_on_grid array, strings sharded
    over the ranks.  Returns (phases of ALL strings, this rank's min gaps): (n_strings, nocc) or (n_strings,)."""
    mesh = [int(x) for x in mesh]
    if len(mesh) != 2:
        raise ValueError("wilson_loops_sharded drives 2-D arrays")
    plans = plan_strings(mesh, dir, world)
    axis, b, e, wb, we = plans[rank]
    local = list(mesh)
    local[axis] = we - wb
    off = [0, 0]
    off[axis] = wb
    w = wf_array_cls(model, local)
    gaps = w.solve_on_grid_window(start_k, off, mesh)
    ph = np.asarray(w.berry_phase(occ, dir, contin=False, berry_evals=berry_evals))
    mine = ph[b - wb:e - wb]
    per = len(occ) if berry_evals else 1
    counts = [(p[2] - p[1]) * per for p in plans]
    allv = comm.allgatherv(mine, counts)
    n_str = mesh[axis]
    return (allv.reshape(n_str, per) if berry_evals else allv.reshape(n_str)), gaps


def mesh_phases_sharded(wf_array_cls, model, mesh, start_k, occ, comm, rank, world, dir=2):
    """configs[4]: solve_on_grid on a 3-D mesh in slabs along axis 0 and berry_phase(occ, dir, contin=False) with
    dir != 0.  Returns (phases (N_a, N_b) over the two other axes in original order, global min gaps)."""
    mesh = [int(x) for x in mesh]
    if len(mesh) != 3 or dir == 0:
        raise ValueError("mesh_phases_sharded drives 3-D arrays with strings along axis 1 or 2")
    plans = plan_slabs(mesh[0], world)
    row0, nrows, own = plans[rank]
    w = wf_array_cls(model, [nrows, mesh[1], mesh[2]])
    gaps = w.solve_on_grid_window(start_k, [row0, 0, 0], mesh)
    ph = np.asarray(w.berry_phase(occ, dir, contin=False))          # (nrows, N_other)
    other = mesh[2] if dir == 1 else mesh[1]
    counts = [p[2] * other for p in plans]
    allv = comm.allgatherv(ph[:own], counts).reshape(mesh[0], other)
    ng = len(gaps)
    allg = comm.allgatherv(np.asarray(gaps, dtype=float), [ng] * world).reshape(world, ng).min(axis=0)
    return allv, allg
