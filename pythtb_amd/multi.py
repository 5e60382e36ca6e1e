"""One-process-per-GPU drivers for the multi-GPU configurations of BASELINE.json:

* solve_all on a k list (configs[1], and the solve_all leg of configs[4]): contiguous chunks of the list
  (reference loop: pythtb.py:1047), ONE gather of the eigenvalues into the band-major (nsta, nkp) array the
  reference returns (pythtb.py:1040,1053-1067) -- `solve_all_sharded`, `solve_all_mesh_sharded`.

* configs[2]: Haldane wf_array([2049, 2049]) solve_on_grid + berry_flux (pythtb.py:2475-2497, :3135-3150): slabs along
  axis 0, each with its recomputed halo row; ONE gather of [partial flux | min gaps] -- `berry_flux_sharded` through the
  wf_array API, `GridSlab` the same slab at the C-ABI level (what bench.py times).
* configs[3]: Kane-Mele wf_array([4097, 513]) -- Wilson-loop eigenphases of the 513 strings along axis 0
  (reference loop: pythtb.py:2987-2996).  Strings shard along axis 1; 513 strings over 8 ranks is an uneven
  split (65 + 7 x 64), so the gather is an all-gather-v.
* configs[4]: cubic16 wf_array([257, 257, 257]) -- solve_on_grid (pythtb.py:2499-2511) and
  berry_phase(range(8), dir=2) (pythtb.py:3002-3025).  Slabs along axis 0, each with its recomputed halo plane;
  a rank reports the planes it owns and the gather assembles the (257, 257) phase array.

Nothing is exchanged while computing: every rank solves its own window of the global mesh
(`wf_array.solve_on_grid_window`; periodic images and halo planes are recomputed, bit-identically).  The one
collective is the gather of the per-rank results at the end, through a `Comm` object:

    RcclComm    tbk_comm_allgatherv_f64 on device buffers (RCCL over xGMI) -- the product path
    SocketComm  the launcher's own TCP rendezvous (launch.Rendezvous: no torch) on host arrays -- hands round RCCL's
                unique id, the one-word agreements, and is the labelled fallback gather when RCCL is unavailable
    GlooComm    torch.distributed (gloo) on host arrays -- the CPU test-suite's stand-in and an optional fallback
                (TBK_RENDEZVOUS=gloo); nothing in the product path needs torch

The drivers take the `wf_array` class to use, so the CPU test-suite can run them (2 ranks, gloo, uneven
counts) with an oracle-backed stand-in where no GPU exists; the library itself never does that.
"""
import ctypes as C
import threading

import numpy as np

from . import shard

__all__ = ["GlooComm", "SocketComm", "RcclComm", "plan_strings", "plan_slabs", "plan_list", "wilson_loops_sharded",
           "mesh_phases_sharded", "solve_all_sharded", "solve_all_mesh_sharded", "call_with_timeout", "agree",
           "rccl_bring_up", "GridSlab", "berry_flux_sharded", "combine_flux_blocks"]


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


def plan_list(n_items, world):
    """Per rank: (begin, end) of its contiguous chunk of a flat k list."""
    return [shard.split_list(n_items, world, r) for r in range(world)]


# ---------------------------------------------------------------------------------------------- communicators
def _displs(counts):
    return [int(x) for x in np.concatenate([[0], np.cumsum(counts)[:-1]])]


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

    def agree(self, ok):
        """True only if `ok` on every rank (a one-word status exchange, not part of the data path)."""
        return agree(self.dist, ok)

    def allgatherv_rows(self, mine, counts):
        """mine (nrows, counts[rank]) on every rank -> (nrows, sum(counts)): rank r's columns at displs[r] (one collective)."""
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        nrows = mine.shape[0]
        flat = self.allgatherv(mine.reshape(-1), [nrows * int(c) for c in counts])
        out = np.empty((nrows, int(sum(counts))))
        pos = 0
        for c, d in zip(counts, _displs(counts)):
            out[:, d:d + c] = flat[pos:pos + nrows * c].reshape(nrows, c)
            pos += nrows * c
        return out


    def gatherv_rows(self, mine, counts, root=0):
        """Rooted form: only `root` gets (nrows, sum(counts)); the other ranks get None and build nothing of that size
        (torch.distributed.gather on padded blocks)."""
        import torch
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        nrows = mine.shape[0]
        assert mine.shape[1] == counts[self.rank]
        cap = max(nrows * max(counts), 1)
        pad = np.zeros(cap)
        pad[:mine.size] = mine.reshape(-1)
        buf = [torch.zeros(cap, dtype=torch.float64) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(torch.from_numpy(pad), buf, dst=root)
        if self.rank != root:
            return None
        out = np.empty((nrows, int(sum(counts))))
        for b, c, d in zip(buf, counts, _displs(counts)):
            out[:, d:d + c] = b.numpy()[:nrows * c].reshape(nrows, c)
        return out


class SocketComm(GlooComm):
    """The same host-array collectives over the launcher's TCP rendezvous (launch.Rendezvous): float64 blocks as byte
    strings through rank 0.  No torch.  Small results only (phases, partial fluxes, status words) -- the fallback when the
    RCCL communicator is unavailable, and what carries RCCL's unique id."""

    def __init__(self, rdzv):
        self.dist = rdzv
        self.rank, self.world = rdzv.rank, rdzv.world
        self.name = "socket"

    def allgatherv(self, mine, counts):
        mine = np.ascontiguousarray(mine, dtype=np.float64).reshape(-1)
        assert mine.size == counts[self.rank]
        parts = self.dist.allgather_bytes(mine.tobytes())
        return np.concatenate([np.frombuffer(p, dtype=np.float64, count=int(c)) for p, c in zip(parts, counts)]) \
            if sum(counts) else np.zeros(0)

    def gatherv_rows(self, mine, counts, root=0):
        """Rooted form: only `root` assembles (nrows, sum(counts)); the other ranks return None."""
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        nrows = mine.shape[0]
        assert mine.shape[1] == counts[self.rank]
        parts = self.dist.allgather_bytes(mine.tobytes())    # (a star through rank 0: the bytes pass there either way)
        if self.rank != root:
            return None
        out = np.empty((nrows, int(sum(counts))))
        for p, c, d in zip(parts, counts, _displs(counts)):
            out[:, d:d + c] = np.frombuffer(p, dtype=np.float64, count=nrows * int(c)).reshape(nrows, int(c))
        return out


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

    def allgatherv_rows_dev(self, send_dev, nrows, counts):
        """send_dev: device pointer of this rank's (nrows, counts[rank]) doubles.  Returns the HOST array
        (nrows, sum(counts)), assembled on the device by ONE grouped exchange (tbk_comm_allgatherv_rows_f64)."""
        _lib, lib, ctx = self._lib, self.lib, self.ctx
        total = int(sum(counts))
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        dsp = np.ascontiguousarray(_displs(counts), dtype=np.int64)
        recv = C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 8 * max(total * nrows, 1), C.byref(recv)))
        try:
            i64p = C.POINTER(C.c_int64)
            _lib.check(lib.tbk_comm_allgatherv_rows_f64(ctx.handle, send_dev, nrows, int(counts[self.rank]), recv,
                                                        cnt.ctypes.data_as(i64p), dsp.ctypes.data_as(i64p), total))
            out = np.zeros((nrows, total))
            if out.size:
                _lib.check(lib.tbk_dev_download(ctx.handle, out.ctypes.data_as(C.c_void_p), recv, out.nbytes))
        finally:
            lib.tbk_dev_free(ctx.handle, recv)
        return out

    def gatherv_rows_dev(self, send_dev, nrows, counts, root=0, download=True):
        """Rooted form of allgatherv_rows_dev (tbk_comm_gatherv_rows_f64): rank `root` returns the HOST array
        (nrows, sum(counts)) -- or, with download=False, the (nrows, 2) array of every row's first and last entry --
        the other ranks return None and allocate nothing of that size."""
        _lib, lib, ctx = self._lib, self.lib, self.ctx
        total = int(sum(counts))
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        dsp = np.ascontiguousarray(_displs(counts), dtype=np.int64)
        recv = C.c_void_p()
        is_root = self.rank == root
        if is_root:
            _lib.check(lib.tbk_dev_alloc(ctx.handle, 8 * max(total * nrows, 1), C.byref(recv)))
        self.last_recv_bytes = 8 * total * nrows if is_root else 0
        try:
            i64p = C.POINTER(C.c_int64)
            _lib.check(lib.tbk_comm_gatherv_rows_f64(ctx.handle, send_dev, nrows, int(counts[self.rank]), recv,
                                                     cnt.ctypes.data_as(i64p), dsp.ctypes.data_as(i64p), total, int(root)))
            if not is_root:
                return None
            if download:
                out = np.zeros((nrows, total))
                if out.size:
                    _lib.check(lib.tbk_dev_download(ctx.handle, out.ctypes.data_as(C.c_void_p), recv, out.nbytes))
                return out
            ends = np.zeros((nrows, 2))
            if total == 0:                                   # nothing gathered: no first / last entry to fetch
                return ends
            for row in range(nrows):
                for j, col in enumerate((0, total - 1)):
                    _lib.check(lib.tbk_dev_download(ctx.handle, ends[row, j:j + 1].ctypes.data_as(C.c_void_p),
                                                    C.c_void_p(recv.value + 8 * (row * total + col)), 8))
            return ends
        finally:
            if recv.value:
                lib.tbk_dev_free(ctx.handle, recv)

    def allgatherv_rows(self, mine, counts):
        _lib, lib, ctx = self._lib, self.lib, self.ctx
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        assert mine.ndim == 2 and mine.shape[1] == counts[self.rank]
        send = C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, max(mine.nbytes, 8), C.byref(send)))
        try:
            if mine.size:
                _lib.check(lib.tbk_dev_upload(ctx.handle, send, mine.ctypes.data_as(C.c_void_p), mine.nbytes))
            return self.allgatherv_rows_dev(send, mine.shape[0], counts)
        finally:
            lib.tbk_dev_free(ctx.handle, send)

    def gatherv_rows(self, mine, counts, root=0):
        _lib, lib, ctx = self._lib, self.lib, self.ctx
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        assert mine.ndim == 2 and mine.shape[1] == counts[self.rank]
        send = C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, max(mine.nbytes, 8), C.byref(send)))
        try:
            if mine.size:
                _lib.check(lib.tbk_dev_upload(ctx.handle, send, mine.ctypes.data_as(C.c_void_p), mine.nbytes))
            return self.gatherv_rows_dev(send, mine.shape[0], counts, root)
        finally:
            lib.tbk_dev_free(ctx.handle, send)

    def close(self):
        self._lib.check(self.lib.tbk_comm_destroy(self.ctx.handle))

    def agree(self, ok):
        """True only if `ok` on every rank: one double per rank through ncclAllGather (status word, not the data path)."""
        return bool(np.all(self.allgather(np.array([1.0 if ok else 0.0])) == 1.0))

    def allgather(self, mine):
        """Equal contributions: ncclAllGather proper (tbk_comm_allgather_f64).  Returns (world, len(mine))."""
        _lib, lib, ctx = self._lib, self.lib, self.ctx
        mine = np.ascontiguousarray(mine, dtype=np.float64).reshape(-1)
        nb = 8 * mine.size
        send, recv = C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nb, C.byref(send)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nb * self.world, C.byref(recv)))
        try:
            _lib.check(lib.tbk_dev_upload(ctx.handle, send, mine.ctypes.data_as(C.c_void_p), nb))
            _lib.check(lib.tbk_comm_allgather_f64(ctx.handle, send, recv, mine.size))
            out = np.zeros((self.world, mine.size))
            _lib.check(lib.tbk_dev_download(ctx.handle, out.ctypes.data_as(C.c_void_p), recv, out.nbytes))
        finally:
            lib.tbk_dev_free(ctx.handle, send)
            lib.tbk_dev_free(ctx.handle, recv)
        return out


# ---------------------------------------------------------------------------------------------- bring-up
def call_with_timeout(fn, seconds):
    """fn() on a worker thread (ctypes releases the GIL inside libtbk / RCCL) -> (ok, value or message, hung).
    A call that never returns is left behind on its daemon thread and reported as hung: the caller carries on with
    its fallback and must finish with os._exit (a stuck collective must not cost the measurement already made)."""
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as e:                      # noqa: BLE001 (reported, not swallowed)
            box["e"] = e
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return False, "no return within %g s" % seconds, True
    if "e" in box:
        return False, " ".join(str(box["e"]).split()), False
    return True, box.get("v"), False


def agree(dist, ok):
    """Every rank takes the same branch: true only if `ok` everywhere -- one word through the launcher's rendezvous
    (launch.Rendezvous.all_min), or a gloo all-reduce when `dist` is torch.distributed."""
    if hasattr(dist, "all_min"):
        return bool(dist.all_min(1 if ok else 0))
    import torch
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def _broadcast_bytes(dist, payload, src=0):
    """`payload` (bytes or None) of rank `src` on every rank, through either kind of rendezvous; None stays None."""
    if hasattr(dist, "broadcast_bytes"):
        out = dist.broadcast_bytes(b"\x01" + payload if payload is not None else b"\x00", src=src)
        return out[1:] if out[:1] == b"\x01" else None
    box = [payload]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def rccl_bring_up(ctx, dist, rank, world, timeout=120.0):
    """The RCCL communicator of a one-process-per-GPU run, brought up FIRST (before any measurement): rank 0 creates the
    id, the rendezvous (the launcher's TCP socket, or gloo) hands it round, every rank joins on a worker thread with a time limit, and the ranks agree on the outcome.
    Returns (RcclComm or None, message, hung)."""
    from . import _lib
    err, uid = "", None
    if rank == 0:
        try:
            buf = (C.c_ubyte * 128)()
            _lib.check(_lib.lib.tbk_comm_unique_id(buf))
            uid = bytes(buf)
        except Exception as e:                          # noqa: BLE001
            err = "unique_id: %s" % " ".join(str(e).split())
    box = [_broadcast_bytes(dist, uid, src=0)]
    ok, comm, hung = box[0] is not None, None, False
    if ok:
        ok, val, hung = call_with_timeout(lambda: RcclComm(ctx, box[0], world, rank), timeout)
        if ok:
            comm = val
        else:
            err = "init: %s" % val
    elif not err:
        err = "unique_id failed on rank 0"
    if not agree(dist, ok):
        return None, err or "another rank failed to join", hung
    return comm, "", hung


# ---------------------------------------------------------------------------------------------- drivers
def _checked_solve_then_agree(comm, solve):
    """Run this rank's solve through the CHECKED entry point (the sticky solver status is read and reset: an iteration
    limit or a NaN model is TBK_ENOCONV here, where the reference's eigvalsh raises -- pythtb.py:939), then let the ranks
    agree on the outcome BEFORE the gather: a failing rank neither leaves the others waiting inside the collective nor
    contributes eigenvalues nobody checked.  Raises on every rank if any rank failed."""
    err = None
    try:
        solve()
    except Exception as e:                              # noqa: BLE001 (re-raised below, after the ranks have agreed)
        err = e
    agree_fn = getattr(comm, "agree", None)             # (recording stand-ins that run the ranks in turn have no peers)
    if agree_fn is None:
        if err is not None:
            raise err
        return
    if not agree_fn(err is None):
        if err is not None:
            raise err
        raise RuntimeError("solve_all (sharded): the eigen-solve failed on another rank; no eigenvalues were gathered")


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
    # ONE collective: every rank's block is [its planes' phases | its min gaps]
    ng = len(gaps)
    counts = [p[2] * other + ng for p in plans]
    flat = comm.allgatherv(np.concatenate([np.asarray(ph[:own], dtype=float).reshape(-1), np.asarray(gaps, dtype=float)]), counts)
    allv, allg, pos = np.empty((mesh[0], other)), np.full(ng, np.inf), 0
    row = 0
    for p, c in zip(plans, counts):
        allv[row:row + p[2]] = flat[pos:pos + p[2] * other].reshape(p[2], other)
        allg = np.minimum(allg, flat[pos + p[2] * other:pos + c])
        row += p[2]
        pos += c
    return allv, allg


class GridSlab(object):
    """One rank's slab of configs[2] at the C-ABI level: a device-resident window [row0, row0 + mesh[0]) x mesh[1:] of a
    global solve_on_grid mesh whose axis 0 has global_n0 points, driven by the asynchronous pair tbk_wfs_solve_grid_async +
    tbk_berry_flux_async (or the one-pass tbk_wfs_solve_grid_flux_async).  Nothing is read back until gaps() / flux_total():
    what bench.py times as a step.  berry_flux_sharded below is the same partitioning through the wf_array API."""

    def __init__(self, lib, _lib, ctx, model, mesh, row0=0, global_n0=None):
        self.lib, self._lib, self.ctx, self.model = lib, _lib, ctx, model
        self.mesh = [int(x) for x in mesh]
        n = model._nsta
        self.n = n
        self.h = C.c_void_p()
        m32 = np.ascontiguousarray(self.mesh, dtype=np.int32)
        _lib.check(lib.tbk_wfs_create(ctx.handle, len(mesh), _lib.iptr(m32), n, n, C.byref(self.h)))
        self.hm = model._device_model()
        self.pbc = np.ascontiguousarray(np.array([np.repeat(np.exp(-2j * np.pi * model._orb[:, model._per[d]]), model._nspin)
                                                  for d in range(len(mesh))]))
        self.row0 = row0
        self.g_n0 = self.mesh[0] if global_n0 is None else global_n0
        self.start = None

    def solve(self, start):
        self.start = np.ascontiguousarray(start, dtype=float)
        self._lib.check(self.lib.tbk_wfs_solve_grid_async(self.h, self.hm, self._lib.dptr(self.start),
                                                          self._lib.dptr(self.pbc.view(float)), self.row0, self.g_n0))

    def flux(self, occ32):
        self._lib.check(self.lib.tbk_berry_flux_async(self.h, self._lib.iptr(occ32), len(occ32), 0, 1, 0))

    def solve_flux(self, start, occ32):
        """solve_on_grid + berry_flux(occ) in one pass (tbk_wfs_solve_grid_flux_async): the array is written once and never read back"""
        self.start = np.ascontiguousarray(start, dtype=float)
        self._lib.check(self.lib.tbk_wfs_solve_grid_flux_async(self.h, self.hm, self._lib.dptr(self.start), self._lib.dptr(self.pbc.view(float)),
                                                               self.row0, self.g_n0, self._lib.iptr(occ32), len(occ32)))

    def gaps(self):
        g = np.zeros(max(self.n - 1, 1))
        self._lib.check(self.lib.tbk_wfs_solve_grid_result(self.h, self._lib.dptr(g)))
        return g

    def flux_total(self, nslices=1):
        t = np.zeros(nslices)
        self._lib.check(self.lib.tbk_berry_flux_result(self.h, self._lib.dptr(t), None))
        return t

    def free(self):
        self._lib.check(self.lib.tbk_wfs_free(self.h))


def combine_flux_blocks(blocks, ng):
    """Rank-ordered combination of the blocks [partial flux | min gaps (ng)] every slab contributes: the flux partial sums are
    added in rank order (a fixed order: the total does not depend on which rank finished first), the gaps min-reduced."""
    blocks = np.asarray(blocks, dtype=float)
    total = 0.0
    for b in blocks:
        total += b[0]
    gaps = blocks[:, 1:1 + ng].min(axis=0) if ng else np.zeros(0)
    return total, gaps


def berry_flux_sharded(wf_array_cls, model, mesh, start_k, occ, comm, rank, world, dirs=None, individual_phases=False):
    """configs[2]: wf_array(model, mesh).solve_on_grid(start_k) + berry_flux(occ, dirs, individual_phases) of a 2-D array
    (pythtb.py:2421-2532, :3135-3150) in slabs along mesh axis 0 (SURVEY.md 8e's first partitioning): every rank solves its
    rows plus ONE halo row -- the next slab's first row, or the periodic image on the last rank, recomputed locally and
    bit-identical because the kernels are deterministic -- takes the flux of its own plaquette rows, and ONE all-gather-v
    carries [partial flux (or this slab's plaquette phases) | min gaps].  Returns (flux, global min gaps): flux is the
    float berry_flux returns, or with individual_phases the (N0-1, N1-1) array (transposed for dirs=[1, 0], like the
    reference's); partial sums are added in rank order."""
    mesh = [int(x) for x in mesh]
    if len(mesh) != 2:
        raise ValueError("berry_flux_sharded drives 2-D arrays")
    dirs = [0, 1] if dirs is None else [int(d) for d in dirs]
    if sorted(dirs) != [0, 1]:
        raise Exception("\n\nDirections must be different and in range!") if len(set(dirs)) < 2 else ValueError("dirs must be [0,1] or [1,0]")
    plans = plan_slabs(mesh[0], world)
    row0, nrows, _ = plans[rank]
    w = wf_array_cls(model, [nrows, mesh[1]])
    gaps = w.solve_on_grid_window(start_k, [row0, 0], mesh)
    gaps = np.zeros(0) if gaps is None else np.asarray(gaps, dtype=float)
    ng = len(gaps)
    part = w.berry_flux(occ, dirs=dirs, individual_phases=individual_phases)
    if not individual_phases:
        blocks = comm.allgatherv(np.concatenate([[float(part)], gaps]), [1 + ng] * world).reshape(world, 1 + ng)
        return combine_flux_blocks(blocks, ng)
    ax = dirs.index(0)                                   # where the slab axis sits in the plaquette array
    mine = np.ascontiguousarray(np.moveaxis(np.asarray(part, dtype=float), ax, 0))      # (own plaquette rows, N1-1)
    ncol = mesh[1] - 1
    counts = [(p[1] - 1) * ncol + ng for p in plans]
    flat = comm.allgatherv(np.concatenate([mine.reshape(-1), gaps]), counts)
    out, allg, pos, row = np.empty((mesh[0] - 1, ncol)), np.full(ng, np.inf), 0, 0
    for p, c in zip(plans, counts):
        nr = p[1] - 1
        out[row:row + nr] = flat[pos:pos + nr * ncol].reshape(nr, ncol)
        allg = np.minimum(allg, flat[pos + nr * ncol:pos + c])
        row += nr
        pos += c
    return np.ascontiguousarray(np.moveaxis(out, 0, ax)), allg


def solve_all_sharded(model, k_list, comm, rank, world, solve_chunk=None, root=None):
    """tb_model.solve_all(k_list) (eigenvalues; pythtb.py:955-1067) with the k list cut into `world` contiguous
    chunks: every rank solves its own chunk and ONE gather assembles ret_eval (nsta, nkp), band-major like
    the reference's (pythtb.py:1040).  root=None: all-gather-v, every rank returns the array.  root=r: the ROOTED
    gather-v (ret_eval is one array on one caller) -- rank r returns it, the others return None and
    never allocate an (nsta, nkp) buffer.  With an RcclComm everything between the k chunk's upload and the final
    download stays on the device (tbk_solve_list_dev_checked -> tbk_comm_[all]gatherv_rows_f64).  `solve_chunk(k) ->
    (nsta, nk)` replaces the device solve in the CPU test-suite (an oracle-backed stand-in); the library never does."""
    k = np.asarray(k_list, dtype=float)
    if k.ndim == 1:                                            # a flat list of scalar k for dim_k = 1 (pythtb.py:1036)
        k = k.reshape(-1, 1)
    k = np.ascontiguousarray(k)
    nk = k.shape[0]
    plans = plan_list(nk, world)
    counts = [e - b for b, e in plans]
    b, e = plans[rank]
    if solve_chunk is not None or not hasattr(comm, "allgatherv_rows_dev"):
        box = {}

        def host_solve():
            box["ev"] = (solve_chunk or model.solve_all)(k[b:e]) if e > b else np.zeros((model._nsta, 0))
        _checked_solve_then_agree(comm, host_solve)    # (tb_model.solve_all is the checked call)
        mine_ev = np.asarray(box["ev"], dtype=float).reshape(model._nsta, e - b)
        if root is not None:
            return comm.gatherv_rows(mine_ev, counts, root)
        return comm.allgatherv_rows(mine_ev, counts)
    from . import _lib
    lib, ctx, n = _lib.lib, comm.ctx, model._nsta
    hm = model._device_model()
    kd, ed = C.c_void_p(), C.c_void_p()
    mine = k[b:e]
    _lib.check(lib.tbk_dev_alloc(ctx.handle, max(mine.nbytes, 8), C.byref(kd)))
    _lib.check(lib.tbk_dev_alloc(ctx.handle, max(8 * n * (e - b), 8), C.byref(ed)))
    try:
        def dev_solve():
            if e > b:
                _lib.check(lib.tbk_dev_upload(ctx.handle, kd, mine.ctypes.data_as(C.c_void_p), mine.nbytes))
                _lib.check(lib.tbk_solve_list_dev_checked(hm, kd, e - b, ed, None))
        _checked_solve_then_agree(comm, dev_solve)
        if root is not None:
            return comm.gatherv_rows_dev(ed, n, counts, root)
        return comm.allgatherv_rows_dev(ed, n, counts)
    finally:
        lib.tbk_dev_free(ctx.handle, kd)
        lib.tbk_dev_free(ctx.handle, ed)


def solve_all_mesh_sharded(model, mesh_size, comm, rank, world, download=True, root=None, stats=None):
    """solve_all(k_uniform_mesh(mesh_size)) (pythtb.py:1792-1861, :955-1067), k-sharded, with every rank generating its
    chunk of the k list on the device (tbk_k_uniform_mesh_range_dev): nothing but the gathered eigenvalues leaves a GPU.
    Needs an RcclComm.  Returns ret_eval (nsta, prod(mesh_size)); with download=False the gather still runs (it is what
    is being measured) and only a (nsta, 2) array of every band's first and last eigenvalue comes back to the host.
    root=r: rooted gather-v -- rank r returns the array, the others None and they allocate no (nsta, nk) buffer.
    `stats` (a dict) receives this rank's solve_ms, gather_ms, sent_bytes and recv_bytes."""
    import time
    from . import _lib
    lib, ctx, n = _lib.lib, comm.ctx, model._nsta
    mesh = np.ascontiguousarray(mesh_size, dtype=np.int32).reshape(-1)
    if mesh.size != model._dim_k:
        raise Exception("\n\nmesh_size must have dim_k entries")
    nk = int(np.prod(mesh.astype(np.int64)))
    plans = plan_list(nk, world)
    counts = [e - b for b, e in plans]
    b, e = plans[rank]
    hm = model._device_model()
    kd, ed, recv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.tbk_dev_alloc(ctx.handle, max(8 * model._dim_k * (e - b), 8), C.byref(kd)))
    _lib.check(lib.tbk_dev_alloc(ctx.handle, max(8 * n * (e - b), 8), C.byref(ed)))
    try:
        def dev_solve():
            _lib.check(lib.tbk_k_uniform_mesh_range_dev(ctx.handle, model._dim_k, _lib.iptr(mesh), b, e - b, kd))
            if e > b:
                _lib.check(lib.tbk_solve_list_dev_checked(hm, kd, e - b, ed, None))
        t0 = time.perf_counter()
        _checked_solve_then_agree(comm, dev_solve)
        t1 = time.perf_counter()
        if stats is not None:
            stats.update(solve_ms=(t1 - t0) * 1e3, sent_bytes=8 * n * (e - b) * (1 if root is not None else world),
                         recv_bytes=8 * n * nk if root is None or root == rank else 0)
        if root is not None:
            out = comm.gatherv_rows_dev(ed, n, counts, root, download=download)
            if stats is not None:
                stats["gather_ms"] = (time.perf_counter() - t1) * 1e3
            return out
        if download:
            out = comm.allgatherv_rows_dev(ed, n, counts)
            if stats is not None:
                stats["gather_ms"] = (time.perf_counter() - t1) * 1e3
            return out
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 8 * n * nk, C.byref(recv)))
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        dsp = np.ascontiguousarray(_displs(counts), dtype=np.int64)
        i64p = C.POINTER(C.c_int64)
        _lib.check(lib.tbk_comm_allgatherv_rows_f64(ctx.handle, ed, n, e - b, recv, cnt.ctypes.data_as(i64p),
                                                    dsp.ctypes.data_as(i64p), nk))
        if stats is not None:
            stats["gather_ms"] = (time.perf_counter() - t1) * 1e3
        ends = np.zeros((n, 2))
        for band in range(n):
            for j, col in enumerate((0, nk - 1)):
                _lib.check(lib.tbk_dev_download(ctx.handle, ends[band, j:j + 1].ctypes.data_as(C.c_void_p),
                                                C.c_void_p(recv.value + 8 * (band * nk + col)), 8))
        return ends
    finally:
        for ptr in (kd, ed, recv):
            if ptr.value:
                lib.tbk_dev_free(ctx.handle, ptr)
