"""wf_array: mesh of wavefunctions kept resident in MI355X HBM.

Mirror of PythTB's `wf_array` (pythtb.py:2283-3205) for the hot path:
`solve_on_grid`, `impose_pbc/impose_loop`, `berry_phase`, `berry_flux`, `[]`,
`choose_states`, `empty_like`, `solve_on_one_point`.  The array lives on the
device between `solve_on_grid` and the Berry calls; `_wfs` is a lazily
materialised NumPy mirror (reading it, or using `wf[i,j]`, hands the host copy
to the caller, after which the device copy is refreshed before its next use).
"""
import copy
import ctypes as C
import sys

import numpy as np

from . import _lib
from .model import _is_int

__all__ = ["wf_array"]

_TWO_PI = 2.0 * np.pi


def _no_2pi(x, clos):
    """Bring x within pi of clos by 2 pi steps (pythtb.py:3867-3874)."""
    return x - _TWO_PI * np.round((x - clos) / _TWO_PI) if abs(clos - x) > np.pi else x


def _one_phase_cont(pha, clos):
    """Sequential 2 pi unwrapping anchored at `clos` (pythtb.py:3876-3889); the loop runs in libtbk (tbk_one_phase_cont)."""
    arr = np.ascontiguousarray(pha, dtype=float)
    out = np.empty_like(arr)
    _lib.check(_lib.lib.tbk_one_phase_cont(_lib.dptr(arr), arr.shape[0], 1, float(clos), _lib.dptr(out), 1))
    return out


def _array_phases_cont(arr_pha, clos):
    """Greedy nearest matching of eigenphase sets on the unit circle along the first index, then 2 pi unwrapping
    (pythtb.py:3891-3921; the reference keeps the LAST index among equal minima).  The double loop runs in libtbk
    (tbk_array_phases_cont): as a Python loop it was 250 of the 310 us of a 41-string, two-band berry_phase call."""
    arr = np.ascontiguousarray(arr_pha, dtype=float)
    ref = np.ascontiguousarray(clos, dtype=float)
    out = np.empty_like(arr)
    _lib.check(_lib.lib.tbk_array_phases_cont(_lib.dptr(arr), arr.shape[0], arr.shape[1], arr.shape[1], _lib.dptr(ref),
                                              _lib.dptr(out), arr.shape[1]))
    return out


class _PointLedger(object):
    """Points of a device-resident wf_array handed out by wf[i,j] as WRITABLE arrays (pythtb.py:2662-2666 returns a live
    view of the storage).  Each point has a snapshot of what the device held when it was handed out (or last synchronised);
    `changed()` is one vectorised bit-compare per chunk of 256 points.  With `own` the ledger also owns the memory the
    caller's arrays view (chunks that are never reallocated), so a write made through a temporary -- `wf[i,j][0] *= z` --
    is still there when the next device use looks; otherwise the live rows are rows of the host mirror.
    A chunk is a record [snap, live or None, ids]; `slot[fi]` = (record, row)."""
    CHUNK = 256

    def __init__(self, pt_shape, own):
        self.pt_shape, self.own = tuple(pt_shape), own
        self.slot, self.chunks, self.n = {}, [], 0

    def __len__(self):
        return self.n

    def __contains__(self, fi):
        return fi in self.slot

    @property
    def idx(self):
        return [fi for ch in self.chunks for fi in ch[2]]

    def add(self, fi, value):
        if not self.chunks or len(self.chunks[-1][2]) == self.CHUNK:
            self.chunks.append([np.empty((self.CHUNK,) + self.pt_shape, dtype=complex),
                                np.empty((self.CHUNK,) + self.pt_shape, dtype=complex) if self.own else None, []])
        ch = self.chunks[-1]
        r = len(ch[2])
        ch[0][r] = value
        if self.own:
            ch[1][r] = value
        ch[2].append(fi)
        self.slot[fi] = (ch, r)
        self.n += 1

    def view(self, fi):
        ch, r = self.slot[fi]
        return ch[1][r]

    def set(self, fi, value):
        """The storage at this point was assigned: snapshot (and the caller's arrays of it) follow."""
        ch, r = self.slot[fi]
        ch[0][r] = value
        if self.own:
            ch[1][r] = value

    def indices(self):
        return np.array(self.idx, dtype=np.int64)

    def changed(self, rows=None):
        """(flat indices, values) of the points whose live array differs bit for bit from its snapshot; the snapshots
        are brought up to date.  `rows`: the mirror as [point][...] for a ledger that does not own its live rows."""
        out_i, out_v = [], []
        for snap, live, ids in self.chunks:
            m = len(ids)
            cid = np.array(ids, dtype=np.int64)
            live = live[:m] if self.own else rows[cid]
            diff = np.flatnonzero((np.ascontiguousarray(live).view(np.uint64).reshape(m, -1)
                                   != snap[:m].view(np.uint64).reshape(m, -1)).any(axis=1))
            if len(diff):
                out_i.append(cid[diff])
                out_v.append(np.array(live[diff]))
                snap[diff] = live[diff]
        if not out_i:
            return np.zeros(0, dtype=np.int64), None
        return np.concatenate(out_i), np.concatenate(out_v)

    def refresh(self, pos, buf, rows=None):
        """The device copy was rewritten: live rows and snapshots take buf[pos[fi]]."""
        for snap, live, ids in self.chunks:
            m = len(ids)
            src = buf[[pos[int(i)] for i in ids]]
            snap[:m] = src
            if self.own:
                live[:m] = src
            else:
                rows[np.array(ids, dtype=np.int64)] = src

    def recycle(self):
        """Forget the pool chunks no array outside the ledger views any more; returns the number of points dropped.
        To be called right after `changed()` was carried to the device (so nothing a temporary wrote is lost).  Every
        array handed out -- and every view derived from one: NumPy collapses `.base` to the owner of the memory -- holds a
        reference to its CHUNK, so a chunk whose only referrer is this ledger has no live handle left: a read loop
        `for i, j: x = wf[i, j]` keeps at most the chunk of its last point (without this the loop ran the pool
        to its cap and then gave up device residency for good)."""
        if not self.own:
            return 0
        keep, dropped = [], 0
        while self.chunks:
            ch = self.chunks.pop()
            live = ch[1]
            ch[1] = None                                    # referrers now: `live` here + getrefcount's argument (+ user views)
            if sys.getrefcount(live) > 2:
                ch[1] = live
                keep.append(ch)
            else:
                for fi in ch[2]:
                    del self.slot[fi]
                dropped += len(ch[2])
            del live
        keep.reverse()
        # (a kept chunk that is not full stays as it is -- its rows never move; add() opens a new chunk after it)
        self.chunks = keep
        self.n -= dropped
        return dropped


class wf_array(object):
    """Array of wavefunctions on a (k or parameter) mesh: _wfs[k1..kD, state, orb(,spin)]."""

    def __init__(self, model, mesh_arr, nsta_arr=None):
        if nsta_arr is None:
            self._nsta_arr = model._nsta
        else:
            if not _is_int(nsta_arr):
                raise Exception("\n\nArgument nsta_arr not an integer")
            self._nsta_arr = nsta_arr
        self._nspin = model._nspin
        self._norb = model._norb
        self._orb = np.copy(model._orb)
        self._model = copy.deepcopy(model)
        self._mesh_arr = np.array(mesh_arr)
        self._dim_arr = len(self._mesh_arr)
        if True in (self._mesh_arr <= 1).tolist():
            raise Exception("\n\nDimension of wf_array object in each direction must be 2 or larger.")
        self._host = None          # NumPy mirror (allocated on first use)
        self._host_valid = False
        self._host_exported = False  # the writable mirror has been handed out through `_wfs` (sticky)
        self._dev = None           # tbk_wfs handle
        self._dev_shape = None
        self._dev_valid = False
        self._pt_views = None      # _PointLedger of points handed out as writable views of the mirror
        self._pt_copies = None     # _PointLedger of points handed out as writable arrays of its own pool (large resident arrays)

    # ------------------------------------------------------------------ storage
    # Two copies of the array exist: the device buffer (what every kernel reads and writes) and a lazily
    # created NumPy mirror.  Who is authoritative:
    #   * device, after solve_on_grid / impose_* and as long as the caller only uses wf[i,j], wf[i,j] = v
    #     (single-point upload), berry_*, position_*.  wf[i,j] is WRITABLE like the reference's view
    #     (pythtb.py:2662-2666): the points handed out are remembered with a snapshot (_PointLedger), compared -- one
    #     vectorised pass, only before the next device use or export, never per wf[i,j] -- and the changed ones uploaded
    #     (_sync_point_writes): the points, never the array;
    #   * host, once the caller has taken the writable mirror through the private attribute `_wfs` (in the
    #     reference that attribute IS the storage, so a script may keep the array and write to it at any
    #     time): from then on the mirror is re-uploaded before every device use and refreshed after every
    #     device-side write, until release_host() is called.
    _SMALL_MIRROR_BYTES = 32 << 20    # wf[i,j] on a resident array at most this big mirrors it whole

    def _shape(self, nsta=None):
        if nsta is None:
            c = self.__dict__.get("_shape0")
            if c is not None and c[0] == self._nsta_arr:
                return c[1]
        shp = [int(x) for x in self._mesh_arr] + [int(self._nsta_arr if nsta is None else nsta), self._norb]
        if self._nspin == 2:
            shp.append(2)
        shp = tuple(shp)
        if nsta is None:
            self._shape0 = (self._nsta_arr, shp)      # (the mesh and the orbitals are fixed at construction)
        return shp

    def _host_array(self):
        """Host mirror, brought up to date, without giving up the device copy."""
        if self._host is None:
            self._host = np.zeros(self._shape(), dtype=complex)
            if not self._dev_valid:
                self._host_valid = True
        if not self._host_valid:
            if self._dev_valid:
                self._sync_point_writes()      # (handed-out rows of the stale mirror may hold writes the download would bury)
                _lib.check(_lib.lib.tbk_wfs_download(self._dev, _lib.dptr(self._host.view(float))))
            self._host_valid = True
        return self._host

    @property
    def _wfs(self):
        self._sync_point_writes()
        arr = self._host_array()
        self._host_exported = True         # the caller may keep the array and write through it at any time
        self._pt_views = None              # (the whole mirror is live now: no per-point bookkeeping)
        return arr

    @_wfs.setter
    def _wfs(self, value):
        self._host = np.ascontiguousarray(value, dtype=complex)
        self._host_valid = True
        self._host_exported = False
        self._dev_valid = False
        self._pt_views = None
        self._pt_copies = None

    # ---- writable wf[i,j] on a device-resident array (pythtb.py:2662-2666 returns a live view of the storage)
    _PT_TRACK_MAX = 1 << 16           # handed-out points followed one by one before the whole mirror becomes the live copy
    _PT_TRACK_BYTES = 256 << 20       # ... or this many bytes of them

    def _pt_cap(self):
        return max(64, min(self._PT_TRACK_MAX, self._PT_TRACK_BYTES // (16 * int(np.prod(self._shape()[self._dim_arr:])))))

    def _point_rows(self):
        """The mirror as [point][state, orb(, spin)] (a view)."""
        return self._host.reshape((-1,) + self._shape()[self._dim_arr:])

    def _upload_points(self, idx, vals):
        if self._dev_valid and self._dev is not None and len(idx):
            pts = np.ascontiguousarray(vals, dtype=complex)
            ia = np.ascontiguousarray(idx, dtype=np.int64)
            _lib.check(_lib.lib.tbk_wfs_upload_points(self._dev, ia.ctypes.data_as(C.POINTER(C.c_int64)), len(ia),
                                                      _lib.dptr(pts.view(float))))

    def _sync_point_writes(self):
        """Carry writes made through arrays obtained from wf[i,j] into the authoritative copy: every handed-out
        point is compared with the snapshot taken when it was handed out (or last synchronised).  Called before the
        device copy is used, exported or overwritten by a download -- not per wf[i,j]."""
        if self._pt_views is not None:
            # (after a device-side write the mirror as a whole is stale, but its handed-out points were refreshed)
            if self._host is not None and not self._host_exported:
                idx, vals = self._pt_views.changed(self._point_rows())
                self._upload_points(idx, vals)
            else:
                self._pt_views = None
        if self._pt_copies is not None:
            idx, vals = self._pt_copies.changed()
            if len(idx):
                self._upload_points(idx, vals)
                if self._host is not None and self._host_valid:
                    self._point_rows()[idx] = vals

    def _refresh_handed_points(self):
        """A kernel rewrote the device copy: arrays the caller still holds from wf[i,j] show the new values,
        like the reference's views of its storage."""
        ledgers = [l for l in (self._pt_views, self._pt_copies) if l is not None and len(l)]
        if self._pt_views is not None and self._host is None:
            ledgers = [l for l in ledgers if l is not self._pt_views]
        idxs = sorted(set(i for l in ledgers for i in l.idx))
        if not idxs:
            return
        ia = np.ascontiguousarray(idxs, dtype=np.int64)
        buf = np.zeros((len(idxs),) + self._shape()[self._dim_arr:], dtype=complex)
        _lib.check(_lib.lib.tbk_wfs_download_points(self._dev, ia.ctypes.data_as(C.POINTER(C.c_int64)), len(idxs),
                                                    _lib.dptr(buf.view(float))))
        pos = {i: n for n, i in enumerate(idxs)}
        for l in ledgers:
            l.refresh(pos, buf, None if l.own else self._point_rows())

    def mark_dirty(self):
        """Extension: declare that the host mirror was modified in place (only needed after
        release_host(); while the mirror is exported every device use re-uploads it anyway)."""
        if self._host is not None and self._host_valid:
            self._dev_valid = False

    def release_host(self):
        """Extension: promise that no array obtained from `_wfs` will be written to any more.  The device
        copy becomes authoritative again and Berry calls stop re-uploading the mirror."""
        self._host_exported = False

    def to_host(self):
        """Extension: read-only NumPy snapshot of the whole array (one download, no later re-uploads)."""
        self._sync_point_writes()
        out = self._host_array().view()
        out.flags.writeable = False
        return out

    def _device_wrote(self):
        """A kernel has just changed the device copy."""
        self._dev_valid = True
        self._host_valid = False
        if self._pt_views is not None or self._pt_copies is not None:
            if tuple(self._dev_shape or ()) == self._shape():
                if self._host_exported:
                    self._pt_views = None
                self._refresh_handed_points()
            else:
                self._pt_views, self._pt_copies = None, None
        if self._host_exported:            # keep the array the caller holds live, like the reference's storage
            self._host_array()

    def _free_dev(self):
        if self._dev is not None:
            _lib.lib.tbk_wfs_free(self._dev)
        self._dev = None
        self._dev_shape = None
        self._dev_valid = False

    def _dev_handle(self, shape):
        """Device buffer of the given full shape (re-created when the shape changes)."""
        if self._dev is not None and self._dev_shape == tuple(shape):
            return self._dev
        self._free_dev()
        mesh = np.ascontiguousarray(shape[:self._dim_arr], dtype=np.int32)
        nsta = int(shape[self._dim_arr])
        ncomp = int(np.prod(shape[self._dim_arr + 1:]))
        h = C.c_void_p()
        _lib.check(_lib.lib.tbk_wfs_create(_lib.default_context().handle, self._dim_arr, _lib.iptr(mesh),
                                           nsta, ncomp, C.byref(h)))
        self._dev = h
        self._dev_shape = tuple(shape)
        return h

    def _ensure_dev(self):
        self._sync_point_writes()
        if self._dev_valid and self._dev is not None and not (self._host_exported and self._host_valid):
            return self._dev
        host = self._host_array()
        if not host.flags["C_CONTIGUOUS"]:
            host = self._host = np.ascontiguousarray(host)
        h = self._dev_handle(host.shape)
        _lib.check(_lib.lib.tbk_wfs_upload(h, _lib.dptr(host.view(float))))
        self._dev_valid = True
        return h

    def __getstate__(self):
        self._sync_point_writes()
        st = dict(self.__dict__)
        st["_host"] = None if self._host is None and not self._dev_valid else np.copy(self._host_array())
        st["_host_valid"] = st["_host"] is not None
        st["_dev"] = None
        st["_dev_shape"] = None
        st["_dev_valid"] = False
        st["_host_exported"] = False
        st["_pt_views"] = None
        st["_pt_copies"] = None
        st.pop("_bufs", None)          # (ctypes pointers of the per-call buffers: rebuilt on first use)
        st.pop("_pbc_cache", None)
        return st

    def __del__(self):
        try:
            self._free_dev()
        except Exception:
            pass

    # ------------------------------------------------------------------ mesh solve
    def solve_on_grid(self, start_k):
        """Solve the model on the regular mesh anchored at start_k and impose
        periodic boundary conditions (pythtb.py:2421-2532).  One kernel launch;
        the eigenvectors stay on the device.  Returns the minimal direct gaps."""
        if self._dim_arr != self._model._dim_k:
            raise Exception("\n\nIf using solve_on_grid method, dimension of wf_array must equal"
                            "\ndim_k of the tight-binding model!")
        if self._nsta_arr != self._model._nsta:
            raise Exception(
                "\n\nWhen initializing this object, you specified nsta_arr to be " + str(self._nsta_arr) + ", but"
                "\nthis does not match the total number of bands specified in the model,"
                "\nwhich was " + str(self._model._nsta) + ".  If you wish to use the solve_on_grid method, do"
                "\nnot specify the nsta_arr parameter when initializing this object.\n\n")
        self._start_k = start_k
        m = self._model
        sk = np.array(start_k, dtype=float)
        if sk.size != self._dim_arr:
            raise Exception("\n\nk-vector of wrong shape!")
        n = m._nsta
        # (the small argument / result buffers of this call and their ctypes pointers live with the array: building the
        # pointer objects anew was 4 us of every call, profiles/api_call_profile.py)
        b = self._call_bufs(n)
        b["start"][:] = sk.reshape(-1)
        pbc_p = self._pbc_phases(True)
        h = self._dev_handle(self._shape())
        _lib.check(_lib.lib.tbk_wfs_solve_grid(h, m._device_model(), b["start_p"], pbc_p, 0, int(self._mesh_arr[0]), b["gaps_p"]))
        self._device_wrote()
        if n <= 1:
            return None
        return b["gaps"][:n - 1].copy()

    def _call_bufs(self, n):
        b = getattr(self, "_bufs", None)
        if b is None or b["n"] != n:
            start, gaps = np.zeros(self._dim_arr, dtype=float), np.zeros(max(n - 1, 1), dtype=float)
            b = self._bufs = {"n": n, "start": start, "gaps": gaps, "start_p": _lib.dptr(start), "gaps_p": _lib.dptr(gaps),
                              "totals": {}, "occ": {}}
        return b

    def _pbc_phases(self, pointer=False):
        """exp(-2 pi i orb[:, per[d]]) per mesh axis and state (pythtb.py:2729), kept between calls (the orbital positions of
        a wf_array are copied at construction; `_per` and the positions are compared)."""
        m = self._model
        c = getattr(self, "_pbc_cache", None)
        if c is not None and c[0] == tuple(m._per[:self._dim_arr]) and c[1] == self._orb.tobytes() and c[2] == self._nspin:
            return c[4] if pointer else c[3]
        pbc = np.zeros((self._dim_arr, m._nsta), dtype=complex)
        for d in range(self._dim_arr):
            pbc[d] = np.repeat(np.exp(-2.j * np.pi * self._orb[:, m._per[d]]), self._nspin)
        self._pbc_cache = (tuple(m._per[:self._dim_arr]), self._orb.tobytes(), self._nspin, pbc, _lib.dptr(pbc.view(float)))
        return self._pbc_cache[4] if pointer else pbc

    def solve_on_grid_flux(self, start_k, occ="All"):
        """Extension (not in the reference): `solve_on_grid(start_k)` and `berry_flux(occ)` of a 2-D array in ONE pass over
        the mesh -- the plaquette phases are formed while the eigenvectors are still in registers, so the array is written
        once and never read back (tbk_wfs_solve_grid_flux_async).  Returns (min gaps, total flux), the values the two
        calls return (the flux up to the order of its sum).  Where the fused kernel does not apply (other than 2-D arrays
        of 2 or 4 states, more than two bands) the two calls are made."""
        from ._lib import TbkError
        m = self._model
        occ_arr = self._occ(occ)
        if occ_arr.ndim == 1:
            occ_arr = self._wrap_occ(occ_arr)
        if (self._dim_arr == 2 and self._dim_arr == m._dim_k and self._nsta_arr == m._nsta and m._nsta in (2, 4)
                and 1 <= len(occ_arr) <= 2):
            start = np.ascontiguousarray(np.array(start_k, dtype=float).reshape(-1))
            if start.shape != (self._dim_arr,):
                raise Exception("\n\nk-vector of wrong shape!")
            n = m._nsta
            pbc = np.zeros((self._dim_arr, n), dtype=complex)
            for d in range(self._dim_arr):
                pbc[d] = np.repeat(np.exp(-2.j * np.pi * self._orb[:, m._per[d]]), self._nspin)
            h = self._dev_handle(self._shape())
            occ32 = np.ascontiguousarray(occ_arr, dtype=np.int32)
            rc = _lib.lib.tbk_wfs_solve_grid_flux_async(h, m._device_model(), _lib.dptr(start), _lib.dptr(pbc.view(float)), 0,
                                                       int(self._mesh_arr[0]), _lib.iptr(occ32), len(occ32))
            if rc == 0:
                gaps = np.zeros(max(n - 1, 1), dtype=float)
                _lib.check(_lib.lib.tbk_wfs_solve_grid_result(h, _lib.dptr(gaps)))
                tot = np.zeros(1)
                _lib.check(_lib.lib.tbk_berry_flux_result(h, _lib.dptr(tot), None))
                self._start_k = start_k
                self._device_wrote()
                return gaps[:n - 1], float(tot[0])
            if rc != 4:                                          # TBK_EUNSUPPORTED: fall through to the two calls
                _lib.check(rc)
        gaps = self.solve_on_grid(start_k)
        return gaps, self.berry_flux(occ)

    def solve_on_grid_window(self, start_k, offset, global_mesh):
        """Extension for k-sharded runs (not in the reference): this array holds the points
        [offset[d], offset[d]+mesh[d]) of a global solve_on_grid mesh of global_mesh[d] points per
        axis; every point is computed locally (periodic images and halo rows are recomputed, so
        no rank talks to another).  Returns the minimal gaps over this window."""
        m = self._model
        if self._dim_arr != m._dim_k or self._nsta_arr != m._nsta:
            raise Exception("\n\nsolve_on_grid_window needs a full-band wf_array with dim_arr == dim_k")
        start = np.ascontiguousarray(np.array(start_k, dtype=float).reshape(-1))
        off = np.ascontiguousarray(offset, dtype=np.int64)
        gm = np.ascontiguousarray(global_mesh, dtype=np.int64)
        if start.shape != (self._dim_arr,) or off.shape != (self._dim_arr,) or gm.shape != (self._dim_arr,):
            raise Exception("\n\nk-vector of wrong shape!")
        n = m._nsta
        pbc = np.zeros((self._dim_arr, n), dtype=complex)
        for d in range(self._dim_arr):
            pbc[d] = np.repeat(np.exp(-2.j * np.pi * self._orb[:, m._per[d]]), self._nspin)
        h = self._dev_handle(self._shape())
        i64p = C.POINTER(C.c_int64)
        _lib.check(_lib.lib.tbk_wfs_solve_window_async(h, m._device_model(), _lib.dptr(start), _lib.dptr(pbc.view(float)),
                                                      off.ctypes.data_as(i64p), gm.ctypes.data_as(i64p)))
        gaps = np.zeros(max(n - 1, 1), dtype=float)
        _lib.check(_lib.lib.tbk_wfs_solve_grid_result(h, _lib.dptr(gaps)))
        self._start_k = start_k
        self._device_wrote()
        return None if n <= 1 else gaps[:n - 1]

    def solve_on_one_point(self, kpt, mesh_indices):
        """Solve at one k and store at mesh_indices (pythtb.py:2534-2566)."""
        (eval, evec) = self._model.solve_one(kpt, eig_vectors=True)
        if _is_int(mesh_indices):
            if self._dim_arr > 1:          # the reference broadcasts over the remaining axes (pythtb.py:2563-2564)
                self._wfs[(mesh_indices,)] = evec
                return
            key = mesh_indices
        else:
            key = tuple(mesh_indices)
            if self._dim_arr == 1 and len(key) == 1:
                key = key[0]
        self[key] = evec

    def _clone_meta(self):
        """A wf_array with this one's model, mesh and bookkeeping but no storage (deepcopy would first download a
        resident array -- 69.5 GB for BASELINE configs[4] -- only to throw the copy away)."""
        new = wf_array.__new__(wf_array)
        for k, v in self.__dict__.items():
            if k in ("_host", "_dev", "_dev_shape", "_pt_views", "_pt_copies", "_bufs", "_pbc_cache"):
                continue
            new.__dict__[k] = copy.deepcopy(v)
        new._host = None
        new._host_valid = False
        new._host_exported = False
        new._dev = None
        new._dev_shape = None
        new._dev_valid = False
        new._pt_views = None
        new._pt_copies = None
        return new

    def choose_states(self, subset):
        """New wf_array holding a subset of the states (pythtb.py:2568-2608).  On a resident array the chosen band
        planes are copied device to device; nothing crosses PCIe."""
        subset = np.array(subset, dtype=int)
        if subset.ndim != 1:
            raise Exception("\n\nParameter subset must be a one-dimensional array.")
        if self._dim_arr > 4:
            raise Exception("\n\n_dim_array too large.")
        self._sync_point_writes()
        new = self._clone_meta()
        new._nsta_arr = subset.shape[0]
        dev_current = self._dev_valid and self._dev is not None and not (self._host_exported and self._host_valid)
        if dev_current and subset.size >= 1:
            if np.any(subset < -self._nsta_arr) or np.any(subset >= self._nsta_arr):
                raise IndexError("index out of bounds in choose_states")      # np.take's error class in the reference
            sub32 = np.ascontiguousarray(subset % self._nsta_arr, dtype=np.int32)
            h = new._dev_handle(new._shape())
            _lib.check(_lib.lib.tbk_wfs_copy_bands(h, self._dev, _lib.iptr(sub32), len(sub32)))
            new._dev_valid = True
        else:
            new._wfs = np.take(self._host_array(), subset, axis=self._dim_arr)
            new._host_exported = False
        return new

    def empty_like(self, nsta_arr=None):
        """Same-shaped uninitialised wf_array, optionally with another number of states
        (pythtb.py:2610-2642)."""
        new = self._clone_meta()
        if nsta_arr is not None:
            new._nsta_arr = nsta_arr
        new._wfs = np.empty(new._shape(), dtype=complex)
        new._host_exported = False
        return new

    def _check_key(self, key):
        if self._dim_arr == 1:
            if not _is_int(key):
                raise TypeError("Key should be an integer!")
            if key < -self._mesh_arr[0] or key >= self._mesh_arr[0]:
                raise IndexError("Key outside the range!")
        else:
            if len(key) != self._dim_arr:
                raise TypeError("Wrong dimensionality of key!")
            for i, k in enumerate(key):
                if not _is_int(k):
                    raise TypeError("Key should be set of integers!")
                if k < -self._mesh_arr[i] or k >= self._mesh_arr[i]:
                    raise IndexError("Key outside the range!")

    def _flat_index(self, key):
        key = (key,) if self._dim_arr == 1 else tuple(key)
        idx = 0
        for d, k in enumerate(key):
            n = int(self._mesh_arr[d])
            idx = idx * n + (int(k) + n if k < 0 else int(k))
        return idx

    def _device_only(self):
        """The device copy is current and the host mirror is not."""
        return self._dev_valid and self._dev is not None and not (self._host_valid and self._host is not None)

    def __getitem__(self, key):
        """States at one mesh point, `(nsta_arr, norb[, 2])`, WRITABLE like the reference's view of its storage
        (pythtb.py:2644-2666: `wf[i,j][0] *= phase` changes the array).  A view of the host mirror when there is
        one, else -- for a large resident array -- just this point fetched from the device into a pool the array owns
        (asking for the same point again returns a view of the same memory, as in the reference); either way the
        point is remembered with a snapshot and writes made through the returned array reach the device copy before
        its next use (_sync_point_writes).  Nothing is compared here: a read loop over the mesh is linear."""
        self._check_key(key)
        pt_shape = self._shape()[self._dim_arr:]
        if (self._device_only() and not self._host_exported
                and int(np.prod(self._shape())) * 16 > self._SMALL_MIRROR_BYTES):
            fi = self._flat_index(key)
            led = self._pt_copies
            if led is not None and fi in led:
                return led.view(fi)
            if led is not None and len(led) >= self._pt_cap():
                # the pool is full: carry what was written through it to the device, then forget the chunks nobody holds an
                # array of any more (a read loop over the mesh stays resident: no whole-array download, no sticky export)
                self._sync_point_writes()
                led.recycle()
            if led is None or len(led) < self._pt_cap():
                if self._pt_views is not None and fi in self._pt_views and self._host is not None:
                    return self._host[key]                     # (handed out as a mirror row earlier: that row is the live one)
                out = np.zeros(pt_shape, dtype=complex)
                idx = np.array([fi], dtype=np.int64)
                _lib.check(_lib.lib.tbk_wfs_download_points(self._dev, idx.ctypes.data_as(C.POINTER(C.c_int64)), 1,
                                                            _lib.dptr(out.view(float))))
                if led is None:
                    led = self._pt_copies = _PointLedger(pt_shape, own=True)
                led.add(fi, out)
                return led.view(fi)
            # too many points to follow one by one: fall through, the whole mirror becomes the live copy (as with `_wfs`)
            self._sync_point_writes()
            self._host_array()
            self._host_exported = True
            self._pt_views = None
        out = self._host_array()[key]
        if self._pt_copies is not None and self._dev_valid and self._dev is not None:
            fi = self._flat_index(key)
            if fi in self._pt_copies:
                # a point lives in ONE buffer: its pool array stays the live one, also once the mirror has been exported
                # (two unrelated live buffers for one point lost the first of two writes: ADVICE r5)
                return self._pt_copies.view(fi)
        if self._dev_valid and self._dev is not None and not self._host_exported:
            fi = self._flat_index(key)
            if self._pt_views is None:
                self._pt_views = _PointLedger(pt_shape, own=False)
            if fi not in self._pt_views:
                if len(self._pt_views) >= self._pt_cap():
                    self._sync_point_writes()
                    # every handed-out row is a view whose base is the mirror: none left (besides `out`) => start afresh
                    host = self._host
                    if self._host_valid and host.base is None and sys.getrefcount(host) <= 4:   # self._host, `host`, out.base, the argument
                        self._pt_views = _PointLedger(pt_shape, own=False)
                    else:
                        del host
                        self._host_array()
                        self._host_exported = True
                        self._pt_views = None
                        return out
                self._pt_views.add(fi, out)
        return out

    def __setitem__(self, key, value):
        """pythtb.py:2663-2672.  On a resident array only this point crosses PCIe."""
        self._check_key(key)
        val = np.array(value, dtype=complex)
        if self._dev_valid and self._dev is not None:
            pt = np.ascontiguousarray(np.broadcast_to(val, self._shape()[self._dim_arr:]))
            fi = self._flat_index(key)
            self._upload_points([fi], pt[None])
            tracked = self._pt_views is not None and fi in self._pt_views
            # the mirror row of a handed-out point is live even while the mirror as a whole is stale (skipping it
            # let the next synchronisation upload the OLD row over this assignment)
            if self._host is not None and (self._host_valid or tracked):
                self._host[key] = pt
            if tracked:
                self._pt_views.set(fi, pt)
            if self._pt_copies is not None and fi in self._pt_copies:   # arrays handed out for this point follow the storage
                self._pt_copies.set(fi, pt)
        else:
            self._host_array()[key] = val

    # ------------------------------------------------------------------ boundary conditions
    def impose_pbc(self, mesh_dir, k_dir):
        """Last slice along mesh_dir = first slice * exp(-2 pi i orb[:,k_dir])
        (pythtb.py:2674-2749); runs on the device copy."""
        if k_dir not in self._model._per:
            raise Exception("Periodic boundary condition can be specified only along periodic directions!")
        if mesh_dir not in range(min(self._dim_arr, 4)):
            raise Exception("\n\nWrong value of mesh_dir.")
        fac = np.exp(-2.j * np.pi * self._orb[:, k_dir])
        phase = np.ascontiguousarray(np.repeat(fac, self._nspin))
        h = self._ensure_dev()
        _lib.check(_lib.lib.tbk_wfs_impose(h, int(mesh_dir), _lib.dptr(phase.view(float))))
        self._device_wrote()

    def impose_loop(self, mesh_dir):
        """Last slice along mesh_dir = first slice (pythtb.py:2751-2791)."""
        if mesh_dir not in range(min(self._dim_arr, 4)):
            raise Exception("\n\nWrong value of mesh_dir.")
        h = self._ensure_dev()
        _lib.check(_lib.lib.tbk_wfs_impose(h, int(mesh_dir), None))
        self._device_wrote()

    # ------------------------------------------------------------------ position operator
    def _occ_list(self, occ):
        if isinstance(occ, str) and occ == "All":
            occ = np.arange(self._nsta_arr, dtype=int)
        else:
            occ = np.array(occ, dtype=int)
        if occ.ndim != 1:
            raise Exception("\n\nParameter occ must be a one-dimensional array or string \"All\".")
        return occ

    def _position_src(self, key, occ):
        """(evec, wfs) arguments for the model's position_* driver: the states stay on the device when the
        device copy is current (only the small results come back), else they are taken from the host mirror."""
        occ = self._occ_list(occ)
        if not _is_int(key):                   # the reference indexes _wfs[tuple(key)] (pythtb.py:2808)
            key = tuple(key)
            if self._dim_arr == 1 and len(key) == 1:
                key = key[0]
        self._check_key(key)
        self._sync_point_writes()
        if self._dev_valid and self._dev is not None and not (self._host_exported and self._host_valid):
            # (the reference indexes _wfs[key][occ] with NumPy: negative state indices count from the end, whichever copy
            # of the array happens to be current)
            if np.any(occ < -self._nsta_arr) or np.any(occ >= self._nsta_arr):
                raise IndexError("state index outside the range!")
            return None, (self._dev, [self._flat_index(key)], occ % self._nsta_arr, None)
        return self._host_array()[key if self._dim_arr == 1 else tuple(key)][occ], None

    def position_matrix(self, key, occ, dir):
        """tb_model.position_matrix for the states `occ` stored at mesh point `key` (pythtb.py:2793-2810)."""
        ev, src = self._position_src(key, occ)
        res = self._model.position_matrix(ev, dir, src)
        return res if src is None else res[0]

    def position_expectation(self, key, occ, dir):
        """pythtb.py:2812-2829."""
        ev, src = self._position_src(key, occ)
        res = self._model.position_expectation(ev, dir, src)
        return res if src is None else res[0]

    def position_hwf(self, key, occ, dir, hwf_evec=False, basis="wavefunction"):
        """pythtb.py:2831-2861 (note the default basis differs from tb_model.position_hwf)."""
        ev, src = self._position_src(key, occ)
        res = self._model.position_hwf(ev, dir, hwf_evec, basis, src)
        if src is None:
            return res
        return (res[0][0], res[1][0]) if hwf_evec else res[0]

    def position_hwf_mesh(self, occ, dir, hwf_evec=False, basis="wavefunction"):
        """Extension: position_hwf for every mesh point in one batched device call on the resident array.
        Returns hwfc[mesh..., nocc] (and hwf[mesh..., nocc, x]) -- the loop the reference's examples
        write around position_hwf (e.g. examples/cubic_slab_hwf.py)."""
        occ = self._occ_list(occ)
        if np.any(occ < -self._nsta_arr) or np.any(occ >= self._nsta_arr):
            raise IndexError("state index outside the range!")
        occ = occ % self._nsta_arr
        mesh = tuple(int(x) for x in self._mesh_arr)
        h = self._ensure_dev()
        res = self._model.position_hwf(None, dir, hwf_evec, basis, (h, None, occ, int(np.prod(mesh))))
        if not hwf_evec:
            return res.reshape(mesh + res.shape[1:])
        hwfc, hwf = res
        return hwfc.reshape(mesh + hwfc.shape[1:]), hwf.reshape(mesh + hwf.shape[1:])

    # ------------------------------------------------------------------ Berry quantities
    def _occ(self, occ):
        if (isinstance(occ, str) and occ == "All") or occ is None:
            return np.arange(self._nsta_arr, dtype=np.int32)
        return np.array(occ, dtype=int)

    def _wrap_occ(self, occ):
        """The band list as NumPy's fancy index reads it (the reference takes `_wfs[:, occ, :]` / `plane_wfs[:, :, occ]`,
        pythtb.py:2981, :2989-2996, :3141): entries in [-nsta_arr, nsta_arr) count from the end when negative, anything
        else is NumPy's IndexError."""
        n = self._nsta_arr
        bad = (occ < -n) | (occ >= n)
        if np.any(bad):
            raise IndexError("index %d is out of bounds for axis 1 with size %d" % (int(occ[bad][0]), n))
        return np.where(occ < 0, occ + n, occ)

    def berry_phase(self, occ="All", dir=None, contin=True, berry_evals=False):
        """Berry phase (or Wilson-loop eigenphases) of every string along `dir`
        (pythtb.py:2863-3066).  Link overlaps, polar factors, ordered products and
        eigenphases are computed on the device; the 2 pi continuity pass is host work."""
        occ = self._occ(occ)
        if occ.ndim != 1:
            raise Exception("\n\nParameter occ must be a one-dimensional array or string \"All\" or None.")
        if self._model._assume_position_operator_diagonal == False:  # noqa: E712
            raise Exception("\n\nBerry-like objects of Wannier90 models need "
                            "my_model.ignore_position_operator_offdiagonal()")
        if self._dim_arr == 1:
            use_dir = 0
        elif self._dim_arr in (2, 3):
            if dir is None or dir not in range(self._dim_arr):
                raise Exception("\n\nWrong direction for Berry phase calculation!")
            use_dir = int(dir)
        else:
            raise Exception("\n\nWrong dimensionality!")
        occ = self._wrap_occ(occ)
        h = self._ensure_dev()
        nocc = len(occ)
        other = [int(self._mesh_arr[d]) for d in range(self._dim_arr) if d != use_dir]
        nstr = int(np.prod(other)) if other else 1
        out = np.zeros(nstr * (nocc if berry_evals else 1), dtype=float)
        occ32 = np.ascontiguousarray(occ, dtype=np.int32)
        _lib.check(_lib.lib.tbk_berry_phase(h, _lib.iptr(occ32), nocc, use_dir, 1 if berry_evals else 0,
                                            _lib.dptr(out)))
        if self._dim_arr == 1:
            ret = out.copy() if berry_evals else np.float64(out[0])
        else:
            ret = out.reshape(other + ([nocc] if berry_evals else []))
        if contin:
            if not berry_evals:
                if self._dim_arr == 2:
                    ret = _one_phase_cont(ret, ret[0])
                elif self._dim_arr == 3:
                    for i in range(ret.shape[1]):
                        clos = ret[0, 0] if i == 0 else ret[0, i - 1]
                        ret[:, i] = _one_phase_cont(ret[:, i], clos)
            else:
                if self._dim_arr == 2:
                    ret = _array_phases_cont(ret, ret[0, :])
                elif self._dim_arr == 3:
                    for i in range(ret.shape[1]):
                        clos = ret[0, 0, :] if i == 0 else ret[0, i - 1, :]
                        ret[:, i] = _array_phases_cont(ret[:, i], clos)
        return ret

    def berry_flux(self, occ="All", dirs=None, individual_phases=False):
        """Berry flux through the (dirs[0],dirs[1]) planes (pythtb.py:3068-3205):
        plaquette phases and their deterministic sum are computed on the device."""
        if self._model._assume_position_operator_diagonal == False:  # noqa: E712
            raise Exception("\n\nBerry-like objects of Wannier90 models need "
                            "my_model.ignore_position_operator_offdiagonal()")
        if dirs is None:
            dirs = [0, 1]
        if dirs[0] == dirs[1]:
            raise Exception("Need to specify two different directions for Berry flux calculation.")
        if dirs[0] >= self._dim_arr or dirs[1] >= self._dim_arr or dirs[0] < 0 or dirs[1] < 0:
            raise Exception("Direction for Berry flux calculation out of bounds.")
        if self._dim_arr not in (2, 3, 4):
            raise Exception("\n\nWrong dimensionality!")
        # (the occupied-band list as a ctypes pointer is kept per distinct list: building it anew is 3 us of a 40 us call)
        b = self._call_bufs(self._nsta_arr)
        try:
            okey = occ if isinstance(occ, str) or occ is None else tuple(occ)
            o = b["occ"].get(okey)
        except TypeError:
            okey, o = None, None
        if o is None:
            occ_arr = self._occ(occ)
            if occ_arr.ndim == 1:
                occ_arr = self._wrap_occ(occ_arr)
            occ32 = np.ascontiguousarray(occ_arr, dtype=np.int32)
            o = (occ32, _lib.iptr(occ32), len(occ32))
            if okey is not None and occ_arr.ndim == 1:
                if len(b["occ"]) > 64:
                    b["occ"].clear()
                b["occ"][okey] = o
        h = self._ensure_dev()
        if self._dim_arr == 2:
            rest, nsl = [], 1
        else:
            rest = [int(self._mesh_arr[d]) for d in range(self._dim_arr) if d not in (dirs[0], dirs[1])]
            nsl = int(np.prod(rest)) if rest else 1
        n0 = int(self._mesh_arr[dirs[0]]) - 1
        n1 = int(self._mesh_arr[dirs[1]]) - 1
        t = b["totals"].get(nsl)
        if t is None:
            t = np.zeros(nsl, dtype=float)
            t = b["totals"][nsl] = (t, _lib.dptr(t))
        plaq = np.zeros((nsl, n0, n1), dtype=float) if individual_phases else None
        _lib.check(_lib.lib.tbk_berry_flux(h, o[1], o[2], int(dirs[0]), int(dirs[1]), t[1], _lib.dptr(plaq)))
        totals = t[0].copy()
        if self._dim_arr == 2:
            return plaq[0] if individual_phases else np.float64(totals[0])
        if individual_phases:
            return plaq.reshape(rest + [n0, n1])
        return totals.reshape(rest)
