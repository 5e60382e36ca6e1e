"""tb_model: host-side mirror of PythTB's model class for the k-space hot path.

Same constructor, setters, attribute names and call signatures as the reference
(`pythtb.py:29-560`, `:862-1103`, `:1792-2026`), so scripts written for PythTB
run unchanged; `_gen_ham`, `_sol_ham`, `solve_all` and `solve_one` execute on the
MI355X through libtbk (no CPU path).  The model->model transforms live in
transforms.py, the text report and the sketch plot in report.py / plotting.py.
"""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["tb_model"]


def _is_int(a):
    return np.issubdtype(type(a), np.integer)        # pythtb.py:3950


class tb_model(object):
    """Tight-binding model in reduced coordinates (reference: pythtb.py:29-184)."""

    def __init__(self, dim_k, dim_r, lat=None, orb=None, per=None, nspin=1):
        if not _is_int(dim_k):
            raise Exception("\n\nArgument dim_k not an integer")
        if dim_k < 0 or dim_k > 4:
            raise Exception("\n\nArgument dim_k out of range. Must be between 0 and 4.")
        if not _is_int(dim_r):
            raise Exception("\n\nArgument dim_r not an integer")
        if dim_r < dim_k or dim_r > 4:
            raise Exception("\n\nArgument dim_r out of range. Must be dim_r>=dim_k and dim_r<=4.")
        self._dim_k = dim_k
        self._dim_r = dim_r

        if lat is None or (isinstance(lat, str) and lat == "unit"):
            self._lat = np.identity(dim_r, float)
            print(" Lattice vectors not specified! I will use identity matrix.")
        else:
            self._lat = np.array(lat, dtype=float)
            if self._lat.shape != (dim_r, dim_r):
                raise Exception("\n\nWrong lat array dimensions")
        if dim_r > 0:
            vol = np.linalg.det(self._lat)
            if np.abs(vol) < 1.0E-6:
                raise Exception("\n\nLattice vectors length/area/volume too close to zero, or zero.")
            if vol < 0.0:
                raise Exception("\n\nLattice vectors need to form right handed system.")

        if orb is None or (isinstance(orb, str) and orb == "bravais"):
            self._norb = 1
            self._orb = np.zeros((1, dim_r))
            print(" Orbital positions not specified. I will assume a single orbital at the origin.")
        elif _is_int(orb):
            self._norb = orb
            self._orb = np.zeros((orb, dim_r))
            print(" Orbital positions not specified. I will assume ", orb, " orbitals at the origin")
        else:
            self._orb = np.array(orb, dtype=float)
            if self._orb.ndim != 2:
                raise Exception("\n\nWrong orb array rank")
            self._norb = self._orb.shape[0]
            if self._orb.shape[1] != dim_r:
                raise Exception("\n\nWrong orb array dimensions")

        if per is None:
            self._per = list(range(self._dim_k))
        else:
            if len(per) != self._dim_k:
                raise Exception("\n\nWrong choice of periodic/infinite direction!")
            self._per = per

        if nspin not in [1, 2]:
            raise Exception("\n\nWrong value of nspin, must be 1 or 2!")
        self._nspin = nspin
        self._assume_position_operator_diagonal = True
        self._nsta = self._norb * self._nspin

        if self._nspin == 1:
            self._site_energies = np.zeros(self._norb, dtype=float)
        else:
            self._site_energies = np.zeros((self._norb, 2, 2), dtype=complex)
        self._site_energies_specified = np.zeros(self._norb, dtype=bool)
        self._hoppings = []
        self._tbk_epoch = 0        # bumped whenever the tables change
        self._tbk_cache = None     # (epoch, device handle)

    # ------------------------------------------------------------------ tables
    def _val_to_block(self, val):
        """scalar / (I,sx,sy,sz) 4-vector / 2x2 -> 2x2 block for nspin=2 (pythtb.py:517-560)."""
        if self._nspin == 1:
            return val
        v = np.array(val)
        if v.shape == (2, 2):
            return v
        blk = np.zeros((2, 2), dtype=complex)
        if v.shape == ():
            blk[0, 0] = blk[1, 1] = v
        elif v.shape == (4,):
            blk[0, 0] = v[0] + v[3]
            blk[1, 1] = v[0] - v[3]
            blk[0, 1] = v[1] - 1.0j * v[2]
            blk[1, 0] = v[1] + 1.0j * v[2]
        else:
            raise Exception(
                "\n\nWrong format of the on-site or hopping term. Must be single number, or\n"
                "in the case of a spinfull model can be array of four numbers or 2x2\nmatrix.")
        return blk

    def set_onsite(self, onsite_en, ind_i=None, mode="set"):
        """On-site energies; modes set/reset/add (pythtb.py:186-306)."""
        if ind_i is None:
            if len(onsite_en) != self._norb:
                raise Exception("\n\nWrong number of site energies")
            values = list(onsite_en)
            targets = list(range(self._norb))
        else:
            if ind_i < 0 or ind_i >= self._norb:
                raise Exception("\n\nIndex ind_i out of scope.")
            values = [onsite_en]
            targets = [ind_i]
        for ons in values:
            a = np.array(ons)
            if a.shape == ():
                if np.abs(a - a.conjugate()) > 1.0E-8:
                    raise Exception("\n\nOnsite energy should not have imaginary part!")
            elif a.shape == (4,):
                if np.max(np.abs(a - a.conjugate())) > 1.0E-8:
                    raise Exception("\n\nOnsite energy or Zeeman field should not have imaginary part!")
            elif a.shape == (2, 2):
                if np.max(np.abs(a - a.T.conjugate())) > 1.0E-8:
                    raise Exception("\n\nOnsite matrix should be Hermitian!")
        how = mode.lower()
        if how == "set":
            if ind_i is not None:
                if self._site_energies_specified[ind_i]:
                    raise Exception("\n\nOnsite energy for this site was already specified! "
                                    "Use mode=\"reset\" or mode=\"add\".")
            elif np.any(self._site_energies_specified):
                raise Exception("\n\nSome or all onsite energies were already specified! "
                                "Use mode=\"reset\" or mode=\"add\".")
        elif how not in ("reset", "add"):
            raise Exception("\n\nWrong value of mode parameter")
        for t, ons in zip(targets, values):
            if how == "add":
                self._site_energies[t] += self._val_to_block(ons)
            else:
                self._site_energies[t] = self._val_to_block(ons)
            self._site_energies_specified[t] = True
        self._tbk_epoch += 1

    def set_hop(self, hop_amp, ind_i, ind_j, ind_R=None, mode="set", allow_conjugate_pair=False):
        """Hopping <phi_0i|H|phi_Rj>; stored as [amp, i, j, R] (pythtb.py:308-515)."""
        if self._dim_k != 0 and ind_R is None:
            raise Exception("\n\nNeed to specify ind_R!")
        if self._dim_k == 1 and _is_int(ind_R):
            full = np.zeros(self._dim_r, dtype=int)
            full[self._per] = ind_R
            ind_R = full
        if self._dim_k != 0 and len(ind_R) != self._dim_r:
            raise Exception("\n\nLength of input ind_R vector must equal dim_r! Even if dim_k<dim_r.")
        if ind_i < 0 or ind_i >= self._norb:
            raise Exception("\n\nIndex ind_i out of scope.")
        if ind_j < 0 or ind_j >= self._norb:
            raise Exception("\n\nIndex ind_j out of scope.")
        if ind_i == ind_j:
            if self._dim_k == 0 or all(int(ind_R[k]) == 0 for k in self._per):
                raise Exception("\n\nDo not use set_hop for onsite terms. Use set_onsite instead!")
        r_per = None if self._dim_k == 0 else np.array(ind_R)[self._per]
        if not allow_conjugate_pair:
            for h in self._hoppings:
                if ind_i == h[2] and ind_j == h[1]:
                    if self._dim_k == 0:
                        raise Exception(
                            "\n\nFollowing matrix element was already implicitely specified:\n"
                            "   i=" + str(ind_i) + " j=" + str(ind_j) + "\n"
                            "Remember, specifying <i|H|j> automatically specifies <j|H|i>.  For\n"
                            "consistency, specify all hoppings for a given bond in the same\n"
                            "direction.  (Or, alternatively, see the documentation on the\n"
                            "'allow_conjugate_pair' flag.)\n")
                    if np.all(r_per == -np.array(h[3])[self._per]):
                        raise Exception(
                            "\n\nFollowing matrix element was already implicitely specified:\n"
                            "   i=" + str(ind_i) + " j=" + str(ind_j) + " R=" + str(ind_R) + "\n"
                            "Remember,specifying <i|H|j+R> automatically specifies <j|H|i-R>.  For\n"
                            "consistency, specify all hoppings for a given bond in the same\n"
                            "direction.  (Or, alternatively, see the documentation on the\n"
                            "'allow_conjugate_pair' flag.)\n")
        block = self._val_to_block(hop_amp)
        entry = [block, int(ind_i), int(ind_j)]
        if self._dim_k != 0:
            entry.append(np.array(ind_R))
        match = None                                   # last entry with the same (i,j,R)
        for pos, h in enumerate(self._hoppings):
            if ind_i == h[1] and ind_j == h[2]:
                if self._dim_k == 0 or np.all(r_per == np.array(h[3])[self._per]):
                    match = pos
        how = mode.lower()
        if how == "set":
            if match is not None:
                raise Exception("\n\nHopping energy for this site was already specified! "
                                "Use mode=\"reset\" or mode=\"add\".")
            self._hoppings.append(entry)
        elif how == "reset":
            if match is None:
                self._hoppings.append(entry)
            else:
                self._hoppings[match] = entry
        elif how == "add":
            if match is None:
                self._hoppings.append(entry)
            else:
                self._hoppings[match][0] += entry[0]
        else:
            raise Exception("\n\nWrong value of mode parameter")
        self._tbk_epoch += 1

    def get_num_orbitals(self):
        return self._norb

    def get_orb(self):
        return self._orb.copy()

    def get_lat(self):
        return self._lat.copy()

    def invalidate_device_cache(self):
        """Call after editing `_hoppings` / `_site_energies` in place."""
        self._tbk_epoch += 1

    def __getstate__(self):                      # deepcopy / pickle: drop the device handle
        st = dict(self.__dict__)
        st["_tbk_cache"] = None
        return st

    def __del__(self):
        cache = getattr(self, "_tbk_cache", None)
        if cache is not None:
            try:
                _lib.lib.tbk_model_free(cache[2])
            except Exception:
                pass

    # ------------------------------------------------------------------ device
    def _flat_tables(self):
        """(orb_per, onsite, hop_i, hop_j, hop_R, hop_amp) in the layout of tbk_model_upload."""
        ns, no, dk = self._nspin, self._norb, self._dim_k
        nh = len(self._hoppings)
        orb_per = np.ascontiguousarray(self._orb[:, self._per], dtype=float).reshape(no, dk)
        onsite = np.zeros((no, ns, ns), dtype=complex)
        if ns == 1:
            onsite[:, 0, 0] = self._site_energies
        else:
            onsite[:] = self._site_energies
        # (whole-column conversions: a Python loop with four NumPy assignments per hopping was 22 us for the 9 hoppings of the
        # Haldane model -- more than the solve a parameter sweep re-uploads the model for)
        hops = self._hoppings
        if nh:
            hop_i = np.fromiter((hp[1] for hp in hops), dtype=np.int32, count=nh)
            hop_j = np.fromiter((hp[2] for hp in hops), dtype=np.int32, count=nh)
            hop_amp = np.empty((nh, ns, ns), dtype=complex)
            if ns == 1:
                hop_amp[:, 0, 0] = [hp[0] for hp in hops]
            else:
                for h, hp in enumerate(hops):
                    hop_amp[h] = hp[0]
            if dk > 0:
                hop_R = np.ascontiguousarray(np.array([hp[3] for hp in hops], dtype=np.int32).reshape(nh, -1)[:, self._per])
            else:
                hop_R = np.zeros((nh, 0), dtype=np.int32)
        else:
            hop_i = np.zeros(0, dtype=np.int32)
            hop_j = np.zeros(0, dtype=np.int32)
            hop_R = np.zeros((0, dk), dtype=np.int32)
            hop_amp = np.zeros((0, ns, ns), dtype=complex)
        return orb_per, onsite, hop_i, hop_j, hop_R, hop_amp

    def _device_model(self):
        ctx = _lib.default_context()
        c = self._tbk_cache
        # the epoch counts edits made through the setters/transforms; the fingerprint also catches scripts
        # that write `_orb` / `_site_energies` / `_hoppings` (length) directly, as some of the reference's do
        mark = (self._tbk_epoch, len(self._hoppings), self._orb.tobytes(), np.asarray(self._site_energies).tobytes())
        if c is not None and c[0] == mark and c[1] is ctx:
            return c[2]
        if c is not None:
            _lib.lib.tbk_model_free(c[2])
            self._tbk_cache = None
        orb_per, onsite, hop_i, hop_j, hop_R, hop_amp = self._flat_tables()
        h = C.c_void_p()
        _lib.check(_lib.lib.tbk_model_upload(
            ctx.handle, self._dim_k, self._norb, self._nspin, _lib.dptr(orb_per),
            _lib.dptr(onsite.view(float)), len(hop_i), _lib.iptr(hop_i), _lib.iptr(hop_j),
            _lib.iptr(hop_R.reshape(-1)) if hop_R.size else None,
            _lib.dptr(hop_amp.view(float)) if hop_amp.size else None, C.byref(h)))
        self._tbk_cache = (mark, ctx, h)
        return h

    # ------------------------------------------------------------------ solve
    def _k_array(self, k_list):
        k = np.asarray(k_list, dtype=float)          # no copy of an array that is already float64
        if k.size == 0:                               # empty list: the reference's loop runs zero times
            return np.zeros((0, self._dim_k), dtype=float)
        if self._dim_k == 1 and k.ndim == 1:
            k = k.reshape(-1, 1)
        if k.ndim != 2 or k.shape[1] != self._dim_k:
            raise Exception("\n\nk-vector of wrong shape!")
        return np.ascontiguousarray(k)

    def _gen_ham(self, k_input=None):
        """H(k) for one k in reduced coordinates (pythtb.py:874-925), built on the device."""
        if k_input is None:
            if self._dim_k != 0:
                raise Exception("\n\nHave to provide a k-vector!")
            k = None
        else:
            kp = np.array(k_input, dtype=float)
            if kp.ndim == 0:
                kp = kp.reshape(1)
            if kp.shape != (self._dim_k,):
                raise Exception("\n\nk-vector of wrong shape!")
            k = np.ascontiguousarray(kp.reshape(1, -1))
        n = self._nsta
        ham = np.zeros((1, n, n), dtype=complex)
        _lib.check(_lib.lib.tbk_gen_ham(self._device_model(), _lib.dptr(k) if self._dim_k else None, 1,
                                        _lib.dptr(ham.view(float))))
        if self._nspin == 1:
            return ham[0]
        return ham[0].reshape(self._norb, 2, self._norb, 2)

    def _sol_ham(self, ham, eig_vectors=False):
        """Eigen-decomposition of one Hamiltonian (pythtb.py:927-953), on the device."""
        n = self._nsta
        hm = np.ascontiguousarray(np.array(ham, dtype=complex).reshape(1, n, n))
        if np.max(hm[0] - hm[0].T.conj()) > 1.0E-9:
            raise Exception("\n\nHamiltonian matrix is not hermitian?!")
        ev = np.zeros((n, 1), dtype=float)
        vec = np.zeros((n, 1, n), dtype=complex) if eig_vectors else None
        _lib.check(_lib.lib.tbk_eigh_batch(_lib.default_context().handle, n, _lib.dptr(hm.view(float)), 1,
                                           _lib.dptr(ev), _lib.dptr(vec.view(float)) if eig_vectors else None))
        if not eig_vectors:
            return ev[:, 0].copy()
        out = vec[:, 0, :]
        if self._nspin == 2:
            out = out.reshape(n, self._norb, 2)
        return ev[:, 0].copy(), out.copy()

    def solve_all(self, k_list=None, eig_vectors=False):
        """Eigenvalues eval[band,k] (and evec[band,k,orb(,spin)]) on a list of k
        (pythtb.py:955-1079): one fused H(k)+eigh launch over the whole list."""
        n = self._nsta
        if k_list is None:
            if self._dim_k != 0:
                raise Exception("\n\nHave to provide a k-vector!")
            nk, k = 1, None
        else:
            k = self._k_array(k_list) if self._dim_k > 0 else None
            nk = len(k_list)
        ev = np.empty((n, nk), dtype=float)                   # filled completely by the device-to-host copies
        vec = np.empty((n, nk, n), dtype=complex) if eig_vectors else None
        if nk > 0:
            _lib.check(_lib.lib.tbk_solve_list(self._device_model(), _lib.dptr(k), nk, _lib.dptr(ev),
                                               _lib.dptr(vec.view(float)) if eig_vectors else None))
        if eig_vectors and self._nspin == 2:
            vec = vec.reshape(n, nk, self._norb, 2)
        if k_list is None:
            return (ev[:, 0], vec[:, 0]) if eig_vectors else ev[:, 0]
        return (ev, vec) if eig_vectors else ev

    def solve_one(self, k_point=None, eig_vectors=False):
        """solve_all for a single k (pythtb.py:1081-1103)."""
        if k_point is None:
            return self.solve_all(eig_vectors=eig_vectors)
        if eig_vectors:
            ev, vec = self.solve_all([k_point], eig_vectors=True)
            return ev[:, 0], vec[:, 0]
        return self.solve_all([k_point])[:, 0]

    # ------------------------------------------------------------------ mesh shortcuts (extensions)
    def _mesh_arg(self, mesh_size):
        mesh = np.array(list(map(round, mesh_size)), dtype=np.int32)     # same checks as k_uniform_mesh
        if mesh.shape != (self._dim_k,):
            print(mesh.shape)
            raise Exception("\n\nIncorrect size of the specified k-mesh!")
        if np.min(mesh) <= 0:
            raise Exception("\n\nMesh must have positive non-zero number of elements.")
        if self._dim_k not in (1, 2, 3):
            raise Exception("\n\nUnsupported dim_k!")
        return np.ascontiguousarray(mesh), int(np.prod(mesh, dtype=np.int64))

    def solve_all_mesh(self, mesh_size, eig_vectors=False):
        """Extension: `solve_all(k_uniform_mesh(mesh_size), eig_vectors)` with the k list generated
        on the device -- the same arrays, minus the 8*dim_k bytes per k-point of upload."""
        mesh, nk = self._mesh_arg(mesh_size)
        n = self._nsta
        ev = np.zeros((n, nk), dtype=float)
        vec = np.zeros((n, nk, n), dtype=complex) if eig_vectors else None
        _lib.check(_lib.lib.tbk_solve_mesh(self._device_model(), _lib.iptr(mesh), _lib.dptr(ev),
                                           _lib.dptr(vec.view(float)) if eig_vectors else None))
        if eig_vectors and self._nspin == 2:
            vec = vec.reshape(n, nk, self._norb, 2)
        return (ev, vec) if eig_vectors else ev

    def dos_mesh(self, mesh_size, bins=50, range=None, per_band=False):
        """Extension: `np.histogram(solve_all(k_uniform_mesh(mesh_size)).flatten(), bins, range)` -- the
        density-of-states reduction of the reference's examples/haldane.py:96-121 -- with the
        eigenvalues kept on the device: (counts, bin_edges), counts int64 `(bins,)`, or
        `(nsta, bins)` with per_band=True.  `bins` is a number of equal-width bins."""
        mesh, nk = self._mesh_arg(mesh_size)
        if not _is_int(bins) or bins < 1:
            raise Exception("\n\ndos_mesh: bins must be a positive integer (equal-width bins)")
        n = self._nsta
        h = self._device_model()
        if range is None:
            lo = np.zeros(n)
            hi = np.zeros(n)
            _lib.check(_lib.lib.tbk_dos_mesh(h, _lib.iptr(mesh), 0, None, None, _lib.dptr(lo), _lib.dptr(hi)))
            range = (lo.min(), hi.max())
        # the edges np.histogram itself would use for this range (incl. its widening of an empty range)
        edges = np.ascontiguousarray(np.histogram_bin_edges(np.zeros(0), bins=int(bins), range=range), dtype=float)
        counts = np.zeros((n, int(bins)), dtype=np.int64)
        _lib.check(_lib.lib.tbk_dos_mesh(h, _lib.iptr(mesh), int(bins), _lib.dptr(edges),
                                         counts.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int64)), None, None))
        return (counts if per_band else counts.sum(axis=0)), edges

    # ------------------------------------------------------------------ position operator
    def ignore_position_operator_offdiagonal(self):
        self._assume_position_operator_diagonal = True

    def _position_dir_check(self, dir):
        if dir in self._per:
            raise Exception("Can not compute position matrix elements along periodic direction!")
        if dir < 0 or dir >= self._dim_r:
            raise Exception("Direction out of range!")
        if self._assume_position_operator_diagonal == False:  # noqa: E712
            raise Exception("\n\nPosition-operator objects of Wannier90 models need "
                            "my_model.ignore_position_operator_offdiagonal()")

    def _position_call(self, evec, dir, want_x, want_c, want_w, orbital, wfs=None):
        """Shared driver of position_matrix/expectation/hwf.  `evec` is one point
        [band,orb(,spin)] or a batch [point,band,orb(,spin)] (extension).  With
        wfs = (device handle, point indices or None, occ) the states are read from a
        resident wf_array instead and `evec` is ignored (always batched)."""
        self._position_dir_check(dir)
        ncomp = self._norb * self._nspin
        pos = np.ascontiguousarray(np.repeat(self._orb[:, dir], self._nspin), dtype=float)   # pythtb.py:2078-2083
        if wfs is None:
            ev = np.array(evec, dtype=complex)
            tail = 2 if self._nspin == 2 else 1
            batched = ev.ndim == tail + 2
            if ev.ndim not in (tail + 1, tail + 2) or ev.shape[-tail:] != ((self._norb, 2) if tail == 2 else (self._norb,)):
                raise Exception("\n\nWrong shape of the eigenvector array")
            nsub = ev.shape[-tail - 1]
            nk = ev.shape[0] if batched else 1
            flat = np.ascontiguousarray(ev.reshape(nk, nsub, ncomp))
        else:
            handle, pts, occ, npts_all = wfs
            batched = True
            nsub = len(occ)
            nk = npts_all if pts is None else len(pts)
        xmat = np.zeros((nk, nsub, nsub), dtype=complex) if want_x else None
        hwfc = np.zeros((nk, nsub), dtype=float) if want_c else None
        width = ncomp if orbital else nsub
        hwf = np.zeros((nk, nsub, width), dtype=complex) if want_w else None
        outs = (_lib.dptr(xmat.view(float)) if want_x else None, _lib.dptr(hwfc),
                _lib.dptr(hwf.view(float)) if want_w else None, 1 if orbital else 0)
        if wfs is None:
            _lib.check(_lib.lib.tbk_position_hwf(
                _lib.default_context().handle, _lib.dptr(flat.view(float)), nk, nsub, ncomp, _lib.dptr(pos), *outs))
        else:
            import ctypes as C
            occ32 = np.ascontiguousarray(occ, dtype=np.int32)
            p64 = None if pts is None else np.ascontiguousarray(pts, dtype=np.int64)
            _lib.check(_lib.lib.tbk_wfs_position_hwf(
                handle, None if p64 is None else p64.ctypes.data_as(C.POINTER(C.c_int64)), nk, _lib.iptr(occ32), nsub,
                _lib.dptr(pos), *outs))
        if want_w and orbital and self._nspin == 2:
            hwf = hwf.reshape(nk, nsub, self._norb, 2)
        pick = (lambda a: a) if batched else (lambda a: None if a is None else a[0])
        return pick(xmat), pick(hwfc), pick(hwf)

    def position_matrix(self, evec, dir, _wfs=None):
        """X_mn = <u_m| r_dir |u_n> for the states `evec` of one k-point (pythtb.py:2034-2098)."""
        xmat, _, _ = self._position_call(evec, dir, True, False, False, False, _wfs)
        herm = xmat - np.swapaxes(xmat.conj(), -1, -2)
        if np.max(herm) > 1.0E-9:
            raise Exception("\n\n Position matrix is not hermitian?!")
        return xmat

    def position_expectation(self, evec, dir, _wfs=None):
        """Diagonal of the position matrix (pythtb.py:2100-2141)."""
        xmat = self.position_matrix(evec, dir, _wfs)
        return np.array(np.real(np.diagonal(xmat, axis1=-2, axis2=-1)), dtype=float)

    def position_hwf(self, evec, dir, hwf_evec=False, basis="orbital", _wfs=None):
        """Hybrid Wannier centres (and functions) = eigen-decomposition of the position
        matrix (pythtb.py:2143-2279)."""
        if not hwf_evec:
            _, hwfc, _ = self._position_call(evec, dir, False, True, False, False, _wfs)
            return hwfc
        which = basis.lower().strip()
        if which in ("wavefunction", "bloch"):
            orbital = False
        elif which == "orbital":
            orbital = True
        else:
            raise Exception("\n\nBasis must be either 'wavefunction', 'bloch', or 'orbital'")
        _, hwfc, hwf = self._position_call(evec, dir, False, True, True, orbital, _wfs)
        return (hwfc, hwf)

    # ------------------------------------------------------------------ k generators (host)
    def k_uniform_mesh(self, mesh_size):
        """Gamma-containing uniform mesh, last index fastest (pythtb.py:1792-1861)."""
        use = np.array(list(map(round, mesh_size)), dtype=int)
        if use.shape != (self._dim_k,):
            print(use.shape)
            raise Exception("\n\nIncorrect size of the specified k-mesh!")
        if np.min(use) <= 0:
            raise Exception("\n\nMesh must have positive non-zero number of elements.")
        if self._dim_k not in (1, 2, 3):
            raise Exception("\n\nUnsupported dim_k!")
        idx = np.indices(tuple(use)).reshape(self._dim_k, -1).T
        # a plain ndarray like the reference's; C-contiguous, so solve_all hands it to the device as is.  (solve_all always
        # solves the k list it is GIVEN; the device-generated mesh is the explicit extension solve_all_mesh.)
        return np.divide(idx, use.astype(float), order='C')

    def k_path(self, kpts, nk, report=True):
        """Piecewise-linear path through `kpts` with `nk` points, spaced by the
        Cartesian metric (pythtb.py:1863-2026).  Returns (k_vec, k_dist, k_node)."""
        if isinstance(kpts, str):
            named = {"full": [[0.0], [0.5], [1.0]], "fullc": [[-0.5], [0.0], [0.5]], "half": [[0.0], [0.5]]}
            nodes = np.array(named[kpts]) if kpts in named else np.array(kpts)
        else:
            nodes = np.array(kpts)
        if nodes.ndim == 1 and self._dim_k == 1:
            nodes = nodes.reshape(-1, 1)
        if nodes.shape[1] != self._dim_k:
            print('input k-space dimension is', nodes.shape[1])
            print('k-space dimension taken from model is', self._dim_k)
            raise Exception("\n\nk-space dimensions do not match")
        if nk < nodes.shape[0]:
            raise Exception("\n\nMust have more points in the path than number of nodes.")
        n_nodes = nodes.shape[0]
        lat_per = np.copy(self._lat)[self._per]
        k_metric = np.linalg.inv(np.dot(lat_per, lat_per.T))
        k_node = np.zeros(n_nodes, dtype=float)
        for s in range(1, n_nodes):
            dk = nodes[s] - nodes[s - 1]
            k_node[s] = k_node[s - 1] + np.sqrt(np.dot(dk, np.dot(k_metric, dk)))
        node_index = [0]
        for s in range(1, n_nodes - 1):
            node_index.append(int(round(k_node[s] / k_node[-1] * (nk - 1))))
        node_index.append(nk - 1)
        k_dist = np.zeros(nk, dtype=float)
        k_vec = np.zeros((nk, self._dim_k), dtype=float)
        k_vec[0] = nodes[0]
        for s in range(1, n_nodes):
            lo, hi = node_index[s - 1], node_index[s]
            if hi == lo:        # two nodes on one path index: the reference's 0/0 (pythtb.py:1991)
                raise ZeroDivisionError("float division by zero")
            frac = (np.arange(lo, hi + 1) - lo).astype(float) / float(hi - lo)
            k_dist[lo:hi + 1] = k_node[s - 1] + frac * (k_node[s] - k_node[s - 1])
            k_vec[lo:hi + 1] = nodes[s - 1] + frac[:, None] * (nodes[s] - nodes[s - 1])
        if report:
            if self._dim_k == 1:
                print(' Path in 1D BZ defined by nodes at ' + str(nodes.flatten()))
            else:
                print('----- k_path report begin ----------')
                keep = np.get_printoptions()
                np.set_printoptions(precision=5)
                print('real-space lattice vectors\n', lat_per)
                print('k-space metric tensor\n', k_metric)
                print('internal coordinates of nodes\n', nodes)
                if lat_per.shape[0] == lat_per.shape[1]:
                    rec = np.linalg.inv(lat_per).T
                    print('reciprocal-space lattice vectors\n', rec)
                    print('cartesian coordinates of nodes\n', np.tensordot(nodes, rec, axes=1))
                print('list of segments:')
                for s in range(1, n_nodes):
                    seg = str(round(k_node[s] - k_node[s - 1], 5)).rjust(7)
                    print('  length = ' + seg + '  from ', nodes[s - 1], ' to ', nodes[s])
                print('node distance list:', k_node)
                print('node index list:   ', np.array(node_index))
                np.set_printoptions(precision=keep["precision"])
                print('----- k_path report end ------------')
            print()
        return (k_vec, k_dist, k_node)


from . import plotting as _plotting  # noqa: E402
from . import report as _report  # noqa: E402
from . import transforms as _transforms  # noqa: E402

_transforms.install(tb_model)
tb_model.display = _report.display
tb_model.visualize = _plotting.visualize
