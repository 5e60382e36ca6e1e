"""CPU oracle for the PythTB k-mesh solve + Berry path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the algorithm of the reference
(`/root/reference/pythtb.py`, PythTB 1.8.0) for the one hot path this repository
accelerates.  It is *not* part of the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it,
and only as the checker / the timed CPU baseline.  `pythtb_amd` never imports it.

Parity pin: every function here is checked (tests/test_oracle_golden.py)
  * against the golden `.npy` files the reference's own tests hold
    (copied as data into tests/golden/reference_tests/), and
  * against vectors generated in the development container by importing the
    reference itself (tests/golden/make_golden.py -> tests/golden/*.npz), and
  * when `/root/reference` is present, live against the imported reference
    (tests/test_oracle_golden.py::test_oracle_live_against_imported_reference, skipped elsewhere).

Each function cites the reference lines it follows.  The per-k / per-link loop
structure is kept on purpose (it is what the CPU baseline times); `*_vec`
variants are vectorised checkers used for larger meshes.

A "model" here is any object with the reference's attribute names
(`_dim_k,_dim_r,_norb,_nspin,_nsta,_per,_orb,_lat,_site_energies,_hoppings`);
`Model` below is the minimal container rebuilt from fixture tables.
"""
import numpy as np

TWO_PI = 2.0 * np.pi


class Model(object):
    """Bare container with the reference's attribute names (pythtb.py:94-184)."""

    def __init__(self, dim_k, dim_r, lat, orb, per, nspin, site_energies, hoppings):
        self._dim_k = int(dim_k)
        self._dim_r = int(dim_r)
        self._lat = np.array(lat, dtype=float).reshape(self._dim_r, self._dim_r)
        self._orb = np.array(orb, dtype=float).reshape(-1, self._dim_r)
        self._norb = self._orb.shape[0]
        self._per = [int(p) for p in per]
        self._nspin = int(nspin)
        self._nsta = self._norb * self._nspin
        self._site_energies = np.array(site_energies)
        self._hoppings = hoppings

    @staticmethod
    def from_tables(t):
        """Rebuild from the dict written by `model_tables` (fixture format)."""
        nspin = int(t["nspin"])
        dim_k = int(t["dim_k"])
        hops = []
        for h in range(len(t["hop_i"])):
            amp = t["hop_amp"][h]
            amp = complex(amp[0, 0]) if nspin == 1 else np.array(amp, dtype=complex)
            ent = [amp, int(t["hop_i"][h]), int(t["hop_j"][h])]
            if dim_k > 0:
                ent.append(np.array(t["hop_R"][h], dtype=int))
            hops.append(ent)
        if nspin == 1:
            site = np.array(t["site_energies"], dtype=float)
        else:
            site = np.array(t["site_energies"], dtype=complex)
        return Model(dim_k, int(t["dim_r"]), t["lat"], t["orb"], list(t["per"]), nspin, site, hops)


def model_tables(m):
    """Dump a model's tables to plain arrays (the fixture format)."""
    nh = len(m._hoppings)
    ns = m._nspin
    hop_amp = np.zeros((nh, ns, ns), dtype=complex)
    hop_i = np.zeros(nh, dtype=np.int32)
    hop_j = np.zeros(nh, dtype=np.int32)
    hop_R = np.zeros((nh, m._dim_r), dtype=np.int32)
    for h, hop in enumerate(m._hoppings):
        hop_amp[h] = np.array(hop[0], dtype=complex).reshape(ns, ns) if ns == 2 else complex(hop[0])
        hop_i[h] = hop[1]
        hop_j[h] = hop[2]
        if m._dim_k > 0:
            hop_R[h] = np.array(hop[3], dtype=int)
    return dict(dim_k=m._dim_k, dim_r=m._dim_r, nspin=ns, lat=np.array(m._lat, dtype=float),
                orb=np.array(m._orb, dtype=float), per=np.array(m._per, dtype=np.int32),
                site_energies=np.array(m._site_energies), hop_amp=hop_amp,
                hop_i=hop_i, hop_j=hop_j, hop_R=hop_R)


# --------------------------------------------------------------------------
# H(k) and its eigen-decomposition
# --------------------------------------------------------------------------

def gen_ham(m, k_input=None):
    """H(k) for ONE k in reduced coordinates.  Follows pythtb.py:874-925.

    Returns (norb,norb) or (norb,2,norb,2) complex, like the reference."""
    if k_input is None:
        if m._dim_k != 0:
            raise Exception("\n\nHave to provide a k-vector!")
        kpnt = None
    else:
        kpnt = np.array(k_input)
        if kpnt.ndim == 0:
            kpnt = kpnt.reshape(1)
        if kpnt.shape != (m._dim_k,):
            raise Exception("\n\nk-vector of wrong shape!")
    no, ns = m._norb, m._nspin
    if ns == 1:
        ham = np.zeros((no, no), dtype=complex)
        ham[np.arange(no), np.arange(no)] = m._site_energies
    else:
        ham = np.zeros((no, 2, no, 2), dtype=complex)
        for a in range(no):
            ham[a, :, a, :] = m._site_energies[a]
    for hop in m._hoppings:
        amp = complex(hop[0]) if ns == 1 else np.array(hop[0], dtype=complex)
        a, b = hop[1], hop[2]
        if m._dim_k > 0:
            # :910-916  rv = (R + tau_b - tau_a)[per];  phase = exp(2 pi i k.rv)
            rv = (-m._orb[a] + m._orb[b] + np.array(hop[3], dtype=float))[m._per]
            amp = amp * np.exp((2.0j) * np.pi * np.dot(kpnt, rv))
        if ns == 1:
            ham[a, b] += amp
            ham[b, a] += np.conj(amp)
        else:
            ham[a, :, b, :] += amp
            ham[b, :, a, :] += amp.conj().T
    return ham


def sol_ham(m, ham, eig_vectors=False):
    """Hermitian eigen-decomposition of one H.  Follows pythtb.py:927-953 and
    _nicefy_eig :3765-3775: ascending real eigenvalues; eigenvectors as ROWS;
    spinor rows reshaped to (nsta,norb,2)."""
    hm = ham if m._nspin == 1 else ham.reshape(2 * m._norb, 2 * m._norb)
    if np.max(hm - hm.T.conj()) > 1.0e-9:
        raise Exception("\n\nHamiltonian matrix is not hermitian?!")
    if not eig_vectors:
        w = np.linalg.eigvalsh(hm)
        return np.sort(np.array(w.real, dtype=float))
    w, v = np.linalg.eigh(hm)
    order = np.argsort(np.array(w.real, dtype=float))
    w = np.array(w.real, dtype=float)[order]
    rows = v.T[order]
    if m._nspin == 2:
        rows = rows.reshape(m._nsta, m._norb, 2)
    return w, rows


def solve_all(m, k_list=None, eig_vectors=False):
    """Per-k Python loop, band-major outputs.  Follows pythtb.py:955-1079."""
    if k_list is None:
        return sol_ham(m, gen_ham(m), eig_vectors)
    nk = len(k_list)
    ev = np.zeros((m._nsta, nk), dtype=float)
    shape = (m._nsta, nk, m._norb) if m._nspin == 1 else (m._nsta, nk, m._norb, 2)
    vec = np.zeros(shape, dtype=complex)
    for ik, k in enumerate(k_list):
        h = gen_ham(m, k)
        if eig_vectors:
            w, v = sol_ham(m, h, True)
            ev[:, ik] = w
            vec[:, ik] = v
        else:
            ev[:, ik] = sol_ham(m, h, False)
    return (ev, vec) if eig_vectors else ev


def solve_one(m, k_point=None, eig_vectors=False):
    """pythtb.py:1081-1103."""
    if k_point is None:
        return solve_all(m, None, eig_vectors)
    if eig_vectors:
        w, v = solve_all(m, [k_point], True)
        return w[:, 0], v[:, 0]
    return solve_all(m, [k_point], False)[:, 0]


def ham_batch(m, kpts):
    """Vectorised H(k) for many k: returns (nk, nsta, nsta) with spin index
    interleaved as 2*orb+spin (the reshape of pythtb.py:933).  Checker only."""
    kpts = np.asarray(kpts, dtype=float).reshape(-1, max(m._dim_k, 1))
    nk = kpts.shape[0]
    no, ns = m._norb, m._nspin
    n = no * ns
    ham = np.zeros((nk, n, n), dtype=complex)
    for a in range(no):
        blk = np.array(m._site_energies[a], dtype=complex).reshape(ns, ns) if ns == 2 \
            else np.array([[m._site_energies[a]]], dtype=complex)
        ham[:, a * ns:(a + 1) * ns, a * ns:(a + 1) * ns] += blk
    for hop in m._hoppings:
        amp = np.array(hop[0], dtype=complex).reshape(ns, ns)
        a, b = hop[1], hop[2]
        if m._dim_k > 0:
            rv = (-m._orb[a] + m._orb[b] + np.array(hop[3], dtype=float))[m._per]
            ph = np.exp((2.0j) * np.pi * (kpts @ rv))
        else:
            ph = np.ones(nk, dtype=complex)
        ham[:, a * ns:(a + 1) * ns, b * ns:(b + 1) * ns] += ph[:, None, None] * amp
        ham[:, b * ns:(b + 1) * ns, a * ns:(a + 1) * ns] += np.conj(ph)[:, None, None] * amp.conj().T
    return ham


def solve_all_vec(m, kpts, eig_vectors=False):
    """Vectorised solve_all (batched LAPACK).  Same outputs/layout as solve_all."""
    ham = ham_batch(m, kpts)
    nk = ham.shape[0]
    if not eig_vectors:
        return np.ascontiguousarray(np.linalg.eigvalsh(ham).T)
    w, v = np.linalg.eigh(ham)                     # v[k,:,b] is eigenvector b
    vec = np.ascontiguousarray(np.transpose(v, (2, 0, 1)))   # [band,k,comp]
    if m._nspin == 2:
        vec = vec.reshape(m._nsta, nk, m._norb, 2)
    return np.ascontiguousarray(w.T), vec


# --------------------------------------------------------------------------
# k generators
# --------------------------------------------------------------------------

def k_uniform_mesh(m, mesh_size):
    """Gamma-containing uniform mesh, last index fastest.  pythtb.py:1792-1861."""
    use = np.array([int(round(x)) for x in mesh_size], dtype=int)
    if use.shape != (m._dim_k,):
        raise Exception("\n\nIncorrect size of the specified k-mesh!")
    if np.min(use) <= 0:
        raise Exception("\n\nMesh must have positive non-zero number of elements.")
    if m._dim_k not in (1, 2, 3):
        raise Exception("\n\nUnsupported dim_k!")
    axes = [np.arange(nn) / float(nn) for nn in use]
    grids = np.meshgrid(*axes, indexing="ij")
    return np.stack([g.reshape(-1) for g in grids], axis=1)


def k_path(m, kpts, nk):
    """Piecewise-linear k path.  pythtb.py:1863-2026 (report omitted)."""
    if isinstance(kpts, str):
        if kpts == "full":
            nodes = np.array([[0.0], [0.5], [1.0]])
        elif kpts == "fullc":
            nodes = np.array([[-0.5], [0.0], [0.5]])
        elif kpts == "half":
            nodes = np.array([[0.0], [0.5]])
        else:
            raise Exception("unknown path keyword")
    else:
        nodes = np.array(kpts)
    if nodes.ndim == 1 and m._dim_k == 1:
        nodes = nodes.reshape(-1, 1)
    if nodes.shape[1] != m._dim_k:
        raise Exception("\n\nk-space dimensions do not match")
    if nk < nodes.shape[0]:
        raise Exception("\n\nMust have more points in the path than number of nodes.")
    nn = nodes.shape[0]
    lat_per = np.array(m._lat)[m._per]
    metric = np.linalg.inv(lat_per @ lat_per.T)          # :1960
    k_node = np.zeros(nn, dtype=float)
    for s in range(1, nn):
        dk = nodes[s] - nodes[s - 1]
        k_node[s] = k_node[s - 1] + np.sqrt(dk @ (metric @ dk))
    idx = [0]
    for s in range(1, nn - 1):
        idx.append(int(round(k_node[s] / k_node[-1] * (nk - 1))))
    idx.append(nk - 1)
    k_dist = np.zeros(nk, dtype=float)
    k_vec = np.zeros((nk, m._dim_k), dtype=float)
    k_vec[0] = nodes[0]
    for s in range(1, nn):
        lo, hi = idx[s - 1], idx[s]
        for j in range(lo, hi + 1):
            f = float(j - lo) / float(hi - lo)
            k_dist[j] = k_node[s - 1] + f * (k_node[s] - k_node[s - 1])
            k_vec[j] = nodes[s - 1] + f * (nodes[s] - nodes[s - 1])
    return k_vec, k_dist, k_node


# --------------------------------------------------------------------------
# wf_array: mesh solve and boundary conditions
# --------------------------------------------------------------------------

def wfs_alloc(m, mesh_arr, nsta_arr=None):
    """_wfs[k1..kD, state, orb(,spin)] zeros.  pythtb.py:2409-2419."""
    mesh = [int(x) for x in mesh_arr]
    if min(mesh) <= 1:
        raise Exception("\n\nDimension of wf_array object in each direction must be 2 or larger.")
    nsta = m._nsta if nsta_arr is None else int(nsta_arr)
    shape = mesh + [nsta, m._norb] + ([2] if m._nspin == 2 else [])
    return np.zeros(shape, dtype=complex)


def impose_pbc(m, wfs, mesh_dir, k_dir):
    """last slice = first slice * exp(-2 pi i orb[:,k_dir]).  pythtb.py:2725-2747."""
    if k_dir not in m._per:
        raise Exception("Periodic boundary condition can be specified only along periodic directions!")
    fac = np.exp(-1j * TWO_PI * m._orb[:, k_dir])
    if m._nspin == 2:
        fac = np.stack([fac, fac], axis=1)
    first = [slice(None)] * mesh_dir + [0, Ellipsis]
    last = [slice(None)] * mesh_dir + [-1, Ellipsis]
    wfs[tuple(last)] = wfs[tuple(first)] * fac


def impose_loop(wfs, mesh_dir):
    """last slice = first slice.  pythtb.py:2782-2789."""
    first = [slice(None)] * mesh_dir + [0, Ellipsis]
    last = [slice(None)] * mesh_dir + [-1, Ellipsis]
    wfs[tuple(last)] = wfs[tuple(first)]


def solve_on_grid(m, mesh_arr, start_k, vectorised=False):
    """Fill a wf mesh: each index i_d < N_d-1 solved at start_k[d]+i_d/(N_d-1),
    then impose_pbc per direction in order.  pythtb.py:2421-2532.
    Returns (wfs, min_gaps) where min_gaps has shape (nsta-1,) or is None."""
    mesh = [int(x) for x in mesh_arr]
    D = len(mesh)
    if D != m._dim_k:
        raise Exception("dimension of wf_array must equal dim_k")
    wfs = wfs_alloc(m, mesh)
    inner = [nn - 1 for nn in mesh]
    gaps = None if m._nsta <= 1 else np.zeros(inner + [m._nsta - 1], dtype=float)
    if vectorised:
        axes = [start_k[d] + np.arange(mesh[d] - 1, dtype=float) / float(mesh[d] - 1) for d in range(D)]
        grids = np.meshgrid(*axes, indexing="ij")
        kk = np.stack([g.reshape(-1) for g in grids], axis=1)
        w, v = solve_all_vec(m, kk, True)
        tail = list(v.shape[2:])
        vv = np.moveaxis(v, 0, 1).reshape(inner + [m._nsta] + tail)
        wfs[tuple(slice(0, nn) for nn in inner)] = vv
        if gaps is not None:
            gaps[...] = (w[1:] - w[:-1]).T.reshape(inner + [m._nsta - 1])
    else:
        for idx in np.ndindex(*inner):
            kpt = [start_k[d] + float(idx[d]) / float(mesh[d] - 1) for d in range(D)]
            w, v = solve_one(m, kpt, True)
            wfs[idx] = v
            if gaps is not None:
                gaps[idx] = w[1:] - w[:-1]
    for d in range(D):
        impose_pbc(m, wfs, d, m._per[d])
    if gaps is None:
        return wfs, None
    return wfs, gaps.min(axis=tuple(range(D)))


# --------------------------------------------------------------------------
# Berry phases and fluxes
# --------------------------------------------------------------------------

def wf_dpr(a, b):
    """<a|b> over orbital(,spin).  pythtb.py:3793-3796."""
    return np.dot(np.conj(a).reshape(-1), np.asarray(b).reshape(-1))


def one_berry_loop(wf, berry_evals=False):
    """Discrete Berry phase of one string wf[kpnt,band,orb(,spin)] over its
    N-1 links.  Follows pythtb.py:3798-3838."""
    npts, nocc = wf.shape[0], wf.shape[1]
    prod = np.identity(nocc, dtype=complex)
    for i in range(npts - 1):
        ovr = np.zeros((nocc, nocc), dtype=complex)
        for a in range(nocc):
            for b in range(nocc):
                ovr[a, b] = wf_dpr(wf[i, a], wf[i + 1, b])
        if berry_evals:
            u, _, vh = np.linalg.svd(ovr)
            prod = prod @ (u @ vh)
        else:
            prod = prod @ ovr
    if berry_evals:
        return np.sort(-np.angle(np.linalg.eigvals(prod)))
    return -np.angle(np.linalg.det(prod))


def one_flux_plane(wfs2d):
    """Plaquette Berry phases of wfs2d[k0,k1,band,...].  pythtb.py:3840-3865."""
    n0, n1 = wfs2d.shape[0], wfs2d.shape[1]
    out = np.zeros((n0 - 1, n1 - 1), dtype=float)
    for i in range(n0 - 1):
        for j in range(n1 - 1):
            loop = np.array([wfs2d[i, j], wfs2d[i + 1, j], wfs2d[i + 1, j + 1],
                             wfs2d[i, j + 1], wfs2d[i, j]], dtype=complex)
            out[i, j] = one_berry_loop(loop)
    return out


def one_flux_plane_vec(wfs2d):
    """Vectorised checker: same 4-link matrix product per plaquette as
    one_flux_plane, batched over the plane."""
    n0, n1, nocc = wfs2d.shape[:3]
    w = wfs2d.reshape(n0, n1, nocc, -1)
    c = [w[:-1, :-1], w[1:, :-1], w[1:, 1:], w[:-1, 1:], w[:-1, :-1]]
    prod = None
    for s in range(4):
        ovr = np.einsum("ijao,ijbo->ijab", np.conj(c[s]), c[s + 1])
        prod = ovr if prod is None else prod @ ovr
    return -np.angle(np.linalg.det(prod))


def no_2pi(x, clos):
    """pythtb.py:3867-3874."""
    while abs(clos - x) > np.pi:
        if clos - x > np.pi:
            x += TWO_PI
        elif clos - x < -np.pi:
            x -= TWO_PI
    return x


def one_phase_cont(pha, clos):
    """pythtb.py:3876-3889."""
    out = np.array(pha, dtype=float, copy=True)
    for i in range(len(out)):
        out[i] = no_2pi(out[i], clos if i == 0 else out[i - 1])
    return out


def array_phases_cont(arr, clos):
    """Greedy nearest matching on the unit circle.  pythtb.py:3891-3921."""
    out = np.zeros_like(arr)
    for i in range(arr.shape[0]):
        ref = clos if i == 0 else out[i - 1]
        free = list(range(arr.shape[1]))
        for j in range(ref.shape[0]):
            best, best_d = None, 1.0e10
            for k in free:
                d = np.abs(np.exp(1j * ref[j]) - np.exp(1j * arr[i, k]))
                if d <= best_d:
                    best, best_d = k, d
            free.remove(best)
            out[i, j] = no_2pi(arr[i, best], ref[j])
    return out


def _occ_index(occ, nsta_arr):
    if occ is None or (isinstance(occ, str) and occ == "All"):
        return np.arange(nsta_arr, dtype=int)
    return np.array(occ, dtype=int)


def berry_phase(wfs, dim_arr, occ="All", dir=None, contin=True, berry_evals=False):
    """Berry phase per string along `dir`.  Follows pythtb.py:2863-3066."""
    occ = _occ_index(occ, wfs.shape[dim_arr])
    if occ.ndim != 1:
        raise Exception('\n\nParameter occ must be a one-dimensional array or string "All" or None.')
    if dim_arr == 1:
        ret = one_berry_loop(wfs[:, occ], berry_evals)
    elif dim_arr in (2, 3):
        if dir is None or dir < 0 or dir >= dim_arr:
            raise Exception("\n\nWrong direction for Berry phase calculation!")
        mv = np.moveaxis(wfs, dir, dim_arr - 1)       # other axes keep their order
        other = mv.shape[:dim_arr - 1]
        flat = []
        for idx in np.ndindex(*other):
            flat.append(one_berry_loop(mv[idx][:, occ], berry_evals))
        ret = np.array(flat, dtype=float).reshape(list(other) + ([len(occ)] if berry_evals else []))
    else:
        raise Exception("\n\nWrong dimensionality!")
    if dim_arr > 1 or berry_evals:
        ret = np.array(ret, dtype=float)
    if contin:
        if not berry_evals:
            if dim_arr == 2:
                ret = one_phase_cont(ret, ret[0])
            elif dim_arr == 3:
                for i in range(ret.shape[1]):
                    clos = ret[0, 0] if i == 0 else ret[0, i - 1]
                    ret[:, i] = one_phase_cont(ret[:, i], clos)
        else:
            if dim_arr == 2:
                ret = array_phases_cont(ret, ret[0, :])
            elif dim_arr == 3:
                for i in range(ret.shape[1]):
                    clos = ret[0, 0, :] if i == 0 else ret[0, i - 1, :]
                    ret[:, i] = array_phases_cont(ret[:, i], clos)
    return ret


def berry_flux(wfs, dim_arr, occ="All", dirs=None, individual_phases=False, vectorised=False):
    """Plaquette fluxes on the (dirs[0],dirs[1]) planes.  pythtb.py:3068-3205."""
    occ = _occ_index(occ, wfs.shape[dim_arr])
    if dirs is None:
        dirs = [0, 1]
    if dirs[0] == dirs[1]:
        raise Exception("Need to specify two different directions for Berry flux calculation.")
    if max(dirs) >= dim_arr or min(dirs) < 0:
        raise Exception("Direction for Berry flux calculation out of bounds.")
    plane = one_flux_plane_vec if vectorised else one_flux_plane
    rest = [d for d in range(dim_arr) if d not in dirs]
    order = [dirs[0], dirs[1]] + rest + list(range(dim_arr, wfs.ndim))
    use = wfs.transpose(order)
    if dim_arr == 2:
        ph = plane(use[:, :, occ])
        return ph if individual_phases else ph.sum()
    if dim_arr not in (3, 4):
        raise Exception("\n\nWrong dimensionality!")
    rest_shape = [wfs.shape[d] for d in rest]
    out = np.zeros(rest_shape + [wfs.shape[dirs[0]] - 1, wfs.shape[dirs[1]] - 1], dtype=float)
    for idx in np.ndindex(*rest_shape):
        sl = use[(slice(None), slice(None)) + idx]
        out[idx] = plane(sl[:, :, occ])
    return out if individual_phases else out.sum(axis=(-2, -1))


# --------------------------------------------------------------------------
# Position operator / hybrid Wannier functions
# --------------------------------------------------------------------------

def position_matrix(m, evec, dir):
    """X_mn = <u_m| r_dir |u_n> for the states evec[band,orb(,spin)].  pythtb.py:2034-2098."""
    if dir in m._per:
        raise Exception("Can not compute position matrix elements along periodic direction!")
    if dir < 0 or dir >= m._dim_r:
        raise Exception("Direction out of range!")
    pos = np.repeat(m._orb[:, dir], m._nspin)
    ev = np.asarray(evec).reshape(np.asarray(evec).shape[0], -1)
    nb = ev.shape[0]
    out = np.zeros((nb, nb), dtype=complex)
    for a in range(nb):
        for b in range(nb):
            out[a, b] = np.dot(ev[a].conj(), pos * ev[b])
    if np.max(out - out.T.conj()) > 1.0e-9:
        raise Exception("\n\n Position matrix is not hermitian?!")
    return out


def position_expectation(m, evec, dir):
    """pythtb.py:2100-2141."""
    return np.array(np.real(position_matrix(m, evec, dir).diagonal()), dtype=float)


def position_hwf(m, evec, dir, hwf_evec=False, basis="orbital"):
    """Eigen-decomposition of the position matrix.  pythtb.py:2143-2279."""
    x = position_matrix(m, evec, dir)
    if not hwf_evec:
        return np.sort(np.array(np.linalg.eigvalsh(x).real, dtype=float))
    w, v = np.linalg.eigh(x)
    order = np.argsort(np.array(w.real, dtype=float))
    w = np.array(w.real, dtype=float)[order]
    rows = v.T[order]
    which = basis.lower().strip()
    if which in ("wavefunction", "bloch"):
        return w, rows
    if which != "orbital":
        raise Exception("\n\nBasis must be either 'wavefunction', 'bloch', or 'orbital'")
    ev = np.asarray(evec)
    flat = ev.reshape(ev.shape[0], -1)
    orb = rows @ flat
    if m._nspin == 2:
        orb = orb.reshape(rows.shape[0], m._norb, 2)
    return w, orb


# --------------------------------------------------------------------------
# Synthetic model builders used by tests / bench (definitions: SURVEY.md 8d)
# --------------------------------------------------------------------------

def _build(dim_k, dim_r, lat, orb, nspin, onsite, hops, per=None):
    """Assemble a Model applying the reference's set_onsite/set_hop storage
    rules for plain 'set' calls (pythtb.py:272,:473-478,:538-550)."""
    orb = np.array(orb, dtype=float)
    no = orb.shape[0]

    def block(v):
        if nspin == 1:
            return v
        a = np.array(v)
        out = np.zeros((2, 2), dtype=complex)
        if a.shape == ():
            out[0, 0] = out[1, 1] = a
        elif a.shape == (4,):
            out[0, 0] = a[0] + a[3]
            out[1, 1] = a[0] - a[3]
            out[0, 1] = a[1] - 1j * a[2]
            out[1, 0] = a[1] + 1j * a[2]
        else:
            out = np.array(a, dtype=complex)
        return out

    if nspin == 1:
        site = np.array([float(np.real(x)) for x in onsite], dtype=float)
    else:
        site = np.array([block(x) for x in onsite], dtype=complex)
    hl = []
    for amp, a, b, R in hops:
        hl.append([block(amp), int(a), int(b), np.array(R, dtype=int)])
    if per is None:
        per = list(range(dim_k))
    return Model(dim_k, dim_r, lat, orb, per, nspin, site, hl)


HONEYCOMB_LAT = [[1.0, 0.0], [0.5, np.sqrt(3.0) / 2.0]]
HONEYCOMB_ORB = [[1.0 / 3.0, 1.0 / 3.0], [2.0 / 3.0, 2.0 / 3.0]]


def graphene(delta=0.0, t=-1.0):
    """examples/graphene.py:14-31."""
    hops = [(t, 0, 1, [0, 0]), (t, 1, 0, [1, 0]), (t, 1, 0, [0, 1])]
    return _build(2, 2, HONEYCOMB_LAT, HONEYCOMB_ORB, 1, [-delta, delta], hops)


def haldane(delta=0.0, t=-1.0, t2=0.15):
    """examples/haldane_bp.py:16-41 (delta=0) / examples/haldane.py (delta=0.2)."""
    t2c = t2 * np.exp(1j * np.pi / 2.0)
    t2cc = np.conj(t2c)
    hops = [(t, 0, 1, [0, 0]), (t, 1, 0, [1, 0]), (t, 1, 0, [0, 1]),
            (t2c, 0, 0, [1, 0]), (t2c, 1, 1, [1, -1]), (t2c, 1, 1, [0, 1]),
            (t2cc, 1, 1, [1, 0]), (t2cc, 0, 0, [1, -1]), (t2cc, 0, 0, [0, 1])]
    return _build(2, 2, HONEYCOMB_LAT, HONEYCOMB_ORB, 1, [-delta, delta], hops)


def kane_mele(topological="odd"):
    """examples/kane_mele.py:14-67 (the three Rashba 'add' calls folded into
    the matching first-neighbour entries, as set_hop mode='add' does :509-511)."""
    esite = 2.5 if topological == "even" else 1.0
    thop = 1.0
    so = 0.6 * thop * 0.5
    ra = 0.25 * thop
    sx = np.array([0.0, 1.0, 0.0, 0.0])
    sy = np.array([0.0, 0.0, 1.0, 0.0])
    sz = np.array([0.0, 0.0, 0.0, 1.0])
    r3h = np.sqrt(3.0) / 2.0
    one = np.array([1.0, 0.0, 0.0, 0.0])
    hops = [(thop * one + 1j * ra * (0.5 * sx - r3h * sy), 0, 1, [0, 0]),
            (thop * one + 1j * ra * (-1.0 * sx), 0, 1, [0, -1]),
            (thop * one + 1j * ra * (0.5 * sx + r3h * sy), 0, 1, [-1, 0]),
            (-1j * so * sz, 0, 0, [0, 1]), (1j * so * sz, 0, 0, [1, 0]), (-1j * so * sz, 0, 0, [1, -1]),
            (1j * so * sz, 1, 1, [0, 1]), (-1j * so * sz, 1, 1, [1, 0]), (1j * so * sz, 1, 1, [1, -1])]
    return _build(2, 2, HONEYCOMB_LAT, HONEYCOMB_ORB, 2, [esite, -esite], hops)


def three_site_chain(t, delta, lmbd):
    """tests/test_examples/three_site/3site_cycle/run.py:4-17."""
    hops = [(t, 0, 1, [0]), (t, 1, 2, [0]), (t, 2, 0, [1])]
    onsite = [delta * -np.cos(TWO_PI * (lmbd - i / 3.0)) for i in range(3)]
    return _build(1, 1, [[1.0]], [[0.0], [1.0 / 3.0], [2.0 / 3.0]], 1, onsite, hops)


def cubic16(seed=0):
    """Synthetic 3-D 16-orbital model, 888 hoppings (SURVEY.md 8d recipe)."""
    rng = np.random.default_rng(seed)
    orb = rng.random((16, 3))
    onsite = np.where(np.arange(16) < 8, -2.0, 2.0) + 0.2 * rng.standard_normal(16)
    hops = []
    for a in range(16):
        for b in range(a + 1, 16):
            hops.append((0.1 * (rng.standard_normal() + 1j * rng.standard_normal()), a, b, [0, 0, 0]))
    for R in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
        for a in range(16):
            for b in range(16):
                hops.append((0.1 * (rng.standard_normal() + 1j * rng.standard_normal()), a, b, R))
    return _build(3, 3, np.identity(3), orb, 1, onsite, hops)
