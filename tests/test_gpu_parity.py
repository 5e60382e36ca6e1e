"""GPU parity: the HIP path (through the C ABI, via the tb_model / wf_array mirror)
against golden vectors captured from the reference and against the CPU oracle.

Tolerances (fp64; BASELINE.json asks for eigenvalues within 1e-10 of NumPy):
  H(k) entries       1e-13      eigenvalues     1e-12
  plaquette / string phases, fluxes, Wilson-loop eigenphases   1e-10
Eigenvectors are gauge dependent and never compared raw: residual, orthonormality
and gauge-invariant Berry quantities are checked instead.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_tables, load_golden
import helpers as hp

pytestmark = pytest.mark.gpu

TOL_H = 1e-13
TOL_E = 1e-12
TOL_P = 1e-10

POINT_CASES = ["graphene", "haldane0", "haldane02", "km_odd", "km_even", "chain3", "per02",
               "molecule", "spin_chain", "cubic16"]


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    pythtb_amd._lib.default_context()        # fail loudly here if no GPU / no library
    return pythtb_amd


def wrap(d):
    return (np.asarray(d) + np.pi) % (2 * np.pi) - np.pi


def assert_phase_sets_close(a, b, tol):
    a = np.asarray(a).reshape(-1, np.asarray(a).shape[-1])
    b = np.asarray(b).reshape(a.shape)
    for ra, rb in zip(a, b):
        free = list(range(len(rb)))
        for x in ra:
            d = [abs(wrap(x - rb[j])) for j in free]
            j = int(np.argmin(d))
            assert d[j] < tol, (ra, rb)
            free.pop(j)


# ------------------------------------------------------------------ kernels 1+2
@pytest.mark.parametrize("name", POINT_CASES)
def test_gen_ham_and_eigenvalues_match_reference(tb, name):
    g = load_golden("point_" + name)
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    n = m._nsta
    if m._dim_k == 0:
        ham = m._gen_ham().reshape(n, n)
        assert np.max(np.abs(ham - g["ham"][0])) < TOL_H
        ev = m.solve_all()
        assert ev.shape == (n,)
        assert np.max(np.abs(ev - g["evals"][:, 0])) < TOL_E
        ev2, vec = m.solve_all(eig_vectors=True)
        assert vec.shape == ((n, m._norb) if m._nspin == 1 else (n, m._norb, 2))
        V = vec.reshape(n, n)
        assert np.max(np.abs(g["ham"][0] @ V.T - V.T * ev2)) < 1e-12
        return
    k = g["k"]
    for ik in range(0, len(k), 7):
        ham = m._gen_ham(k[ik]).reshape(n, n)
        assert np.max(np.abs(ham - g["ham"][ik])) < TOL_H
    ev = m.solve_all(k)
    assert ev.shape == (n, len(k)) and ev.flags["C_CONTIGUOUS"]
    assert np.max(np.abs(ev - g["evals"])) < TOL_E
    ev2, vec = m.solve_all(k, eig_vectors=True)
    assert np.max(np.abs(ev2 - g["evals"])) < TOL_E
    assert vec.shape == ((n, len(k), m._norb) if m._nspin == 1 else (n, len(k), m._norb, 2))
    V = vec.reshape(n, len(k), n)
    for ik in range(len(k)):
        Vk = V[:, ik, :]                                   # rows are eigenvectors
        assert np.max(np.abs(Vk.conj() @ Vk.T - np.identity(n))) < 1e-13
        assert np.max(np.abs(g["ham"][ik] @ Vk.T - Vk.T * ev2[:, ik])) < 1e-12
    # solve_one and _sol_ham on the same matrices
    e1, v1 = m.solve_one(k[3], eig_vectors=True)
    assert np.max(np.abs(e1 - g["evals"][:, 3])) < TOL_E and v1.shape == vec[:, 0].shape
    hshape = (m._norb, m._norb) if m._nspin == 1 else (m._norb, 2, m._norb, 2)
    e3 = m._sol_ham(g["ham"][5].reshape(hshape))
    assert np.max(np.abs(e3 - g["evals"][:, 5])) < TOL_E
    e4, v4 = m._sol_ham(g["ham"][5].reshape(hshape), eig_vectors=True)
    V4 = v4.reshape(n, n)
    assert np.max(np.abs(g["ham"][5] @ V4.T - V4.T * e4)) < 1e-12


def test_sol_ham_rejects_non_hermitian(tb):
    m = hp.haldane(tb.tb_model)
    with pytest.raises(Exception, match="not hermitian"):
        m._sol_ham(np.array([[0.0, 1.0], [0.5, 0.0]], dtype=complex))


def test_eigh_batch_random_sizes_vs_numpy(tb):
    """_sol_ham kernel on random Hermitian matrices, every size the build supports
    up to 24 plus 33 and 64 (register path n<=4, wavefront path above)."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(7)
    for n in list(range(1, 25)) + [33, 64, 65, 100, 137, 256]:   # > 64: workgroup-per-matrix kernel
        nk = 37 if n <= 24 else 5
        a = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = np.ascontiguousarray(a + np.transpose(a.conj(), (0, 2, 1)))
        if n >= 4:                                          # exact degeneracies and a zero matrix
            h[0] = np.diag([1.0] * (n // 2) + [2.0] * (n - n // 2))
            h[1] = 0.0
        ev = np.zeros((n, nk))
        vec = np.zeros((n, nk, n), dtype=complex)
        _lib.check(_lib.lib.tbk_eigh_batch(_lib.default_context().handle, n, _lib.dptr(h.view(float)), nk,
                                           _lib.dptr(ev), _lib.dptr(vec.view(float))))
        ref = np.linalg.eigvalsh(h).T
        scale = max(1.0, np.abs(ref).max())
        assert np.max(np.abs(ev - ref)) < 2e-13 * scale * n, n
        for ik in range(nk):
            V = vec[:, ik, :]
            assert np.max(np.abs(V.conj() @ V.T - np.identity(n))) < 1e-12, n
            assert np.max(np.abs(h[ik] @ V.T - V.T * ev[:, ik])) < 1e-11 * scale, n


def test_solve_all_input_forms(tb):
    """1-D models accept a flat list of scalars; lists of lists and arrays agree."""
    m = hp.chain3(tb.tb_model, -1.0, 2.0, 0.3)
    ks = [0.0, 0.1, 0.37, -0.2]
    a = m.solve_all(ks)
    b = m.solve_all([[x] for x in ks])
    c = m.solve_all(np.array(ks).reshape(-1, 1))
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.array_equal(m.solve_one(0.37), a[:, 2])
    with pytest.raises(Exception, match="wrong shape"):
        hp.haldane(tb.tb_model).solve_all([[0.1, 0.2, 0.3]])


def test_solve_all_deterministic(tb):
    m = hp.kane_mele(tb.tb_model)
    k = np.random.default_rng(1).random((300, 2))
    e1, v1 = m.solve_all(k, eig_vectors=True)
    e2, v2 = m.solve_all(k, eig_vectors=True)
    assert np.array_equal(e1, e2) and np.array_equal(v1, v2)


# ------------------------------------------------------------------ grids, flux, phases
GRID_CASES = {
    "haldane0_33": dict(occs=[0, 1, 2], dirs=[0, 1], flux=[(0, 1), (1, 0)]),
    "haldane02_20x28": dict(occs=[0, 1], dirs=[0, 1], flux=[(0, 1)]),
    "km_odd_65x33": dict(occs=[0], dirs=[0, 1], flux=[(0, 1)]),     # other occ sets split Kramers pairs
    "km_even_41": dict(occs=[0], dirs=[1], flux=[(0, 1)]),
    "chain3_41": dict(occs=[0, 1, 2], dirs=[0], flux=[]),
    "per02_11": dict(occs=[0, 1], dirs=[0, 1], flux=[(0, 1)]),
    "spin_chain_25": dict(occs=[0, 1], dirs=[0], flux=[]),
    "cubic16_9": dict(occs=[0, 1], dirs=[0, 1, 2], flux=[(0, 1), (1, 2), (2, 0)]),
    "quad4_4354": dict(occs=[0, 1], dirs=[], flux=[(0, 1), (2, 3), (3, 1)]),     # 4-D: flux only (pythtb.py:3028-3029)
}


@pytest.mark.parametrize("name", sorted(GRID_CASES))
def test_grid_flux_phase_match_reference(tb, name):
    g = load_golden("grid_" + name)
    spec = GRID_CASES[name]
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    mesh = [int(x) for x in g["mesh"]]
    w = tb.wf_array(m, mesh)
    gaps = w.solve_on_grid(list(g["start_k"]))
    if m._nsta > 1:
        assert gaps.shape == (m._nsta - 1,)
        assert np.max(np.abs(gaps - g["min_gaps"])) < TOL_E
    else:
        assert gaps is None
    dim = len(mesh)
    for io in spec["occs"]:
        occ = [int(x) for x in g["occ%d" % io]]
        for (d0, d1) in spec["flux"]:
            tag = "occ%d_d%d%d" % (io, d0, d1)
            plaq = w.berry_flux(occ, dirs=[d0, d1], individual_phases=True)
            assert plaq.shape == g["flux_plaq_" + tag].shape
            assert np.max(np.abs(wrap(plaq - g["flux_plaq_" + tag]))) < TOL_P
            tot = w.berry_flux(occ, dirs=[d0, d1])
            assert np.shape(tot) == g["flux_tot_" + tag].shape
            assert np.max(np.abs(tot - g["flux_tot_" + tag])) < 1e-9
        for d in spec["dirs"]:
            for contin in (True, False):
                ref = g["phase_occ%d_dir%d_c%d_e0" % (io, d, int(contin))]
                got = w.berry_phase(occ, d if dim > 1 else None, contin=contin)
                assert np.shape(got) == ref.shape
                assert np.max(np.abs(wrap(got - ref))) < TOL_P, (name, io, d, contin)
                ref = g["phase_occ%d_dir%d_c%d_e1" % (io, d, int(contin))]
                got = w.berry_phase(occ, d if dim > 1 else None, contin=contin, berry_evals=True)
                assert np.shape(got) == ref.shape
                assert_phase_sets_close(got, ref, 1e-9)
    if dim == 4:
        with pytest.raises(Exception, match="Wrong dimensionality"):
            w.berry_phase([0], 1)
    # the device array and its host mirror agree with impose_pbc semantics
    host = w._wfs
    assert host.shape == tuple(mesh + [m._nsta, m._norb] + ([2] if m._nspin == 2 else []))
    for d in range(dim):
        fac = np.repeat(np.exp(-2j * np.pi * m._orb[:, m._per[d]]), m._nspin).reshape(host.shape[dim + 1:])
        first = np.take(host, 0, axis=d)
        last = np.take(host, -1, axis=d)
        assert np.max(np.abs(last - first * fac)) < 1e-14


def test_contin_matches_reference_exactly_when_no_branch_issue(tb):
    g = load_golden("grid_haldane0_33")
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    w = tb.wf_array(m, [33, 33])
    w.solve_on_grid([-0.5, -0.5])
    for d in (0, 1):
        got = w.berry_phase([0], d, contin=True)
        ref = g["phase_occ0_dir%d_c1_e0" % d]
        assert np.max(np.abs(got - ref)) < TOL_P            # same 2 pi branch, not only mod 2 pi
    assert abs(w.berry_flux([0]) / (2 * np.pi) + 1.0) < 1e-12   # Chern number -1


def test_manual_fill_cone_and_3site(tb):
    """wf_arrays filled through [] / solve_on_one_point (reference tests cone, 3site_cycle)."""
    g = load_golden("manual_cone")
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    n = 31
    w = tb.wf_array(m, [n])
    for i in range(n):
        w.solve_on_one_point(g["circ_k"][i], i)
    w[-1] = w[0]
    got = np.array([w.berry_phase([0], 0), w.berry_phase([1], 0), w.berry_phase([0, 1], 0)])
    assert np.max(np.abs(wrap(got - g["circ_phase"]))) < TOL_P
    ws = tb.wf_array(m, [n, n])
    for i in range(n):
        for j in range(n):
            ws[i, j] = m.solve_one(g["sq_k"][i, j], eig_vectors=True)[1]
    got = np.array([ws.berry_flux([0]), ws.berry_flux([1]), ws.berry_flux([0, 1])])
    assert np.max(np.abs(got - g["sq_flux"])) < 1e-9
    assert np.max(np.abs(ws.berry_flux([0], individual_phases=True) - g["sq_plaq"])) < TOL_P

    g = load_golden("manual_3site")
    t, delta, lam, nkp = float(g["t"]), float(g["delta"]), g["lam"], int(g["nkp"])
    w = tb.wf_array(hp.chain3(tb.tb_model, t, delta, 0.0), [nkp, len(lam)])
    for il, lm in enumerate(lam):
        mm = hp.chain3(tb.tb_model, t, delta, lm)
        kv, _, _ = mm.k_path([[-0.5], [0.5]], nkp, report=False)
        _, evec = mm.solve_all(kv, eig_vectors=True)
        for ik in range(nkp):
            w[ik, il] = evec[:, ik, :]
    w.impose_pbc(0, 0)
    assert np.max(np.abs(wrap(w.berry_phase([0], 0) - g["wann"] * 2 * np.pi))) < TOL_P
    assert abs(w.berry_flux([0]) - g["flux"]) < 1e-9
    got = np.array([w.berry_flux([0]), w.berry_flux([1]), w.berry_flux([2]),
                    w.berry_flux([0, 1]), w.berry_flux([0, 1, 2])])
    assert np.max(np.abs(got - g["flux_all"])) < 1e-9


def test_wf_array_api_contract(tb):
    m = hp.haldane(tb.tb_model)
    w = tb.wf_array(m, [9, 7])
    w.solve_on_grid([0.0, 0.0])
    assert w[2, 3].shape == (2, 2) and w[-1, -1].shape == (2, 2)
    with pytest.raises(IndexError):
        w[9, 0]
    with pytest.raises(TypeError):
        w[1]
    with pytest.raises(Exception, match="Wrong direction"):
        w.berry_phase([0])
    with pytest.raises(Exception, match="two different directions"):
        w.berry_flux([0], dirs=[1, 1])
    sub = w.choose_states([1])
    assert sub._wfs.shape == (9, 7, 1, 2)
    assert np.allclose(sub._wfs[:, :, 0], w._wfs[:, :, 1])
    assert abs(sub.berry_flux([0]) - w.berry_flux([1])) < 1e-10
    assert w.berry_flux("All") == w.berry_flux(None) == w.berry_flux(range(2))
    e = w.empty_like(nsta_arr=5)
    assert e._wfs.shape == (9, 7, 5, 2)
    with pytest.raises(Exception, match="2 or larger"):
        tb.wf_array(m, [1, 5])
    with pytest.raises(Exception, match="nsta_arr"):
        tb.wf_array(m, [5, 5], nsta_arr=1).solve_on_grid([0, 0])
    # impose_loop copies without phases
    w2 = tb.wf_array(m, [5, 5])
    w2.solve_on_grid([0.0, 0.0])
    w2.impose_loop(1)
    assert np.array_equal(w2._wfs[:, -1], w2._wfs[:, 0])


# ------------------------------------------------------------------ the reference's own golden files
REF = os.path.join(GOLDEN, "reference_tests")


def ref_npy(group, name):
    return np.load(os.path.join(REF, group, name))


def test_reference_goldens_band_structures(tb):
    atol, rtol = 1e-12, 1e-8                               # reference tests: rtol=1e-8, atol=1e-14
    g = hp.graphene(tb.tb_model, delta=0.0)
    path = [[0., 0.], [2. / 3., 1. / 3.], [.5, .5], [0., 0.]]
    kv, _, _ = g.k_path(path, 121, report=False)
    np.testing.assert_allclose(g.solve_all(kv), ref_npy("graphene", "evals.npy"), rtol=rtol, atol=atol)

    h = hp.haldane(tb.tb_model, delta=0.2)
    path = [[0., 0.], [2. / 3., 1. / 3.], [.5, .5], [1. / 3., 2. / 3.], [0., 0.]]
    kv, _, _ = h.k_path(path, 101, report=False)
    np.testing.assert_allclose(h.solve_all(kv), ref_npy("haldane", "evals.npy"), rtol=rtol, atol=atol)
    kp = [[i / 20.0, j / 20.0] for i in range(20) for j in range(20)]
    np.testing.assert_allclose(h.solve_all(kp).flatten(), ref_npy("haldane", "evals_dos.npy"), rtol=rtol, atol=atol)

    ev = np.array([hp.kane_mele(tb.tb_model, t).solve_all(kv) for t in ("even", "odd")])
    np.testing.assert_allclose(ev, ref_npy("kane_mele", "kane_mele_evals.npy"), rtol=rtol, atol=atol)

    cb = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.0, 1.0]], [[0.0, 0.0], [0.5, 0.5]])
    cb.set_onsite([-1.1, 1.1])
    for R in ([0, 0], [1, 0], [0, 1], [1, 1]):
        cb.set_hop(0.6, 1, 0, R)
    kv2, _, _ = cb.k_path([[0.0, 0.0], [0.0, 0.5], [0.5, 0.5], [0.0, 0.0]], 301, report=False)
    np.testing.assert_allclose(cb.solve_all(kv2), ref_npy("checkerboard", "evals.npy"), rtol=rtol, atol=atol)

    tr = hp.quiet(tb.tb_model, 1, 2, [[2.0, 0.0], [0.0, 1.0]], [[0.0, 0.0], [0.5, 1.0]], per=[0])
    tr.set_hop(2.0, 0, 0, [1, 0])
    tr.set_hop(2.0, 1, 1, [1, 0])
    tr.set_hop(0.8 + 0.6j, 0, 1, [0, 0])
    tr.set_hop(0.8 + 0.6j, 1, 0, [1, 0])
    kv3, _, _ = tr.k_path("fullc", 100, report=False)
    np.testing.assert_allclose(tr.solve_all(kv3), ref_npy("trestle", "evals.npy"), rtol=rtol, atol=atol)

    bl = hp.quiet(tb.tb_model, 2, 3, [[1.0, 0.0, 0.0], [0.0, 1.25, 0.0], [0.0, 0.0, 3.0]],
                  [[0.0, 0.0, -0.15], [0.5, 0.5, 0.15]])
    bl.set_onsite([-1.1, 1.1])
    for R in ([0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]):
        bl.set_hop(0.6, 1, 0, R)
    kv4, _, _ = bl.k_path([[0.0, 0.0], [0.0, 0.5], [0.5, 0.5], [0.0, 0.0]], 81, report=False)
    np.testing.assert_allclose(bl.solve_all(kv4), ref_npy("buckled_layer", "evals.npy"), rtol=rtol, atol=atol)

    sq32 = np.sqrt(3.0) / 2.0
    mol = hp.quiet(tb.tb_model, 0, 3, np.identity(3),
                   [[(2.0 / 3.0) * sq32, 0.0, 0.0], [(-1.0 / 3.0) * sq32, 0.5, 0.0],
                    [(-1.0 / 3.0) * sq32, -0.5, 0.0], [0.0, 0.0, 1.0]])
    mol.set_onsite([-0.5, -0.5, -0.5, 0.5])
    for i in range(4):
        for j in range(i + 1, 4):
            mol.set_hop(1.0, i, j)
    np.testing.assert_allclose(mol.solve_all(), ref_npy("0dim", "evals.npy"), rtol=rtol, atol=atol)


def test_reference_goldens_berry(tb):
    # haldane_bp (tests/test_examples/haldane/haldane_bp/run.py): 31x31, start -0.5
    m = hp.haldane(tb.tb_model, delta=0.0)
    w = tb.wf_array(m, [31, 31])
    w.solve_on_grid([-0.5, -0.5])
    for name, occ in (("phi_a1", [0]), ("phi_b1", [1]), ("phi_c1", [0, 1])):
        got = w.berry_phase(occ, 0, contin=True)
        assert np.max(np.abs(got - ref_npy("haldane_bp", name + ".npy"))) < TOL_P
    assert abs(w.berry_flux([0]) - ref_npy("haldane_bp", "flux_a1.npy")) < 1e-9
    assert abs(ref_npy("haldane_bp", "flux_a2.npy") + 2 * np.pi) < 1e-9
    # kane_mele wan_cent: berry_phase([0,1],dir=1,contin=False,berry_evals=True)/2pi on 41x41
    cents = []
    for top in ("even", "odd"):
        wk = tb.wf_array(hp.kane_mele(tb.tb_model, top), [41, 41])
        wk.solve_on_grid([-0.5, -0.5])
        cents.append(wk.berry_phase([0, 1], dir=1, contin=False, berry_evals=True) / (2 * np.pi))
    ref = ref_npy("kane_mele", "kane_mele_wan_cent.npy")
    for a, b in zip(cents, ref):
        assert_phase_sets_close(a * 2 * np.pi, b * 2 * np.pi, 1e-9)
    # cone goldens
    gc = load_golden("manual_cone")
    mc = hp.graphene(tb.tb_model, delta=-0.1)
    wc = tb.wf_array(mc, [31])
    for i in range(31):
        wc.solve_on_one_point(gc["circ_k"][i], i)
    wc[-1] = wc[0]
    for name, occ in (("bphase_circ0", [0]), ("bphase_circ1", [1]), ("bphase_circ01", [0, 1])):
        assert abs(wrap(wc.berry_phase(occ, 0) - ref_npy("cone", name + ".npy"))) < TOL_P
    # 3site_cycle_fin fluxes: (lambda, k) array, 5 occupations
    t, delta = -1.3, 2.0
    lam = np.linspace(0.0, 1.0, 21, endpoint=True)
    kv, _, _ = hp.chain3(tb.tb_model, t, delta, 0.0).k_path([[-0.5], [0.5]], 31, report=False)
    wl = tb.wf_array(hp.chain3(tb.tb_model, t, delta, 0.0), [21, 31])
    for il, lm in enumerate(lam):
        _, evec = hp.chain3(tb.tb_model, t, delta, lm).solve_all(kv, eig_vectors=True)
        for ik in range(31):
            wl[il, ik] = evec[:, ik, :]
    got = np.array([wl.berry_flux([0]), wl.berry_flux([1]), wl.berry_flux([2]),
                    wl.berry_flux([0, 1]), wl.berry_flux([0, 1, 2])])
    assert np.max(np.abs(got - ref_npy("3site_cycle_fin", "3site_cycle_fluxes.npy"))) < 1e-9


# ------------------------------------------------------------------ oracle at larger / random sizes
@pytest.mark.parametrize("norb,dim_k,nspin,seed", [
    (1, 1, 1, 0), (2, 2, 1, 1), (3, 3, 1, 2), (4, 2, 1, 3), (1, 2, 2, 4), (2, 3, 2, 5),
    (5, 2, 1, 6), (3, 1, 2, 7), (7, 3, 1, 8), (4, 2, 2, 9), (11, 2, 1, 10), (16, 3, 1, 11),
    (3, 4, 1, 12), (20, 1, 1, 13), (2, 0, 2, 14),
])
def test_random_models_vs_oracle(tb, norb, dim_k, nspin, seed):
    from oracle import tb_oracle as orc
    m = hp.random_model(tb.tb_model, norb, dim_k, nspin, seed)
    n = m._nsta
    if dim_k == 0:
        assert np.max(np.abs(m.solve_all() - orc.solve_all(m))) < TOL_E * 10
        assert np.max(np.abs(m._gen_ham() - orc.gen_ham(m))) < TOL_H * 10
        return
    k = np.random.default_rng(seed + 100).uniform(-1.5, 1.5, size=(200, dim_k))
    ham = orc.ham_batch(m, k)
    scale = max(1.0, np.abs(ham).max())
    for ik in (0, 17, 199):
        assert np.max(np.abs(m._gen_ham(k[ik]).reshape(n, n) - ham[ik])) < TOL_H * scale * 10
    ev, vec = m.solve_all(k, eig_vectors=True)
    ref = orc.solve_all_vec(m, k)
    assert np.max(np.abs(ev - ref)) < TOL_E * scale * 10
    assert np.max(np.abs(ev - m.solve_all(k))) < 1e-13 * scale   # eigenvalue-only kernel variant
    V = vec.reshape(n, len(k), n)
    for ik in range(0, len(k), 13):
        assert np.max(np.abs(ham[ik] @ V[:, ik].T - V[:, ik].T * ev[:, ik])) < 1e-11 * scale


@pytest.mark.parametrize("case", ["haldane", "km", "chain3x", "cubic16"])
def test_mesh_berry_vs_oracle(tb, case):
    """solve_on_grid + berry_* against the oracle run on the oracle's own eigenvectors."""
    from oracle import tb_oracle as orc
    if case == "haldane":
        m, mesh, start, occs = hp.haldane(tb.tb_model, 0.3), [48, 37], [-0.37, 0.11], [[0], [1], [0, 1]]
    elif case == "km":
        m, mesh, start, occs = hp.kane_mele(tb.tb_model, "odd"), [40, 23], [0.013, -0.21], [[0, 1], [2, 3], [0, 1, 2, 3]]
    elif case == "chain3x":
        m, mesh, start, occs = hp.chain3(tb.tb_model, -1.0, 2.0, 0.1), [57], [0.05], [[0], [1, 2]]
    else:
        m, mesh, start, occs = hp.cubic16(tb.tb_model), [6, 5, 7], [0.1, 0.2, 0.3], [list(range(8)), list(range(16)), [2, 9, 4]]
    w = tb.wf_array(m, mesh)
    gaps = w.solve_on_grid(start)
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-11
    D = len(mesh)
    for occ in occs:
        if D >= 2:
            for dirs in ([0, 1], [1, 0]) + (([1, 2], [2, 0]) if D == 3 else ()):
                got = w.berry_flux(occ, dirs=list(dirs), individual_phases=True)
                ref = orc.berry_flux(owfs, D, occ, list(dirs), individual_phases=True, vectorised=True)
                assert got.shape == ref.shape
                assert np.max(np.abs(wrap(got - ref))) < TOL_P
                tot = w.berry_flux(occ, dirs=list(dirs))
                assert np.max(np.abs(tot - ref.sum(axis=(-2, -1)))) < 1e-9
        for d in range(D):
            for be in (False, True):
                got = w.berry_phase(occ, d if D > 1 else None, contin=False, berry_evals=be)
                ref = orc.berry_phase(owfs, D, occ, d if D > 1 else None, contin=False, berry_evals=be)
                assert np.shape(got) == np.shape(ref)
                if be:
                    assert_phase_sets_close(got, ref, 1e-9)
                else:
                    assert np.max(np.abs(wrap(got - ref))) < TOL_P


def test_signed_band_indices_like_numpy(tb):
    """occ is a NumPy fancy index in the reference (pythtb.py:2981, :2989-2996, :3141): negative entries count from the top
    band, an entry outside [-nsta, nsta) is IndexError -- in berry_phase, berry_flux and the fused extension alike."""
    from oracle import tb_oracle as orc
    rng = np.random.default_rng(11)
    for m, mesh, start in ((hp.haldane(tb.tb_model, 0.0), [33, 33], [-0.5, -0.5]),
                           (hp.kane_mele(tb.tb_model, "odd"), [21, 17], [0.13, 0.21]),   # (no mesh line through a time-reversal invariant
                           # coordinate: on k_0 = 0 a single band's links across k_1 = 0, 1/2 join Kramers partners -- overlap 0, no phase)
                           (hp.cubic16(tb.tb_model), [9, 9, 9], [0.0, 0.0, 0.0])):
        n, D = m._nsta, len(mesh)
        w = tb.wf_array(m, mesh)
        w.solve_on_grid(start)
        owfs, _ = orc.solve_on_grid(m, mesh, start, vectorised=True)
        occs = [[-1], [-2, -1], range(-2, 0), [0, -1]] + [list(rng.choice(np.arange(-n, n), size=int(rng.integers(1, min(n, 4) + 1)),
                                                                       replace=False)) for _ in range(4)]
        for occ in occs:
            pos = [int(o) % n for o in occ]
            if len(set(pos)) != len(pos):
                continue                                   # a band twice: a singular overlap, nothing to compare
            for d in range(D):
                got = w.berry_phase(occ, d, contin=False)
                ref = orc.berry_phase(owfs, D, list(occ), d, contin=False)
                assert np.max(np.abs(wrap(got - ref))) < TOL_P
                assert np.array_equal(got, w.berry_phase(pos, d, contin=False))
            gotp = w.berry_flux(occ, individual_phases=True)
            refp = orc.berry_flux(owfs, D, list(occ), [0, 1], individual_phases=True, vectorised=True)
            assert np.max(np.abs(wrap(gotp - refp))) < TOL_P
            assert np.max(np.abs(w.berry_flux(occ) - refp.sum(axis=(-2, -1)))) < 1e-9
            assert np.array_equal(gotp, w.berry_flux(pos, individual_phases=True))
        for bad in ([n], [-n - 1], [0, n + 3]):
            with pytest.raises(IndexError):
                w.berry_phase(bad, 0, contin=False)
            with pytest.raises(IndexError):
                w.berry_flux(bad)
        if D == 2:
            w2 = tb.wf_array(m, mesh)
            g, f = w2.solve_on_grid_flux(start, occ=[-n])       # the bottom band by its negative index
            assert abs(f - w.berry_flux([0])) < 1e-9
            with pytest.raises(IndexError):
                w2.solve_on_grid_flux(start, occ=[n])


@pytest.mark.parametrize("n,nspin,mesh", [(5, 1, [40, 23]), (3, 2, [19, 70]), (7, 1, [9, 8, 35]), (6, 1, [5, 66, 4])])
def test_narrow_states_berry_calls_on_the_lds_tile_kernels(tb, n, nspin, mesh):
    """1..4 bands of states with 5..7 components (round 6): berry_phase in its determinant form and berry_flux take the LDS-tile
    kernels of tbk_berry_lanes.inl (products of link determinants / every link's determinant + k_flux_from_dets) -- against the
    oracle on the oracle's own eigenvectors, every direction and plane, both forms of the tile (strings across / along the lanes:
    the meshes put the string axis first, last and in the middle), and against the kernels they replace (TBK_WILSON_REG=1)."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    m = hp.random_model(tb.tb_model, n, len(mesh), nspin, seed=200 + n, nhop=4 * n * nspin, rmax=1)
    D = len(mesh)
    start = [0.07, -0.31, 0.2][:D]
    w = tb.wf_array(m, mesh)
    w.solve_on_grid(start)
    owfs, _ = orc.solve_on_grid(m, mesh, start, vectorised=True)
    for occ in ([0], [1, 0], [0, 2, 3], [1, 2, 3, 4]):
        for d in range(D):
            got = np.asarray(w.berry_phase(occ, d, contin=False))
            ref = np.asarray(orc.berry_phase(owfs, D, occ, d, contin=False))
            assert got.shape == ref.shape and np.max(np.abs(wrap(got - ref))) < TOL_P, (occ, d)
            with _lib.knob("TBK_WILSON_REG", 1):
                old = np.asarray(w.berry_phase(occ, d, contin=False))
            assert np.max(np.abs(wrap(got - old))) < 1e-11, (occ, d)
        for dirs in ([0, 1], [1, 0]) + (([1, 2], [2, 0]) if D == 3 else ()):
            got = w.berry_flux(occ, dirs=list(dirs), individual_phases=True)
            ref = orc.berry_flux(owfs, D, occ, list(dirs), individual_phases=True, vectorised=True)
            assert got.shape == ref.shape and np.max(np.abs(wrap(got - ref))) < TOL_P, (occ, dirs)
            tot = w.berry_flux(occ, dirs=list(dirs))
            assert np.max(np.abs(tot - got.sum(axis=(-2, -1)))) < 1e-9
            with _lib.knob("TBK_WILSON_REG", 1):
                old = w.berry_flux(occ, dirs=list(dirs), individual_phases=True)
            assert np.max(np.abs(wrap(got - old))) < 1e-11, (occ, dirs)


def test_upload_roundtrip_and_user_written_arrays(tb):
    """A wf_array filled on the host (oracle eigenvectors, different gauge) gives the
    same gauge-invariant numbers; download(upload(x)) is the identity."""
    from oracle import tb_oracle as orc
    m = hp.haldane(tb.tb_model, 0.1)
    owfs, _ = orc.solve_on_grid(m, [21, 17], [-0.5, -0.5], vectorised=True)
    w = tb.wf_array(m, [21, 17])
    w._wfs = owfs
    f = w.berry_flux([0], individual_phases=True)
    assert np.array_equal(w._wfs, owfs)
    assert np.max(np.abs(f - orc.berry_flux(owfs, 2, [0], individual_phases=True, vectorised=True))) < TOL_P
    w[3, 4] = w[3, 4] * np.exp(0.7j)                          # a gauge change leaves fluxes alone
    assert np.max(np.abs(w.berry_flux([0], individual_phases=True) - f)) < TOL_P


# ------------------------------------------------------------------ BASELINE.json config sizes
def test_config_B_haldane_1024_eigenvalues(tb):
    """configs[1]: Haldane (delta=0.2) solve_all on k_uniform_mesh([1024,1024])."""
    from oracle import tb_oracle as orc
    m = hp.haldane(tb.tb_model, 0.2)
    k = m.k_uniform_mesh([1024, 1024])
    ev = m.solve_all(k)
    assert ev.shape == (2, 1024 * 1024)
    assert np.all(ev[0] <= ev[1])                             # sorted per k
    assert abs(ev.sum()) < 1e-6                               # traceless up to 2*delta*0: sum E = tr H = 0
    assert abs(ev.min() + ev.max()) < 1e-9                    # min -3.0x, max 3.0x symmetric
    sel = np.random.default_rng(0).integers(0, len(k), 4096)
    assert np.max(np.abs(ev[:, sel] - orc.solve_all_vec(m, k[sel]))) < TOL_E
    ev2, vec = m.solve_all(k[:65536], eig_vectors=True)
    assert np.max(np.abs(np.einsum("bko,bko->bk", vec.conj(), vec).real - 1.0)) < 1e-13
    # the REFERENCE's own numbers for this recipe on the 256^2 sub-mesh (make_golden.py --full: per-band sum / min / max);
    # every 4th point of the 1024^2 mesh per axis IS that sub-mesh (i/256 = 4i/1024 exactly)
    g = load_golden("full_size")
    sub = ev.reshape(2, 1024, 1024)[:, ::4, ::4].reshape(2, -1)
    assert np.max(np.abs(sub.sum(axis=1) - g["B256_sum"])) < 1e-9
    assert np.max(np.abs(sub.min(axis=1) - g["B256_min"])) < 1e-12 and np.max(np.abs(sub.max(axis=1) - g["B256_max"])) < 1e-12
    ev256 = m.solve_all(m.k_uniform_mesh([256, 256]))
    assert np.max(np.abs(ev256.sum(axis=1) - g["B256_sum"])) < 1e-9
    assert np.max(np.abs(ev256.min(axis=1) - g["B256_min"])) < 1e-12 and np.max(np.abs(ev256.max(axis=1) - g["B256_max"])) < 1e-12


def test_config_C_haldane_2048_chern(tb):
    """configs[2]: Chern number of the lower Haldane band on the 2048x2048 mesh is exactly -1."""
    m = hp.haldane(tb.tb_model, 0.0)
    w = tb.wf_array(m, [2049, 2049])
    gaps = w.solve_on_grid([-0.5, -0.5])
    flux = w.berry_flux([0])
    assert abs(flux / (2 * np.pi) + 1.0) < 1e-10
    assert round(flux / (2 * np.pi)) == -1
    assert abs(gaps[0] - 1.55884812) < 1e-6                   # BASELINE.md section 2 (reference run)
    plaq = w.berry_flux([0], individual_phases=True)
    assert plaq.shape == (2048, 2048)
    assert abs(plaq.sum() - flux) < 1e-9                      # checksum of checksums
    assert abs(w.berry_flux([0, 1])) < 1e-9                   # both bands together: zero flux
    full = os.path.join(GOLDEN, "full_size.npz")
    if os.path.exists(full):
        g = load_golden("full_size")
        assert abs(flux - float(g["C_flux"])) < 1e-9
        assert np.max(np.abs(gaps - g["C_min_gaps"])) < 1e-12
        assert np.max(np.abs(plaq.sum(axis=1) - g["C_flux_row_sums"])) < 1e-10


def test_config_D_kane_mele_wilson_loop(tb):
    """configs[3] on one GPU: Kane-Mele (odd) 4096x512 mesh, Wilson-loop eigenphases."""
    m = hp.kane_mele(tb.tb_model, "odd")
    w = tb.wf_array(m, [4097, 513])
    gaps = w.solve_on_grid([-0.5, -0.5])
    assert gaps[0] < 1e-10 and gaps[2] < 1e-10 and abs(gaps[1] - 0.86770552) < 1e-6   # Kramers pairs on the mesh
    wc = w.berry_phase([0, 1], dir=0, contin=False, berry_evals=True)
    assert wc.shape == (513, 2)
    # time reversal: the centres at k_y and -k_y coincide as sets; Z2 odd: they wind
    assert_phase_sets_close(wc[1:256], wc[-2:-257:-1], 1e-8)
    tot = w.berry_phase([0, 1], dir=0, contin=False)
    assert np.max(np.abs(wrap(wc.sum(axis=1) - tot))) < 1e-9   # sum of eigenphases = phase of det
    full = os.path.join(GOLDEN, "full_size.npz")
    if os.path.exists(full):
        g = load_golden("full_size")
        assert np.max(np.abs(gaps - g["D_min_gaps"])) < 1e-11
        assert_phase_sets_close(wc, g["D_wan_cent"], 1e-8)
    # the Z2 integer BASELINE configs[3] names: the centres wind an odd number of times over half the zone
    assert tb.z2_from_wilson_centres(wc) == 1 and tb.z2_from_wilson_centres(wc, half="lower") == 1


@pytest.mark.parametrize("phase,z2", [("odd", 1), ("even", 0)])
def test_kane_mele_z2_index(tb, phase, z2):
    """Wilson-loop Z2 of both Kane-Mele phases (examples/kane_mele.py:27-34) on the 41 x 41 array of the reference's
    example, along either direction, from the device's eigenphases."""
    w = tb.wf_array(hp.kane_mele(tb.tb_model, phase), [41, 41])
    w.solve_on_grid([-0.5, -0.5])
    for d in (0, 1):
        wc = w.berry_phase([0, 1], dir=d, contin=False, berry_evals=True)
        assert tb.z2_from_wilson_centres(wc) == z2 and tb.z2_from_wilson_centres(wc, half="lower") == z2
    if phase == "even":
        g = load_golden("grid_km_even_41")
        assert tb.z2_from_wilson_centres(w.berry_phase([0, 1], dir=1, contin=False, berry_evals=True)) == 0 and g is not None


# ------------------------------------------------------------------ edge cases and plumbing
def test_edge_cases(tb):
    m = hp.haldane(tb.tb_model, 0.1)
    ev = m.solve_all([])                                       # empty list
    assert ev.shape == (2, 0)
    ev, vec = m.solve_all(np.zeros((0, 2)), eig_vectors=True)
    assert ev.shape == (2, 0) and vec.shape == (2, 0, 2)
    one = m.solve_all([[0.25, 0.75]])                          # single k, ragged python list of lists
    assert one.shape == (2, 1)
    big = m.solve_all(np.tile([[0.25, 0.75]], (1000, 1)))      # identical k -> identical columns
    assert np.all(big == one)
    # smallest legal meshes, 1-state models (solve_on_grid returns None: no gaps)
    w = tb.wf_array(m, [2, 2])
    w.solve_on_grid([0.0, 0.0])
    assert w.berry_flux([0], individual_phases=True).shape == (1, 1)
    s = hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[0.3]])
    s.set_onsite([0.7])
    s.set_hop(0.5 + 0.2j, 0, 0, [1])
    ws = tb.wf_array(s, [17])
    assert ws.solve_on_grid([0.0]) is None
    assert abs(abs(ws[3][0, 0]) - 1.0) < 1e-14
    ph = ws.berry_phase([0])
    assert abs(((ph - 2 * np.pi * 0.3 + np.pi) % (2 * np.pi)) - np.pi) < 1e-12    # single band: 2 pi tau
    ev = s.solve_all([0.0, 0.25, 0.5])
    assert np.allclose(ev[0], 0.7 + 2 * np.real((0.5 + 0.2j) * np.exp(2j * np.pi * np.array([0.0, 0.25, 0.5]))), atol=1e-14)
    # long-range hoppings (|R| up to 5) take the dynamic-pmax kernels
    lr = hp.random_model(tb.tb_model, 2, 2, 1, 21, nhop=12, rmax=5)
    from oracle import tb_oracle as orc
    wl = tb.wf_array(lr, [19, 23])
    gaps = wl.solve_on_grid([0.1, -0.3])
    owfs, ogaps = orc.solve_on_grid(lr, [19, 23], [0.1, -0.3], vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-11
    ref = orc.berry_flux(owfs, 2, [0], individual_phases=True, vectorised=True)
    assert np.max(np.abs(wrap(wl.berry_flux([0], individual_phases=True) - ref))) < TOL_P
    # non-contiguous, repeated-order occupation lists; transposed planes negate the flux
    w4 = tb.wf_array(hp.kane_mele(tb.tb_model, "even"), [12, 9])
    w4.solve_on_grid([0.05, 0.07])
    a = w4.berry_flux([3, 0], individual_phases=True)
    b = w4.berry_flux([0, 3], individual_phases=True)
    assert np.max(np.abs(a - b)) < 1e-12
    assert np.max(np.abs(w4.berry_flux([0, 1], dirs=[1, 0], individual_phases=True)
                         + w4.berry_flux([0, 1], dirs=[0, 1], individual_phases=True).T)) < 1e-12


def test_large_nocc_and_unsupported_sizes(tb):
    from pythtb_amd import _lib
    m = hp.random_model(tb.tb_model, 20, 2, 1, 33)
    w = tb.wf_array(m, [6, 5])
    w.solve_on_grid([0.0, 0.0])
    from oracle import tb_oracle as orc
    owfs, _ = orc.solve_on_grid(m, [6, 5], [0.0, 0.0], vectorised=True)
    occ = list(range(12))
    assert np.max(np.abs(wrap(w.berry_flux(occ, individual_phases=True)
                              - orc.berry_flux(owfs, 2, occ, individual_phases=True, vectorised=True)))) < TOL_P
    got = w.berry_phase(occ, 0, contin=False, berry_evals=True)
    assert_phase_sets_close(got, orc.berry_phase(owfs, 2, occ, 0, contin=False, berry_evals=True), 1e-9)
    # more than TBK_MAX_NOCC bands: link determinants by LU (flux and det-type Berry phases of any size) ...
    for occ in (list(range(17)), list(range(20)), [19, 0, 7] + list(range(1, 7)) + list(range(8, 19))):
        assert np.max(np.abs(wrap(w.berry_flux(occ, individual_phases=True)
                                  - orc.berry_flux(owfs, 2, occ, individual_phases=True, vectorised=True)))) < TOL_P
        assert abs(w.berry_flux(occ) - orc.berry_flux(owfs, 2, occ, vectorised=True)) < 1e-9
        for d in (0, 1):
            got = w.berry_phase(occ, d, contin=False)
            assert np.max(np.abs(wrap(got - orc.berry_phase(owfs, 2, occ, d, contin=False)))) < TOL_P
        for d in (0, 1):                                      # ... and the Wilson-loop eigenphases too
            got = w.berry_phase(occ, d, contin=False, berry_evals=True)
            assert_phase_sets_close(got, orc.berry_phase(owfs, 2, occ, d, contin=False, berry_evals=True), 1e-9)
    # a ribbon with 70 occupied bands of 140 (the LU matrix no longer fits in LDS) and a 1-D string
    rib = hp.haldane(tb.tb_model, 1.2).cut_piece(70, 1)        # trivial phase: the ribbon is gapped at half filling
    wr = tb.wf_array(rib, [9])
    wr.solve_on_grid([0.0])
    orib, _ = orc.solve_on_grid(rib, [9], [0.0], vectorised=True)
    for occ in (list(range(70)), list(range(140))):
        assert abs(wrap(wr.berry_phase(occ, contin=False) - orc.berry_phase(orib, 1, occ, None, contin=False))) < 1e-8
    got = wr.berry_phase(list(range(70)), contin=False, berry_evals=True)          # 70 hybrid Wannier centres of the ribbon
    assert_phase_sets_close(got, orc.berry_phase(orib, 1, list(range(70)), None, contin=False, berry_evals=True), 1e-8)
    # the Cayley transform behind these eigenphases has a pole at theta = alpha + pi: put it exactly on one of
    # the eigenphases (output = -theta) and the call must notice and redo that string with another alpha
    ctx = _lib.default_context()
    ctx.prof_enable(1)
    ctx.prof_reset()
    try:
        with _lib.knob("TBK_WILSON_ALPHA", repr(float(-got[35] - np.pi))):
            again = wr.berry_phase(list(range(70)), contin=False, berry_evals=True)
        launches = ctx.prof_report()["wilson_cayley"]["launches"]
    finally:
        ctx.prof_enable(0)
    assert launches == 2                                       # first alpha rejected, second accepted
    assert_phase_sets_close(again, got, 1e-11)
    big = hp.quiet(tb.tb_model, 1, 1, [[1.0]], 2100)
    with pytest.raises(_lib.TbkError, match="limit"):
        big.solve_all([0.1])                                   # nsta > TBK_MAX_NSTA fails loudly


@pytest.mark.parametrize("norb", [1, 2, 3, 4])
def test_mesh_kernels_long_range_hops_and_fallback(tb, norb):
    """n <= 4 meshes: hops reaching three and five cells along the last axis take the run-time-degree form of the
    row-polynomial kernel (k_grid_rows<N,-1>); TBK_GRID_KERNEL=1 selects the term-walking kernel that serves
    models whose row table would not fit in LDS.  Both against the oracle, and against each other."""
    from oracle import tb_oracle as orc
    rng = np.random.default_rng(100 + norb)
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((norb, 2)))
    m.set_onsite([-8.0] + list(rng.standard_normal(norb - 1)))          # band 0 stays isolated
    for (i, j, R) in [(0, norb - 1, [0, 3]), (0, 0, [1, -5]), (norb - 1, 0, [2, 1]), (0, norb - 1, [1, 0])] + \
                     [(i, i + 1, [0, 1]) for i in range(norb - 1)]:
        m.set_hop(0.4 * complex(rng.standard_normal(), rng.standard_normal()), i, j, R, mode="add")
    mesh, start = [37, 70], [0.13, -0.41]
    w = tb.wf_array(m, mesh)
    gaps = w.solve_on_grid(start)
    host = w._wfs.copy()
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    if norb > 1:
        assert np.max(np.abs(gaps - ogaps)) < 1e-11
    ref_e = orc.solve_all_vec(m, orc.mesh_points(mesh, start) if hasattr(orc, "mesh_points") else np.array(
        [[start[0] + a / (mesh[0] - 1), start[1] + b / (mesh[1] - 1)] for a in range(mesh[0]) for b in range(mesh[1])]))
    ham = orc.ham_batch(m, np.array([[start[0] + a / (mesh[0] - 1), start[1] + b / (mesh[1] - 1)] for a in (0, 17, 35) for b in (0, 33, 68)]))
    V = host.reshape(mesh[0], mesh[1], norb, norb)
    for idx, (a, b) in enumerate([(a, b) for a in (0, 17, 35) for b in (0, 33, 68)]):
        vec = V[a, b]                                           # [band][orb]
        e = np.real(np.einsum('bi,ij,bj->b', vec.conj(), ham[idx], vec))
        assert np.max(np.abs(ham[idx] @ vec.T - vec.T * e)) < 1e-11
        assert np.max(np.abs(e - ref_e[:, a * mesh[1] + b])) < 1e-11
    occ = [0]
    assert norb == 1 or ogaps[0] > 1.0
    flux = w.berry_flux(occ, individual_phases=True)
    assert np.max(np.abs(wrap(flux - orc.berry_flux(owfs, 2, occ, individual_phases=True, vectorised=True)))) < 1e-9
    with tb._lib.knob("TBK_GRID_KERNEL", 1):
        w2 = tb.wf_array(m, mesh)
        gaps2 = w2.solve_on_grid(start)
        flux2 = w2.berry_flux(occ, individual_phases=True)
    if norb > 1:
        assert np.max(np.abs(gaps2 - gaps)) < 1e-12
    assert np.max(np.abs(wrap(flux2 - flux))) < 1e-9


def test_wilson_pipeline_batches_and_axes(tb):
    """Workgroup-level Wilson-loop eigenphases (3 or more bands): strings along every axis of a 3-D mesh, band
    lists in arbitrary order, and the string batching (forced to a few strings per batch) must not change a bit."""
    from oracle import tb_oracle as orc
    m = hp.random_model(tb.tb_model, 7, 3, 1, 21, nhop=30, rmax=1)
    mesh = [5, 6, 7]
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.05, -0.1, 0.2])
    owfs, ogaps = orc.solve_on_grid(m, mesh, [0.05, -0.1, 0.2], vectorised=True)
    for occ in ([0, 1, 2], [4, 2, 3, 0], list(range(7))):
        for d in range(3):
            got = w.berry_phase(occ, d, contin=False, berry_evals=True)
            ref = orc.berry_phase(owfs, 3, occ, d, contin=False, berry_evals=True)
            assert got.shape == ref.shape
            assert_phase_sets_close(got, ref, 1e-9)
            with tb._lib.knob("TBK_WILSON_BATCH_BYTES", 2 * (mesh[d] - 1) * len(occ) ** 2 * 16 * 4):   # four strings per batch
                again = w.berry_phase(occ, d, contin=False, berry_evals=True)
            assert np.array_equal(again, got)
    # the reference's continuity post-processing on top of the pipeline's raw eigenphases (2-D mesh, smooth model)
    h3 = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.0, 1.0]], [[0.0, 0.0], [0.3, 0.1], [0.6, 0.5], [0.1, 0.7], [0.8, 0.2]])
    h3.set_onsite([-3.0, -2.6, -2.2, 2.5, 3.0])
    rng = np.random.default_rng(8)
    for i in range(5):
        for j in range(i + 1, 5):
            for R in ([0, 0], [1, 0], [0, 1]):
                h3.set_hop(0.3 * complex(rng.standard_normal(), rng.standard_normal()), i, j, R)
    w3 = tb.wf_array(h3, [21, 25])
    w3.solve_on_grid([0.0, 0.0])
    o3, g3 = orc.solve_on_grid(h3, [21, 25], [0.0, 0.0], vectorised=True)
    assert g3[2] > 2.0
    for d in (0, 1):
        got = w3.berry_phase([0, 1, 2], d, contin=True, berry_evals=True)
        ref = orc.berry_phase(o3, 2, [0, 1, 2], d, contin=True, berry_evals=True)
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) < 1e-8
        assert np.max(np.abs(w3.berry_phase([0, 1, 2], d, contin=True) - orc.berry_phase(o3, 2, [0, 1, 2], d, contin=True))) < 1e-8
    # nine or more bands: link determinants by one workgroup per link, on slices / strings of a 3-D mesh
    m2 = hp.random_model(tb.tb_model, 12, 3, 1, 5, nhop=60, rmax=1)
    mesh2 = [4, 5, 6]
    w2 = tb.wf_array(m2, mesh2)
    w2.solve_on_grid([0.0, 0.1, -0.2])
    o2, _ = orc.solve_on_grid(m2, mesh2, [0.0, 0.1, -0.2], vectorised=True)
    for occ in (list(range(9)), [11, 3] + list(range(4, 11)) + [0, 1, 2]):
        for dirs in ([0, 2], [1, 2], [2, 0]):
            got = w2.berry_flux(occ, dirs=dirs, individual_phases=True)
            ref = orc.berry_flux(o2, 3, occ, dirs, individual_phases=True, vectorised=True)
            assert got.shape == ref.shape and np.max(np.abs(wrap(got - ref))) < 1e-9
            assert np.max(np.abs(w2.berry_flux(occ, dirs=dirs) - orc.berry_flux(o2, 3, occ, dirs, vectorised=True))) < 1e-8
        for d in range(3):
            assert np.max(np.abs(wrap(w2.berry_phase(occ, d, contin=False) - orc.berry_phase(o2, 3, occ, d, contin=False)))) < 1e-9


def test_wide_models_workgroup_kernel(tb):
    """65 <= nsta <= 256 (ribbons, slabs): one workgroup per matrix, matrices in global memory."""
    from oracle import tb_oracle as orc
    m = hp.random_model(tb.tb_model, 45, 1, 2, 5, nhop=150, rmax=2)        # 90 states, spinor, 1-D
    k = np.linspace(-0.5, 0.5, 23)
    ev, vec = m.solve_all(k, eig_vectors=True)
    ref = orc.solve_all_vec(m, k.reshape(-1, 1))
    scale = np.abs(ref).max()
    assert np.max(np.abs(ev - ref)) < 1e-12 * scale
    ham = orc.ham_batch(m, k.reshape(-1, 1))
    V = vec.reshape(90, len(k), 90)
    for ik in range(len(k)):
        assert np.max(np.abs(V[:, ik].conj() @ V[:, ik].T - np.identity(90))) < 1e-12
        assert np.max(np.abs(ham[ik] @ V[:, ik].T - V[:, ik].T * ev[:, ik])) < 1e-11 * scale
    assert np.max(np.abs(m.solve_all(k) - ref)) < 1e-12 * scale              # eigenvalue-only variant
    with tb._lib.knob("TBK_BIG_BATCH", 5):                                  # whole-chip solver in several batches: same bits
        assert np.array_equal(m.solve_all(k), m.solve_all(k)) and np.array_equal(m.solve_all(k[:7]), m.solve_all(k)[:, :7])
        # (eigenvalue-only solves of 65..1024 states take tridiagonalise + bisection by default; the Jacobi kernels' own
        # eigenvalue-only variants -- still the route for TBK_TRIG=0, n > 1024 and a few very large matrices -- keep their
        # batch-splitting invariant under test)
        with tb._lib.knob("TBK_TRIG", 0):
            evj = m.solve_all(k)
            assert np.max(np.abs(evj - ref)) < 1e-12 * scale
            assert np.array_equal(evj, m.solve_all(k)) and np.array_equal(m.solve_all(k[:7]), evj[:, :7])
    w = tb.wf_array(m, [14])
    gaps = w.solve_on_grid([0.0])
    owfs, ogaps = orc.solve_on_grid(m, [14], [0.0], vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-10 * scale
    for occ in (list(range(8)), [3], list(range(20, 36))):
        got = w.berry_phase(occ)
        assert abs(wrap(got - orc.berry_phase(owfs, 1, occ))) < 1e-8         # random model: small gaps
    assert np.max(np.abs(w._wfs[-1] - w._wfs[0] * np.repeat(np.exp(-2j * np.pi * m._orb[:, 0]), 2).reshape(45, 2))) < 1e-13
    m2 = hp.random_model(tb.tb_model, 130, 2, 1, 8, nhop=500, rmax=1)        # 130 states, 2-D mesh
    w2 = tb.wf_array(m2, [4, 5])
    g2 = w2.solve_on_grid([0.1, 0.2])
    _, og2 = orc.solve_on_grid(m2, [4, 5], [0.1, 0.2], vectorised=True)
    assert np.max(np.abs(g2 - og2)) < 1e-10 * np.abs(og2).max() + 1e-11
    e0 = m2.solve_one([0.1, 0.2])
    assert np.max(np.abs(e0 - orc.solve_all_vec(m2, [[0.1, 0.2]])[:, 0])) < 1e-11 * np.abs(e0).max()


def test_wide_batches_block_jacobi(tb):
    """Batches of wide matrices (nk n^2 >= 1.4e6, n >= 96): block Jacobi with 16x16 sub-solves; n = 100 is padded
    to 112 with decoupled diagonal entries that must never surface."""
    from oracle import tb_oracle as orc
    rib = hp.haldane(tb.tb_model, 1.2).cut_piece(50, 1)                     # 100 bands, gapped at half filling
    k = np.linspace(0.0, 1.0, 160, endpoint=False) + 0.013
    ham = orc.ham_batch(rib, k.reshape(-1, 1))
    ref = np.linalg.eigvalsh(ham).T
    scale = np.abs(ref).max()
    ev, vec = rib.solve_all(k, eig_vectors=True)
    assert np.max(np.abs(ev - ref)) < 2e-12 * scale
    assert np.max(np.abs(rib.solve_all(k) - ref)) < 2e-12 * scale            # eigenvalue-only variant
    for ik in (0, 7, 80, 159):
        V = vec[:, ik, :]
        assert np.max(np.abs(V.conj() @ V.T - np.identity(100))) < 1e-11
        assert np.max(np.abs(ham[ik] @ V.T - V.T * ev[:, ik])) < 1e-11 * scale
    with tb._lib.knob("TBK_BLOCKED", 0):                                    # same batch through the other solvers
        ev0 = rib.solve_all(k)
    assert np.max(np.abs(ev0 - ev)) < 2e-12 * scale
    with tb._lib.knob("TBK_TRIG", 0):                                       # the block-Jacobi / whole-chip EIGENVALUE-ONLY kernels
        ev_blk = rib.solve_all(k)
        with tb._lib.knob("TBK_BLOCKED", 0):
            ev_big = rib.solve_all(k)
    assert np.max(np.abs(ev_blk - ref)) < 2e-12 * scale and np.max(np.abs(ev_big - ref)) < 2e-12 * scale
    w = tb.wf_array(rib, [161])                                            # mesh mode: gaps, images, Berry phase
    gaps = w.solve_on_grid([0.0])
    owfs, ogaps = orc.solve_on_grid(rib, [161], [0.0], vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-10 * scale
    occ = list(range(50))
    assert abs(wrap(w.berry_phase(occ) - orc.berry_phase(owfs, 1, occ))) < 1e-8
    m2 = hp.random_model(tb.tb_model, 64, 1, 2, 9, nhop=400, rmax=2)        # 128 states, spinor: no padding
    k2 = np.linspace(-0.5, 0.5, 90)
    ev2 = m2.solve_all(k2)
    ref2 = orc.solve_all_vec(m2, k2.reshape(-1, 1))
    assert np.max(np.abs(ev2 - ref2)) < 2e-12 * np.abs(ref2).max()
    with tb._lib.knob("TBK_BIG_BATCH", 37):                                 # several workspace batches: same bits
        ev_b, vec_b = rib.solve_all(k, eig_vectors=True)
    assert np.array_equal(ev_b, ev) and np.array_equal(vec_b, vec)
    # supplied matrices through the same solver, with the special cases a Jacobi method can trip over
    from pythtb_amd import _lib
    rng = np.random.default_rng(4)
    n, nk = 99, 150                                                        # padded to 112
    a = rng.normal(size=(nk, n, n)) + 1j * rng.normal(size=(nk, n, n))
    hmat = np.ascontiguousarray(a + a.conj().transpose(0, 2, 1))
    hmat[3] = np.diag(np.arange(n, dtype=float)[::-1])                      # diagonal, descending
    hmat[4] = np.ones((n, n))                                               # rank one: n-1 degenerate eigenvalues
    hmat[5] = 0.0
    hmat[6] = np.identity(n)
    hmat[7] = np.kron(np.identity(n // 3), np.array([[0, 1j, 0], [-1j, 0, 2], [0, 2, 0]]))   # 33 identical blocks
    e = np.zeros((n, nk))
    v = np.zeros((n, nk, n), dtype=complex)
    _lib.check(_lib.lib.tbk_eigh_batch(_lib.default_context().handle, n, _lib.dptr(hmat), nk, _lib.dptr(e), _lib.dptr(v)))
    for i in (0, 3, 4, 5, 6, 7, 149):
        want = np.linalg.eigvalsh(hmat[i])
        sc = max(1.0, np.abs(want).max())
        assert np.max(np.abs(e[:, i] - want)) < 2e-12 * sc
        assert np.max(np.abs(hmat[i] @ v[:, i].T - v[:, i].T * e[:, i])) < 1e-11 * sc
        assert np.max(np.abs(v[:, i].conj() @ v[:, i].T - np.identity(n))) < 1e-11


def test_very_wide_models_whole_chip_jacobi(tb):
    """nsta > 256 (flakes, thick slabs): one kernel launch per Jacobi round over all 2x2 blocks."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    import ctypes as C
    m = hp.random_model(tb.tb_model, 301, 1, 1, 11, nhop=1200, rmax=1)      # odd n: the bye player
    k = np.array([-0.37, 0.0, 0.21])
    ev, vec = m.solve_all(k, eig_vectors=True)
    ref = orc.solve_all_vec(m, k.reshape(-1, 1))
    scale = np.abs(ref).max()
    assert np.max(np.abs(ev - ref)) < 1e-12 * scale
    ham = orc.ham_batch(m, k.reshape(-1, 1))
    for ik in range(len(k)):
        V = vec[:, ik, :]
        assert np.max(np.abs(V.conj() @ V.T - np.identity(301))) < 1e-12
        assert np.max(np.abs(ham[ik] @ V.T - V.T * ev[:, ik])) < 1e-11 * scale
    assert np.max(np.abs(m.solve_all(k) - ref)) < 1e-12 * scale
    # mesh mode: periodic image, gaps, Berry phase of a band group
    m2 = hp.random_model(tb.tb_model, 140, 1, 2, 3, nhop=500, rmax=1)       # 280 spinor states
    w = tb.wf_array(m2, [5])
    gaps = w.solve_on_grid([0.0])
    owfs, ogaps = orc.solve_on_grid(m2, [5], [0.0], vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-10 * np.abs(ogaps).max() + 1e-11
    assert np.max(np.abs(w._wfs[-1] - w._wfs[0] * np.repeat(np.exp(-2j * np.pi * m2._orb[:, 0]), 2).reshape(140, 2))) < 1e-13
    # supplied matrices, even n, through the C ABI
    rng = np.random.default_rng(5)
    n = 512
    a = rng.normal(size=(2, n, n)) + 1j * rng.normal(size=(2, n, n))
    hmat = np.ascontiguousarray(a + a.conj().transpose(0, 2, 1))
    e = np.zeros((n, 2))
    v = np.zeros((n, 2, n), dtype=complex)
    _lib.check(_lib.lib.tbk_eigh_batch(_lib.default_context().handle, n, _lib.dptr(hmat), 2, _lib.dptr(e), _lib.dptr(v)))
    for i in range(2):
        want = np.linalg.eigvalsh(hmat[i])
        assert np.max(np.abs(e[:, i] - want)) < 1e-12 * np.abs(want).max()
        assert np.max(np.abs(hmat[i] @ v[:, i].T - v[:, i].T * e[:, i])) < 1e-10 * np.abs(want).max()


def test_row16_register_solver(tb):
    """n = 15, 16 on k lists / supplied matrices: one DPP row of 16 lanes per matrix (15 exercises the padding row)."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    for norb, nspin, seed in ((16, 1, 3), (15, 1, 4), (8, 2, 5)):
        m = hp.random_model(tb.tb_model, norb, 2, nspin, seed, nhop=90, rmax=1)
        n = norb * nspin
        k = np.random.default_rng(seed).uniform(-0.5, 0.5, size=(37, 2))
        ev, vec = m.solve_all(k, eig_vectors=True)
        ref = orc.solve_all_vec(m, k)
        scale = np.abs(ref).max()
        assert np.max(np.abs(ev - ref)) < 1e-12 * scale
        assert np.max(np.abs(m.solve_all(k) - ref)) < 1e-12 * scale
        ham = orc.ham_batch(m, k)
        V = vec.reshape(n, len(k), n)
        for ik in range(len(k)):
            assert np.max(np.abs(V[:, ik].conj() @ V[:, ik].T - np.identity(n))) < 1e-12
            assert np.max(np.abs(ham[ik] @ V[:, ik].T - V[:, ik].T * ev[:, ik])) < 1e-11 * scale
    rng = np.random.default_rng(9)
    for n in (15, 16):
        a = rng.normal(size=(70, n, n)) + 1j * rng.normal(size=(70, n, n))
        hmat = np.ascontiguousarray(a + a.conj().transpose(0, 2, 1))
        hmat[3] = np.diag(np.arange(n, dtype=float))                 # already diagonal
        hmat[4] = np.ones((n, n))                                     # rank one: n-1 degenerate eigenvalues
        hmat[5] = 0.0
        e = np.zeros((n, 70))
        v = np.zeros((n, 70, n), dtype=complex)
        _lib.check(_lib.lib.tbk_eigh_batch(_lib.default_context().handle, n, _lib.dptr(hmat), 70, _lib.dptr(e), _lib.dptr(v)))
        for i in range(70):
            want = np.linalg.eigvalsh(hmat[i])
            sc = max(1.0, np.abs(want).max())
            assert np.max(np.abs(e[:, i] - want)) < 1e-12 * sc
            assert np.max(np.abs(hmat[i] @ v[:, i].T - v[:, i].T * e[:, i])) < 1e-11 * sc
            assert np.max(np.abs(v[:, i].conj() @ v[:, i].T - np.identity(n))) < 1e-12


def test_rccl_single_rank_allgather(tb):
    """tbk_comm_* with a 1-rank communicator (the N>1 path's collective, as far as one GPU goes)."""
    import ctypes as C
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    uid = (C.c_ubyte * 128)()
    _lib.check(lib.tbk_comm_unique_id(uid))
    _lib.check(lib.tbk_comm_init(ctx.handle, uid, 1, 0))
    try:
        src = np.array([1.5, -2.0, 3.25])
        send, recv = C.c_void_p(), C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 24, C.byref(send)))
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 24, C.byref(recv)))
        _lib.check(lib.tbk_dev_upload(ctx.handle, send, src.ctypes.data_as(C.c_void_p), 24))
        _lib.check(lib.tbk_comm_allgather_f64(ctx.handle, send, recv, 3))
        out = np.zeros(3)
        _lib.check(lib.tbk_dev_download(ctx.handle, out.ctypes.data_as(C.c_void_p), recv, 24))
        assert np.array_equal(out, src)
        _lib.check(lib.tbk_dev_free(ctx.handle, send))
        _lib.check(lib.tbk_dev_free(ctx.handle, recv))
    finally:
        _lib.check(lib.tbk_comm_destroy(ctx.handle))


def test_sharded_windows_equal_unsharded(tb):
    """SURVEY 8e on one GPU: run every rank's window in turn (flux: slabs along axis 0 with a halo
    row; Berry strings along dir 0: windows along axis 1) and compare with the unsharded arrays."""
    from pythtb_amd import shard
    m = hp.kane_mele(tb.tb_model, "odd")
    mesh, start = [49, 37], [-0.5, -0.5]
    full = tb.wf_array(m, mesh)
    gaps = full.solve_on_grid(start)
    ref_flux = full.berry_flux([0, 1], individual_phases=True)
    ref_wl = full.berry_phase([0, 1], 0, contin=False, berry_evals=True)
    ref_bp = full.berry_phase([2, 3], 0, contin=False)
    host = full._wfs.copy()
    for world in (2, 3, 8):
        parts, gmins = [], []
        for r in range(world):                               # flux: slabs of plaquette rows
            row0, nrows = shard.split_rows(mesh[0], world, r)
            w = tb.wf_array(m, [nrows, mesh[1]])
            gmins.append(w.solve_on_grid_window(start, [row0, 0], mesh))
            assert np.array_equal(w._wfs, host[row0:row0 + nrows])          # bit-identical, halo row included
            p = w.berry_flux([0, 1], individual_phases=True)
            assert np.array_equal(p, ref_flux[row0:row0 + nrows - 1])
            parts.append(w.berry_flux([0, 1]))
        assert abs(sum(parts) - ref_flux.sum()) < 1e-11
        assert np.max(np.abs(np.min(gmins, axis=0) - gaps)) == 0.0
        wl, bp = [], []
        for r in range(world):                               # strings along dir 0: cut axis 1
            axis, b, e = shard.split_strings(mesh, 0, world, r)
            assert axis == 1
            if e - b < 2:                                    # a wf_array axis needs >= 2 points: widen by one
                b = max(0, min(b, mesh[1] - 2))
                e2 = b + 2
            else:
                e2 = e
            w = tb.wf_array(m, [mesh[0], e2 - b])
            w.solve_on_grid_window(start, [0, b], mesh)
            wl.append(w.berry_phase([0, 1], 0, contin=False, berry_evals=True)[:e - b])
            bp.append(w.berry_phase([2, 3], 0, contin=False)[:e - b])
        # Bit for bit, the string at the periodic-image column included: a window that holds the image column without column 0
        # forms it as the lanes of column 0 do in the full array -- column 0's vector under column 0's phases, then times
        # (image phase conj column-0 phase) (GridArgs::img_win; it was "to rounding" in round 5)
        wl, bp = np.concatenate(wl), np.concatenate(bp)
        assert np.array_equal(wl, ref_wl) and np.array_equal(bp, ref_bp)
        # ... and the eigenvectors themselves, for a window that starts past column 0 and for every column solved on its own
        w = tb.wf_array(m, [mesh[0], mesh[1] - 5])
        w.solve_on_grid_window(start, [0, 5], mesh)
        assert np.array_equal(w._wfs, host[:, 5:])


@pytest.mark.parametrize("half,mesh", [(3, [23, 53]), (6, [41, 53]), (7, [41, 53]), (8, [41, 53]), (12, [41, 53])])
def test_warm_started_solver_keeps_images_and_windows_bit_identical(tb, half, mesh):
    """The n = 9..21 wavefront kernel and the n = 22..40 workgroup kernel warm-start Jacobi along the last mesh
    axis on meshes of more than ~2000 points (the eigenvector gauge then depends on the chain of predecessors;
    n = 6 goes through the register kernel, which starts cold).  Chains are aligned to the global index, so
    periodic images, halo rows and windows cut at any offset must still reproduce the full array bit for
    bit, and closed-loop Berry phases along the chain direction must stay gauge invariant."""
    from oracle import tb_oracle as orc
    rng = np.random.default_rng(77)                        # two groups of `half` bands, gap ~4
    n = 2 * half
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.2, 1.1]], rng.random((n, 2)))
    m.set_onsite(list(np.linspace(-3.0, -2.2, half)) + list(np.linspace(2.2, 3.0, half)))
    for i in range(n):
        for j in range(i, n):
            for R in ([0, 0], [1, 0], [0, 1], [1, -1]):
                if i == j and R == [0, 0]:
                    continue
                m.set_hop(0.25 * 3.0 / half * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    start = [0.1, -0.2]
    full = tb.wf_array(m, mesh)
    gaps = full.solve_on_grid(start)
    host = full._wfs.copy()
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-11
    for d in range(2):                                     # images are exact copies times the pbc phase
        fac = np.exp(-2j * np.pi * m._orb[:, m._per[d]])
        assert np.max(np.abs(np.take(host, -1, axis=d) - np.take(host, 0, axis=d) * fac)) < 1e-14
    assert ogaps[half - 1] > 1.0                              # the two groups stay separated on this mesh
    for occ in (list(range(half)), list(range(half, n)), list(range(n))):
        for d in (0, 1):
            got = full.berry_phase(occ, d, contin=False)
            assert np.max(np.abs(wrap(got - orc.berry_phase(owfs, 2, occ, d, contin=False)))) < TOL_P
        assert np.max(np.abs(wrap(full.berry_flux(occ, individual_phases=True)
                                  - orc.berry_flux(owfs, 2, occ, individual_phases=True, vectorised=True)))) < TOL_P
    for (o0, n0, o1, n1) in ((0, mesh[0], 0, 53), (5, 9, 0, 53), (0, mesh[0], 7, 30), (3, 20, 21, 32), (mesh[0] - 1, 1, 50, 3)):
        n0e, n1e = max(n0, 2), max(n1, 2)                  # a wf_array axis needs >= 2 points
        o0e, o1e = min(o0, mesh[0] - n0e), min(o1, mesh[1] - n1e)
        w = tb.wf_array(m, [n0e, n1e])
        w.solve_on_grid_window(start, [o0e, o1e], mesh)
        assert np.array_equal(w._wfs, host[o0e:o0e + n0e, o1e:o1e + n1e]), (o0, n0, o1, n1)
    ev1, vec1 = m.solve_all(np.random.default_rng(0).random((300, 2)), eig_vectors=True)   # list mode: warm runs
    ref = orc.solve_all_vec(m, np.random.default_rng(0).random((300, 2)))
    assert np.max(np.abs(ev1 - ref)) < 1e-12
    V = vec1.reshape(n, 300, n)
    assert max(np.max(np.abs(V[:, i].conj() @ V[:, i].T - np.identity(n))) for i in range(300)) < 1e-13
