"""Position-operator / hybrid-Wannier path (the first "next" row of SURVEY.md 8f): the oracle
against the fixtures on CPU, the HIP path against the same fixtures on the GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_tables, load_golden
import helpers as hp
from oracle import tb_oracle as orc


def wrap(d):
    return (np.asarray(d) + np.pi) % (2 * np.pi) - np.pi


def slab_check(model, solve_all, wf_factory, px_tol):
    """Runs tests/test_examples/slab/cubic_slab_hwf/run.py through the given implementation."""
    g = load_golden("hwf_cubic_slab")
    ref_dir = os.path.join(GOLDEN, "reference_tests", "cubic_slab_hwf")
    evals = solve_all(model, g["kpts"])
    assert np.max(np.abs(evals - g["evals"])) < 1e-11
    np.testing.assert_allclose(evals, np.load(os.path.join(ref_dir, "evals.npy")), rtol=1e-8, atol=1e-11)
    hwfc, px, pexp, extra = wf_factory(model)
    assert np.max(np.abs(hwfc - g["hwfc"])) < 1e-10
    np.testing.assert_allclose(hwfc, np.load(os.path.join(ref_dir, "hwfc.npy")), rtol=1e-8, atol=1e-10)
    # per-band expectations depend on the basis inside degenerate groups (the slab has symmetry-
    # degenerate bands on this mesh); their sum over the occupied group = tr X is invariant
    assert np.max(np.abs(pexp.sum(axis=-1) - g["pexp"].sum(axis=-1))) < 1e-10
    assert np.max(np.abs(pexp.sum(axis=-1) - hwfc.sum(axis=-1))) < 1e-10
    assert np.max(np.abs(wrap(2 * np.pi * (px - g["px"])))) < px_tol
    xm, c25, proj = extra
    assert np.max(np.abs(c25 - g["hwfc_2_5"])) < 1e-10
    assert np.max(np.abs(np.linalg.eigvalsh(xm) - g["hwfc_2_5"])) < 1e-10      # gauge-free view of X
    assert np.max(np.abs(np.abs(xm) - np.abs(g["xmat_2_5"]))) < 1e-10
    assert np.max(np.abs(proj - g["hwfproj_2_5"])) < 1e-9


def test_oracle_hwf_cubic_slab():
    g = load_golden("hwf_cubic_slab")
    m = orc.Model.from_tables(golden_tables(g))
    nl, nk = 9, 9

    def wf(model):
        wfs, _ = orc.solve_on_grid(model, [nk, nk], [0.0, 0.0])
        hw = np.zeros((nk, nk, nl, model._norb), dtype=complex)
        hwfc = np.zeros((nk, nk, nl))
        pexp = np.zeros((nk, nk, nl))
        for ix in range(nk):
            for iy in range(nk):
                hwfc[ix, iy], hw[ix, iy] = orc.position_hwf(model, wfs[ix, iy][:nl], 2, True, "orbital")
                pexp[ix, iy] = orc.position_expectation(model, wfs[ix, iy][:nl], 2)
        orc.impose_pbc(model, hw, 0, 0)
        orc.impose_pbc(model, hw, 1, 1)
        px = np.array([orc.berry_phase(hw, 2, [n], 0) / (2 * np.pi) for n in range(nl)])
        ev = wfs[2, 5][[0, 3, 4]]
        c25, w25 = orc.position_hwf(model, ev, 2, True, "wavefunction")
        return hwfc, px, pexp, (orc.position_matrix(model, ev, 2), c25, np.abs(w25) ** 2)

    slab_check(m, lambda mm, k: orc.solve_all_vec(mm, k), wf, 1e-9)


def test_oracle_hwf_haldane_ribbon():
    g = load_golden("hwf_haldane_ribbon")
    m = orc.Model.from_tables(golden_tables(g))
    ev, vec = orc.solve_all_vec(m, g["k_vec"], True)
    ev = ev - 0.25
    assert np.max(np.abs(ev - g["rib_eval"])) < 1e-11
    flat, pos = [], []
    for i in range(ev.shape[1]):
        pos.append(orc.position_expectation(m, vec[:, i], 1))
        flat.append(orc.position_hwf(m, vec[ev[:, i] < 0.0, i], 1))
    assert np.array_equal([len(f) for f in flat], g["nocc"])
    assert np.max(np.abs(np.concatenate(flat) - g["hwfcs_flat"])) < 1e-10
    # per-state expectations are gauge free only for non-degenerate states
    gaps = np.minimum(np.diff(ev, axis=0, prepend=-np.inf), np.diff(ev, axis=0, append=np.inf))
    ok = (gaps > 1e-6).T
    assert np.max(np.abs((np.array(pos) - g["pos_exps"])[ok])) < 1e-8


@pytest.mark.gpu
def test_gpu_hwf_cubic_slab():
    import pythtb_amd as tb
    g = load_golden("hwf_cubic_slab")
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    nl, nk = 9, 9

    def wf(model):
        arr = tb.wf_array(model, [nk, nk])
        arr.solve_on_grid([0.0, 0.0])
        hwf_arr = arr.empty_like(nsta_arr=nl)
        hwfc = np.zeros((nk, nk, nl))
        pexp = np.zeros((nk, nk, nl))
        for ix in range(nk):
            for iy in range(nk):
                val, vec = arr.position_hwf([ix, iy], occ=list(range(nl)), dir=2, hwf_evec=True, basis="orbital")
                hwfc[ix, iy] = val
                hwf_arr[ix, iy] = vec
                pexp[ix, iy] = arr.position_expectation([ix, iy], list(range(nl)), 2)
        hwf_arr.impose_pbc(0, 0)
        hwf_arr.impose_pbc(1, 1)
        px = np.array([hwf_arr.berry_phase(dir=0, occ=[n]) / (2.0 * np.pi) for n in range(nl)])
        # batched extension == the per-point loop
        c_all, w_all = arr.position_hwf_mesh(list(range(nl)), 2, hwf_evec=True, basis="orbital")
        assert np.max(np.abs(c_all - hwfc)) < 1e-12 and w_all.shape == (nk, nk, nl, model._norb)
        assert np.max(np.abs(arr.position_hwf_mesh(list(range(nl)), 2) - hwfc)) < 1e-12
        xm = arr.position_matrix([2, 5], [0, 3, 4], 2)
        c25, w25 = arr.position_hwf([2, 5], [0, 3, 4], 2, hwf_evec=True)
        assert np.max(np.abs(xm @ w25.T - w25.T * c25)) < 1e-12              # rows are eigenvectors of X
        return hwfc, px, pexp, (xm, c25, np.abs(w25) ** 2)

    slab_check(m, lambda mm, k: mm.solve_all(k), wf, 1e-9)
    with pytest.raises(Exception, match="periodic direction"):
        m.position_matrix(m.solve_one([0.1, 0.2], eig_vectors=True)[1], 0)
    with pytest.raises(Exception, match="out of range"):
        m.position_matrix(m.solve_one([0.1, 0.2], eig_vectors=True)[1], 3)
    with pytest.raises(Exception, match="Basis must be"):
        m.position_hwf(m.solve_one([0.1, 0.2], eig_vectors=True)[1], 2, hwf_evec=True, basis="nope")


@pytest.mark.gpu
def test_gpu_hwf_haldane_ribbon_and_spin():
    import pythtb_amd as tb
    g = load_golden("hwf_haldane_ribbon")
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    ev, vec = m.solve_all(g["k_vec"], eig_vectors=True)
    ev = ev - 0.25
    assert np.max(np.abs(ev - g["rib_eval"])) < 1e-11
    flat = [m.position_hwf(vec[ev[:, i] < 0.0, i], 1) for i in range(ev.shape[1])]
    assert np.array_equal([len(f) for f in flat], g["nocc"])
    assert np.max(np.abs(np.concatenate(flat) - g["hwfcs_flat"])) < 1e-10
    pos = np.array([m.position_expectation(vec[:, i], dir=1) for i in range(ev.shape[1])])
    gaps = np.minimum(np.diff(ev, axis=0, prepend=-np.inf), np.diff(ev, axis=0, append=np.inf))
    assert np.max(np.abs((pos - g["pos_exps"])[(gaps > 1e-6).T])) < 1e-8
    # spinor model with a non-periodic direction, against the oracle on the same eigenvectors
    s = hp.model_from_tables(tb.tb_model, golden_tables(load_golden("point_spin_chain")))
    e, v = s.solve_one([0.17], eig_vectors=True)
    for occ in ([0, 1, 2], [1, 4], list(range(6))):
        x = s.position_matrix(v[occ], 1)
        assert np.max(np.abs(x - orc.position_matrix(s, v[occ], 1))) < 1e-13
        c, w = s.position_hwf(v[occ], 1, hwf_evec=True, basis="orbital")
        assert w.shape == (len(occ), 3, 2)
        assert np.max(np.abs(c - orc.position_hwf(s, v[occ], 1))) < 1e-12
        pos = np.repeat(s._orb[:, 1], 2)
        wf = w.reshape(len(occ), -1)
        assert np.max(np.abs(np.einsum("ij,j,ij->i", wf.conj(), pos, wf).real - c)) < 1e-12   # <hwf|r|hwf> = centre


@pytest.mark.gpu
@pytest.mark.parametrize("nsub", [1, 2, 3, 5, 7, 8, 9, 10])
def test_gpu_position_matrix_tile_kernel_equals_the_entry_kernel(nsub):
    """Batches of up to 8 states take k_position_matrix_tile (four points per wavefront through LDS, 2 x 2 blocks of X per lane;
    tbk_position.hip): X, the centres and the functions against the thread-per-entry kernel (TBK_POS_TILE=0) on a ragged batch
    (99 points: the last wavefront holds three), with an occupied-band list that is not the lowest bands, and on a point list."""
    import ctypes as C
    import pythtb_amd as tb
    from pythtb_amd import _lib
    g = load_golden("hwf_cubic_slab")
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    w = tb.wf_array(m, [11, 9])
    w.solve_on_grid([0.0, 0.0])
    h = w._ensure_dev()
    n = m._nsta
    rng = np.random.default_rng(nsub)
    occ = np.sort(rng.choice(n, nsub, replace=False)).astype(np.int32)
    pos = np.ascontiguousarray(np.repeat(m._orb[:, 2], m._nspin), dtype=float)
    pts = np.ascontiguousarray(rng.choice(99, 70, replace=False).astype(np.int64))
    got = {}
    for tile in (0, 1):
        with _lib.knob("TBK_POS_TILE", tile):
            out = []
            for plist in (None, pts):
                nk = 99 if plist is None else len(plist)
                x = np.zeros((nk, nsub, nsub), dtype=complex)
                c = np.zeros((nk, nsub))
                f = np.zeros((nk, nsub, n), dtype=complex)
                _lib.check(_lib.lib.tbk_wfs_position_hwf(h, None if plist is None else plist.ctypes.data_as(C.POINTER(C.c_int64)), nk,
                                                         _lib.iptr(occ), nsub, _lib.dptr(pos), _lib.dptr(x.view(float)), _lib.dptr(c),
                                                         _lib.dptr(f.view(float)), 1))
                out.append((x, c, f))
            got[tile] = out
    host = w.to_host().reshape(99, n, n)
    for (x0, c0, f0), (x1, c1, f1), plist in zip(got[0], got[1], (np.arange(99), pts)):
        assert np.max(np.abs(x1 - x0)) < 1e-15 * max(1.0, np.abs(pos).max())
        v = host[plist][:, occ, :]
        ref = np.einsum("kmj,j,knj->kmn", v.conj(), pos, v)
        assert np.max(np.abs(x1 - ref)) < 1e-13
        assert np.max(np.abs(c1 - c0)) < 1e-12 and np.max(np.abs(np.abs(f1) - np.abs(f0))) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("nl,nocc", [(16, 16), (16, 13), (20, 11), (20, 17), (24, 24), (29, 26), (32, 32), (33, 21)])
def test_gpu_position_hwf_mesh_9_to_16_states(nl, nocc):
    """9..16 states: k_position_matrix_tile with all 64 lanes on one point; 17..32 states (round 6): k_position_matrix_tile4, a 4 x 4
    block of X per lane.  The centres of a slab on a mesh against the
    thread-per-entry kernel (TBK_POS_TILE=0) and against numpy on the position matrix built from the downloaded states."""
    import pythtb_amd as tb
    from pythtb_amd import _lib
    rng = np.random.default_rng(nl)
    m = hp.quiet(tb.tb_model, 2, 3, np.identity(3), [[0.0, 0.0, float(i)] for i in range(nl)], per=[0, 1])
    m.set_onsite(list(0.1 * rng.standard_normal(nl)))
    for i in range(nl):
        m.set_hop(-1.0, i, i, [1, 0, 0])
        m.set_hop(-0.8, i, i, [0, 1, 0])
        if i + 1 < nl:
            m.set_hop(-0.7 + 0.1j, i, i + 1, [0, 0, 0])
    w = tb.wf_array(m, [11, 9])
    w.solve_on_grid([0.0, 0.0])
    occ = list(range(nocc))
    a = w.position_hwf_mesh(occ, 2)
    with _lib.knob("TBK_POS_TILE", 0):
        b = w.position_hwf_mesh(occ, 2)
    assert np.max(np.abs(a - b)) < 1e-12
    host = w.to_host()
    pos = m._orb[:, 2]
    for (i, j) in ((0, 0), (4, 7), (10, 8)):
        v = host[i, j][occ]
        x = np.einsum("mj,j,nj->mn", v.conj(), pos, v)
        assert np.max(np.abs(np.linalg.eigvalsh(x) - a[i, j])) < 1e-12
