"""65..1024 states per k WITH eigenvectors by the direct method (pythtb_amd/csrc/tbk_solve_trigv.inl) -- ribbons and slabs, the
models cut_piece / make_supercell produce (pythtb.py:1105-1637) and solve_on_grid then diagonalises with numpy.linalg.eigh
(:944, :2479).  Against numpy and the oracle; clustered spectra (Kramers pairs of spinful ribbons); windows of a mesh."""
import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def _quality(ham, ev, V):
    nk, n = ham.shape[0], ham.shape[1]
    ref = np.linalg.eigvalsh(ham)
    nrm = np.maximum(np.abs(ref).max(axis=1), 1e-300)
    Vk = V.transpose(1, 0, 2)
    res = np.abs(np.einsum("kij,kbj->kbi", ham, Vk) - Vk * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / nrm
    orth = np.abs(np.einsum("kbi,kci->kbc", Vk.conj(), Vk) - np.eye(n)).reshape(nk, -1).max(axis=1)
    return (np.abs(ev.T - ref).max(axis=1) / nrm).max(), res.max(), orth.max()


@pytest.mark.parametrize("kind,width", [("haldane", 35), ("haldane", 120), ("kane_mele", 20), ("kane_mele", 45)])
def test_ribbon_band_structure_with_vectors(tb, kind, width):
    """solve_all(k_path, eig_vectors=True) of a ribbon; the spinful one has exactly degenerate Kramers pairs at k = 0, 1/2."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    bulk = hp.haldane(tb.tb_model, 0.3) if kind == "haldane" else hp.kane_mele(tb.tb_model, "odd")
    rib = hp.quiet(bulk.cut_piece, width, 1, glue_edgs=False)
    n = rib._nsta
    assert 65 <= n <= 1024
    k = np.concatenate([[0.0, 0.5], np.linspace(0.0, 1.0, 21, endpoint=False) + 0.013])[:, None]
    note = _lib.lib.tbk_solver_regime(n, 1, 0, len(k), len(k), 256, 1, None).decode()
    assert note == "trigv"
    ev, V = rib.solve_all(k, eig_vectors=True)
    ham = orc.ham_batch(rib, k)
    q = _quality(ham, ev, V.reshape(n, len(k), n))
    assert max(q) < 3e-13, q
    with _lib.knob("TBK_TRIGV", 0):
        evj = rib.solve_all(k)
    assert np.max(np.abs(ev - evj)) < 1e-11 * np.abs(evj).max()


def test_ribbon_on_a_mesh_against_the_oracle_and_windows(tb):
    """wf_array of a 1-D ribbon mesh: min gaps and the Berry phase of the lower half against the oracle; a window of the mesh
    solved on its own repeats the full solve bit for bit (every point is solved independently)."""
    from oracle import tb_oracle as orc
    # (delta = 1.2: the trivial phase -- no edge states crossing the gap at half filling, so the occupied half is a smooth bundle
    # and its Berry phase is defined on a 31-point loop)
    rib = hp.quiet(hp.haldane(tb.tb_model, 1.2).cut_piece, 40, 1, glue_edgs=False)
    n = rib._nsta                                    # 80 states
    mesh, start = [31], [0.05]
    w = tb.wf_array(rib, mesh)
    gaps = w.solve_on_grid(start)
    owfs, ogaps = orc.solve_on_grid(rib, mesh, start, vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-11
    occ = list(range(n // 2))
    d = w.berry_phase(occ) - orc.berry_phase(owfs, 1, occ)
    assert abs((d + np.pi) % (2 * np.pi) - np.pi) < 1e-8
    host = w.to_host()
    V = host.reshape(mesh[0], n, n)
    assert max(np.max(np.abs(v.conj() @ v.T - np.identity(n))) for v in V) < 1e-12
    # periodic image = first point times the pbc phase
    fac = np.exp(-2j * np.pi * rib._orb[:, rib._per[0]])
    assert np.max(np.abs(host[-1] - host[0] * fac[None, :])) < 1e-13
    ww = tb.wf_array(rib, [9])
    ww.solve_on_grid_window(start, [17], mesh)
    assert np.array_equal(ww.to_host(), host[17:26])


def test_supplied_matrices_with_clusters(tb):
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(9)
    n, nk = 96, 14
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    u = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
    h[0] = 0.0
    h[1] = np.diag(np.arange(n) % 7).astype(complex)                                  # many repeated levels, diagonal
    p = u @ np.diag(np.repeat(np.arange(n // 2, dtype=float), 2)) @ u.conj().T
    h[2] = 0.5 * (p + p.conj().T)                                                     # exact pairs
    p = u @ np.diag(np.arange(n, dtype=float) + np.where(np.arange(n) % 2, 1e-9 - 1.0, 0.0)) @ u.conj().T
    h[3] = 0.5 * (p + p.conj().T)                                                     # pairs split by 1e-9
    p = u @ np.diag(np.repeat(np.arange(n // 4, dtype=float), 4)) @ u.conj().T
    h[4] = 0.5 * (p + p.conj().T)                                                     # fourfold levels
    t = np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    h[5] = t                                                                          # already tridiagonal
    h[6] = np.diag(np.abs(np.arange(n) - (n - 1) / 2.0)) + t                          # Wilkinson: pairs agreeing to rounding
    ev, vec = np.zeros((n, nk)), np.zeros((n, nk, n), dtype=complex)
    _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
    assert np.isfinite(vec).all()
    for i in range(nk):
        q = _quality(h[i:i + 1], ev[:, i:i + 1], vec[:, i:i + 1])
        assert max(q) < 3e-13, (i, q)


def test_large_sizes(tb):
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(10)
    for n, nk in ((560, 6), (1024, 6)):
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = h + h.conj().transpose(0, 2, 1)
        assert _lib.lib.tbk_solver_regime(n, 1, 2, nk, nk, 256, 1, None).decode() == "trigv"
        ev, vec = np.zeros((n, nk)), np.zeros((n, nk, n), dtype=complex)
        _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(h.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
        q = _quality(h, ev, vec)
        assert max(q) < 1e-12, (n, q)
