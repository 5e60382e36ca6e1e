"""n = 9..16 states with eigenvectors on chip-filling batches: tridiagonalise | eigenvalues | twisted-factorisation vectors +
MFMA Newton-Schulz + back-transformation (pythtb_amd/csrc/tbk_solve_tw16.inl), the path config E's solve_on_grid takes.
The reference runs numpy.linalg.eigh per matrix (pythtb.py:939-947): eigenvalues, residuals and orthonormality against it;
the near-degenerate matrices the path hands to the QL-replay kernels; the bits of that fallback."""
import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def _eigh_batch(h):
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    nk, n, _ = h.shape
    ev, vec = np.zeros((n, nk)), np.zeros((n, nk, n), dtype=complex)
    hc = np.ascontiguousarray(h)
    _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
    return ev, vec


def _quality(h, ev, vec):
    n, nk = ev.shape
    V = vec.transpose(1, 0, 2)
    ref = np.linalg.eigvalsh(h)
    nrm = np.maximum(np.abs(ref).max(axis=1), 1e-300)
    res = np.abs(np.einsum("kij,kbj->kbi", h, V) - V * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / nrm
    orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).reshape(nk, -1).max(axis=1)
    return (np.abs(ev.T - ref).max(axis=1) / nrm).max(), res.max(), orth.max()


def _forced():
    """every size onto the twisted path: no Jacobi for small batches, three-kernel forms from the first matrix"""
    import contextlib
    from pythtb_amd import _lib
    st = contextlib.ExitStack()
    st.enter_context(_lib.knob("TBK_QL16_MIN", 0))
    st.enter_context(_lib.knob("TBK_QL16_SPLIT_MIN", 0))
    return st


def special_matrices(n, rng):
    """the structures the twisted factorisation has to survive"""
    out = []
    out.append(np.zeros((n, n), dtype=complex))                                       # zero matrix: T splits everywhere
    out.append(np.diag(np.arange(n, dtype=float)).astype(complex))                   # diagonal
    out.append(np.diag([1.0] * (n // 2) + [2.0] * (n - n // 2)).astype(complex))     # two exactly repeated levels
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    a = a + a.conj().T
    b = a.copy()
    b[: n // 2, n // 2:] = 0
    b[n // 2:, : n // 2] = 0
    out.append(b)                                                                     # block diagonal
    c = a.copy()
    c[: n // 2, n // 2:] *= 1e-9
    c[n // 2:, : n // 2] *= 1e-9
    out.append(c)                                                                     # weakly coupled blocks
    u = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
    pairs = u @ np.diag(np.repeat(np.arange(n // 2 + 1, dtype=float), 2)[:n]) @ u.conj().T
    out.append(0.5 * (pairs + pairs.conj().T))                                        # exact pairs (Kramers-like)
    near = u @ np.diag(np.arange(n, dtype=float) + np.where(np.arange(n) % 2, 1e-7 - 1.0, 0.0)) @ u.conj().T
    out.append(0.5 * (near + near.conj().T))                                          # pairs split by 1e-7
    t = np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    out.append(t.astype(complex))                                                     # already tridiagonal (Toeplitz)
    w = np.diag(np.abs(np.arange(n) - (n - 1) / 2.0)) + t                             # Wilkinson-like: pairs agreeing to many digits
    out.append(w.astype(complex))
    out.append((a * 1e-150))                                                          # tiny scale
    out.append((a * 1e+120))                                                          # huge scale
    # round 5 (k_e16's scaled Sturm recurrence and its choice of Newton-Schulz step):
    g = np.diag(10.0 ** (-20.0 * np.arange(n))).astype(complex)                       # graded tridiagonal: the products of the e^2 leave
    for i in range(n - 1):                                                            # the range of the scaled recurrence (pivots instead)
        g[i, i + 1] = g[i + 1, i] = 10.0 ** (-20.0 * i - 10.0)
    out.append(g)
    lev = np.arange(n, dtype=float)
    lev[1], lev[2] = lev[0] + 1e-3 * n, lev[0] + 2e-3 * n                              # three levels inside 1e-2 |T|: the full step
    tri = u @ np.diag(lev) @ u.conj().T
    out.append(0.5 * (tri + tri.conj().T))
    lev = np.arange(n, dtype=float)
    lev[n // 2] = lev[n // 2 - 1] + 3e-4 * n                                          # a pair inside 1e-3 |T| but far above gaptol
    par = u @ np.diag(lev) @ u.conj().T
    out.append(0.5 * (par + par.conj().T))
    return out


@pytest.mark.parametrize("n", [9, 10, 12, 13, 15, 16])
def test_supplied_matrices_against_lapack(tb, n):
    rng = np.random.default_rng(100 + n)
    nk = 3000
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    sp = special_matrices(n, rng)
    h[:len(sp)] = np.array(sp)
    with _forced():
        ev, vec = _eigh_batch(h)
        ev2, vec2 = _eigh_batch(h)
    assert np.isfinite(vec).all()
    assert np.array_equal(ev, ev2) and np.array_equal(vec, vec2)                      # run to run: the same bits
    e_err, res, orth = _quality(h, ev, vec)
    assert e_err < 1e-14 and res < 2e-14 and orth < 2e-14, (e_err, res, orth)
    assert np.all(np.diff(ev, axis=0) >= 0.0)
    # each special case on its own, so that one of them cannot hide behind the maximum
    for i in range(len(sp)):
        q = _quality(h[i:i + 1], ev[:, i:i + 1], vec[:, i:i + 1])
        assert max(q) < 2e-14, (i, q)


@pytest.mark.parametrize("n", [9, 12, 16])
def test_eigenvalues_only_through_the_fused_kernel(tb, n):
    """round 5: eigenvalue-only calls of 9..16 states (supplied matrices and k lists, any count) run k_e16<.., false> -- no reflector
    record, no vectors, nothing listed (a lane whose Newton iteration does not settle bisects on the exact count).  Against LAPACK,
    against round 3's pair of kernels (TBK_E16_EVALS=0), sorted, the same bits for a matrix whatever shares its batch, and a NaN raises."""
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(300 + n)
    nk = 9000
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    sp = special_matrices(n, rng)
    h[:len(sp)] = np.array(sp)

    def evals(hh):
        ev = np.zeros((n, len(hh)))
        hc = np.ascontiguousarray(hh)
        _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), len(hh), _lib.dptr(ev), None))
        return ev
    with _lib.knob("TBK_E16", 1), _lib.knob("TBK_E16_EVALS", 1):
        ev = evals(h)
        few = evals(h[:len(sp) + 5])                      # another batch size, other neighbours in the wavefront
        shuffled = evals(h[::-1])
    with _lib.knob("TBK_E16_EVALS", 0):
        ev0 = evals(h)
    ref = np.linalg.eigvalsh(h)
    nrm = np.maximum(np.abs(ref).max(axis=1), 1e-300)
    assert np.all(np.diff(ev, axis=0) >= 0.0)
    err = np.abs(ev.T - ref).max(axis=1) / nrm
    err[0] = 0.0                                          # (the zero matrix: |T| = 0)
    assert err.max() < 1e-14, (err.argmax(), err.max())
    assert np.abs(ev[:, 0]).max() < 1e-30
    assert np.max(np.abs(ev - ev0).max(axis=0)[1:] / nrm[1:]) < 1e-14
    assert np.array_equal(few, ev[:, :few.shape[1]])
    assert np.array_equal(shuffled[:, ::-1], ev)
    bad = h[:40].copy()
    bad[7, 2, 3] = bad[7, 3, 2] = np.nan
    with _lib.knob("TBK_E16", 1), _lib.knob("TBK_E16_EVALS", 1):
        with pytest.raises(_lib.TbkError):
            evals(bad)
        assert np.array_equal(evals(h[:40]), ev[:, :40])   # (the flag does not stick)


@pytest.mark.parametrize("n", [9, 13, 16])
def test_eigenvalues_only_on_hard_spectra(tb, n):
    """profiles/evals16_stress.py in small: spectra on which a Newton iteration that is trusted goes wrong (pairs split by 1e-15..1e-3
    merge to their mean, a triple within 1e-6, graded over twelve decades, an (n - 2)-fold level) -- the eigenvalue-only form checks
    every result against the exact Sturm count and bisects where it fails: rounding level on all of them, and sorted."""
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(900 + n)
    nk = 400

    def conj_unitary(lev):
        a = rng.standard_normal((len(lev), n, n)) + 1j * rng.standard_normal((len(lev), n, n))
        q = np.linalg.qr(a)[0]
        h = (q * lev[:, None, :]) @ q.conj().transpose(0, 2, 1)
        return 0.5 * (h + h.conj().transpose(0, 2, 1))
    base = np.sort(rng.standard_normal((nk, (n + 1) // 2)), axis=1)
    split = 10.0 ** rng.uniform(-15, -3, size=(nk, 1))
    fam = {
        "pairs": conj_unitary(np.repeat(base, 2, axis=1)[:, :n] + np.tile([0.0, 1.0], (n + 1) // 2)[:n] * split),
        "graded": conj_unitary(np.sort(10.0 ** rng.uniform(-6, 6, size=(nk, n)), axis=1)),
        "rank2": conj_unitary(np.concatenate([np.ones((nk, n - 2)), 1.0 + rng.standard_normal((nk, 2))], axis=1)),
        "triple": conj_unitary(np.sort(rng.standard_normal((nk, n)), axis=1) * np.r_[np.ones(n - 3), 0, 0, 0] +
                               np.r_[np.zeros(n - 3), 0.5, 0.5 + 1e-6, 0.5 + 2e-6]),
    }
    with _lib.knob("TBK_E16", 1), _lib.knob("TBK_E16_EVALS", 1):
        for name, h in fam.items():
            ev = np.zeros((n, nk))
            hc = np.ascontiguousarray(h)
            _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev), None))
            ref = np.linalg.eigvalsh(h)
            err = (np.abs(ev.T - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
            assert np.all(np.diff(ev, axis=0) >= 0.0), name
            assert err < 2e-14, (name, err)


def test_tridiagonal_newton_schulz_step_against_the_full_one(tb):
    """round 5: k_e16 takes the Newton-Schulz step with V^T V - I cut to its tridiagonal part unless T splits or eigenvalues crowd;
    TBK_E16_NS_FULL=1 is round 4's form (the full step on the matrix cores for every matrix).  Same eigenvalues bit for bit (the
    step only touches the vectors), both orthonormal to rounding, the vectors equal up to that rounding; crowded and split matrices
    (the special ones) take the full step either way: identical bits."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(55)
    n, nk = 16, 4000
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    sp = special_matrices(n, rng)
    h[:len(sp)] = np.array(sp)
    with _forced(), _lib.knob("TBK_E16", 1):              # (whatever the environment says: this test is about the fused kernel's two forms)
        with _lib.knob("TBK_E16_NS_FULL", 0):
            ev_t, v_t = _eigh_batch(h)
        with _lib.knob("TBK_E16_NS_FULL", 1):
            ev_f, v_f = _eigh_batch(h)
    assert np.array_equal(ev_t, ev_f)
    q_t, q_f = _quality(h, ev_t, v_t), _quality(h, ev_f, v_f)
    assert max(q_t) < 2e-14 and max(q_f) < 2e-14, (q_t, q_f)
    assert q_t[2] < 2.0 * q_f[2] + 1e-15, (q_t, q_f)                                   # orthonormality: no worse than the full step's
    assert not np.array_equal(v_t[:, len(sp):], v_f[:, len(sp):])                      # (the generic matrices do take the short form)
    assert np.max(np.abs(v_t - v_f)) < 1e-12
    for i in (0, 1, 2, 3, 5):                                                          # zero, diagonal, repeated levels, blocks, exact pairs
        assert np.array_equal(v_t[:, i], v_f[:, i]), i


def test_every_matrix_listed_equals_the_ql_replay_form_bit_for_bit(tb):
    """TBK_TW16_GAPTOL = 1e300 lists every matrix: the result is the QL-replay kernels' own (TBK_TW16=0), bit for bit; and a
    threshold of 0 lists none but the residual failures -- exact pairs still come out orthonormal (their T splits)."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(7)
    n, nk = 16, 2500
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    sp = special_matrices(n, rng)
    h[:len(sp)] = np.array(sp)
    with _forced():
        with _lib.knob("TBK_TW16", 0):
            ev0, v0 = _eigh_batch(h)
        with _lib.knob("TBK_TW16_GAPTOL", "1e300"):
            ev1, v1 = _eigh_batch(h)
        ev2, v2 = _eigh_batch(h)
    assert np.array_equal(ev0, ev1) and np.array_equal(v0, v1)
    assert max(_quality(h, ev2, v2)) < 2e-14
    # where the two forms differ they differ by rounding only (eigenvalues), and the unlisted vectors are not the replay's bits
    assert np.max(np.abs(ev2 - ev0).max(axis=0) / np.maximum(np.abs(ev0).max(axis=0), 1e-300)) < 1e-13
    assert not np.array_equal(v2[:, len(sp):], v0[:, len(sp):])


def test_listed_matrices_overflowing_the_sweep_record_fall_back_to_the_single_kernel(tb):
    """the fallback's rotation record holds TBK_QLW_CAP / 16 sweeps per listed matrix; when that overflows, the call is repeated
    on the single kernel like the QL-replay form's own overflow (same bits as TBK_QL16_SPLIT=0)."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(21)
    n, nk = 14, 300
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    with _forced():
        with _lib.knob("TBK_TW16_GAPTOL", "1e300"), _lib.knob("TBK_QLW_CAP", 64):
            ev_f, v_f = _eigh_batch(h)
        with _lib.knob("TBK_QL16_SPLIT", 0):
            ev1, v1 = _eigh_batch(h)
    assert np.array_equal(ev_f, ev1) and np.array_equal(v_f, v1)


def test_k_list_and_mesh_windows(tb):
    """cubic16 (config E's model): solve_all with eigenvectors on a k list against numpy on _gen_ham's matrices, and
    solve_on_grid windows of one global mesh -- every point is solved on its own, so windows repeat the full solve bit for bit."""
    from pythtb_amd import _lib
    m = hp.cubic16(tb.tb_model)
    rng = np.random.default_rng(3)
    k = rng.random((700, 3))
    with _forced():
        ev, vec = m.solve_all(k, eig_vectors=True)
        mesh, start = [9, 7, 12], [0.1, -0.2, 0.05]
        w = tb.wf_array(m, mesh)
        gaps = w.solve_on_grid(start)
        host = w.to_host().copy()
        for off, sub in (([0, 0, 0], [4, 7, 12]), ([3, 2, 5], [6, 5, 7]), ([8, 6, 11], [1 + 0, 1 + 0, 1 + 0])):
            sub = [max(2, s) for s in sub]
            off = [min(o, mesh[d] - sub[d]) for d, o in enumerate(off)]
            ww = tb.wf_array(m, sub)
            ww.solve_on_grid_window(start, off, mesh)
            sl = tuple(slice(o, o + s) for o, s in zip(off, sub))
            assert np.array_equal(ww.to_host(), host[sl])
    H = np.array([m._gen_ham(kk) for kk in k[::7]])
    q = _quality(H, ev[:, ::7], vec[:, ::7, :])
    assert max(q) < 2e-14, q
    V = host.reshape(-1, 16, 16)
    assert max(np.max(np.abs(v.conj() @ v.T - np.identity(16))) for v in V) < 1e-13
    assert gaps.shape == (15,) and gaps[7] > 0.3


def test_padded_sizes_on_a_mesh_against_the_oracle(tb):
    """n = 11 (five padding rows): gaps and Berry phases of a random 11-orbital model against the oracle."""
    from oracle import tb_oracle as orc
    m = hp.random_model(tb.tb_model, 11, 2, 1, 31)
    mesh, start = [13, 17], [0.0, 0.3]
    with _forced():
        w = tb.wf_array(m, mesh)
        gaps = w.solve_on_grid(start)
        ph = w.berry_phase(range(5), 1, contin=False)
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-12
    oph = orc.berry_phase(owfs, 2, list(range(5)), 1, contin=False)
    if np.min(ogaps[4]) > 1e-3:                       # (the occupied set must be separated for its phase to be defined)
        assert np.max(np.abs(np.angle(np.exp(1j * (np.asarray(ph) - np.asarray(oph)))))) < 1e-9


def test_spin_degenerate_bands_on_a_mesh(tb):
    """An 8-orbital model with two spin components and no spin-orbit coupling: EVERY level of every H(k) is exactly twice
    degenerate (Kramers-like pairs), T splits -- or nearly so -- at every point.  The fused kernel has to put the two members of a
    pair into different unreduced blocks (or list the matrix): eigenvalues against numpy, residuals and orthonormality of what
    comes out, ascending bands, on a k list and on a mesh with its windows."""
    rng = np.random.default_rng(17)
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((8, 2)), nspin=2)
    m.set_onsite(list(rng.standard_normal(8)))
    for i in range(8):
        for j in range(i + 1, 8):
            m.set_hop(0.3 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0])
    for R in ([1, 0], [0, 1], [1, 1]):
        for i in range(8):
            for j in range(8):
                if rng.random() < 0.4:
                    m.set_hop(0.2 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    k = rng.random((900, 2))
    with _forced():
        ev, vec = m.solve_all(k, eig_vectors=True)
        w = tb.wf_array(m, [21, 18])
        gaps = w.solve_on_grid([0.05, -0.1])
        host = w.to_host().copy()
        ww = tb.wf_array(m, [7, 9])
        ww.solve_on_grid_window([0.05, -0.1], [5, 4], [21, 18])
        assert np.array_equal(ww.to_host(), host[5:12, 4:13])
    H = np.array([m._gen_ham(kk).reshape(16, 16) for kk in k[::9]])
    q = _quality(H, ev[:, ::9], vec.reshape(16, len(k), 16)[:, ::9, :])
    assert max(q) < 2e-14, q
    assert np.all(np.diff(ev, axis=0) >= 0.0)
    assert np.max(np.abs(ev[0::2] - ev[1::2])) < 1e-13                    # the pairs
    assert np.max(gaps[0::2]) < 1e-13 and np.min(gaps[1::2]) >= 0.0       # gaps inside the pairs vanish
    V = host.reshape(-1, 16, 16)
    assert max(np.max(np.abs(v.conj() @ v.T - np.identity(16))) for v in V) < 1e-13


def test_twins_are_repaired_in_the_kernel_not_listed(tb):
    """Bands degenerate on the whole mesh used to send EVERY matrix of the fused kernel to the QL-replay list (4 x the time): Newton
    with the multiplicity of the bracket finds the double root, and the second eigenvector of a pair comes from inverse iteration
    against the first (e16_twin).  The fallback must now be a small part of the launch; pairs split by 1e-7 and by 1e-12 (no
    degeneracy, merely close) take the same route and come out to rounding level."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(23)
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((8, 2)), nspin=2)
    m.set_onsite(list(rng.standard_normal(8)))
    for i in range(8):
        for j in range(i + 1, 8):
            m.set_hop(0.3 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0])
    for R in ([1, 0], [0, 1]):
        for i in range(8):
            for j in range(8):
                if rng.random() < 0.4:
                    m.set_hop(0.2 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    ctx = _lib.default_context()
    w = tb.wf_array(m, [257, 257])
    with _lib.knob("TBK_E16", 1):                        # (the fused kernel, whatever the environment says)
        w.solve_on_grid([0.0, 0.0])
        ctx.prof_enable(1)
        ctx.prof_reset()
        ctx.solver_stats(reset=True)
        gaps = w.solve_on_grid([0.0, 0.0])
        rep = ctx.prof_report()
        ctx.prof_enable(0)
    listed = ctx.solver_stats()["listed_matrices"]
    # a count, not a time (tbk_ctx_solver_stats): before the repair all 256^2 matrices of this mesh were listed
    assert "e16" in rep and listed <= 256 * 256 // 100, (listed, rep)
    assert np.max(gaps[0::2]) < 1e-13
    V = w.to_host().reshape(-1, 16, 16)[::97]
    assert max(np.max(np.abs(v.conj() @ v.T - np.identity(16))) for v in V) < 1e-13
    # supplied matrices with pairs split by a chosen amount
    n, nk = 16, 4096
    for split in (0.0, 1e-12, 1e-7):
        h = np.empty((nk, n, n), dtype=complex)
        for k in range(nk):
            u = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
            lev = np.repeat(np.sort(rng.standard_normal(n // 2)), 2) + np.tile([0.0, split], n // 2)
            a = (u * lev) @ u.conj().T
            h[k] = 0.5 * (a + a.conj().T)
        with _forced():
            ev, vec = _eigh_batch(h)
        assert max(_quality(h, ev, vec)) < 2e-14, split
        assert np.all(np.diff(ev, axis=0) >= 0.0)
