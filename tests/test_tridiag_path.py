"""n = 17..64 states per k, large batches: the three-kernel tridiagonal path (pythtb_amd/csrc/tbk_solve_qlw.inl --
Householder reduction in LDS, lane-per-matrix implicit QL, rotation replay) that stands in for the reference's
numpy.linalg.eigh / eigvalsh per k-point (pythtb.py:939-947).  The dispatcher only takes it above 8 x CUs matrices;
TBK_QLW_MIN=0 sends small batches through it too, so that the oracle (a per-k Python loop) can follow.

Checked: supplied matrices against numpy (eigenvalues, residuals, orthonormality; random, zero, degenerate, already
tridiagonal, decoupled blocks, sizes on both sides of the 32- and 48-row kernels), models on k lists and meshes against
the oracle, batches cut into several workspace chunks, ragged last wavefronts, and that windows of a mesh reproduce
the whole mesh bit for bit."""
import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def _eigh_batch(h, vectors=True):
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    nk, n = h.shape[0], h.shape[1]
    hh = np.ascontiguousarray(h, dtype=complex)
    ev = np.zeros((n, nk))
    vec = np.zeros((n, nk, n), dtype=complex) if vectors else None
    _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hh.view(float)), nk, _lib.dptr(ev),
                                  _lib.dptr(vec.view(float)) if vectors else None))
    return ev, vec


def _special(n, rng):
    out = [np.zeros((n, n), dtype=complex),
           np.diag(np.arange(n) % 3).astype(complex),                                   # degenerate, already diagonal
           np.kron(np.eye(n // 2 + 1), [[0, 1], [1, 0]])[:n, :n].astype(complex),       # repeated +-1
           (np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)).astype(complex),  # already tridiagonal
           np.diag(np.full(n - 2, 1j), 2) + np.diag(np.full(n - 2, -1j), -2),           # two decoupled chains
           np.identity(n, dtype=complex) * 3.5]
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    out.append(1e-9 * (a + a.conj().T) + np.diag(np.repeat([1.0, 2.0], [n // 2, n - n // 2])))   # two tight clusters
    out.append(1e6 * (a + a.conj().T))                                                   # large scale
    blk = np.zeros((n, n), dtype=complex)                                                # block diagonal: early deflation
    blk[:n // 3, :n // 3] = (a + a.conj().T)[:n // 3, :n // 3]
    blk[n // 3:, n // 3:] = (a + a.conj().T)[n // 3:, n // 3:]
    out.append(blk)
    return out


@pytest.mark.parametrize("n", [17, 20, 21, 23, 25, 28, 29, 31, 32, 33, 40, 47, 48, 49, 63, 64])
def test_supplied_matrices_against_numpy(n):
    from pythtb_amd import _lib
    rng = np.random.default_rng(100 + n)
    nk = 130                                                   # two full wavefronts of the QL kernel and a ragged third
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    sp = _special(n, rng)
    h[3:3 + len(sp)] = sp
    with _lib.knob("TBK_QLW_MIN", 0):
        ev, vec = _eigh_batch(h)
        with _lib.knob("TBK_QLW_BISECT", 0):
            ev_only, _ = _eigh_batch(h, vectors=False)         # tridiagonalise + lane-per-matrix QL
        ev_bis, _ = _eigh_batch(h, vectors=False)              # small batch: tridiagonalise + bisection
    with _lib.knob("TBK_QLW", 0):
        ev_jac, _ = _eigh_batch(h, vectors=False)
    ref = np.linalg.eigvalsh(h).T
    scale = np.maximum(np.max(np.abs(ref), axis=0), 1.0)
    assert np.max(np.abs(ev - ref) / scale) < 5e-14
    assert np.array_equal(ev, ev_only)                         # the same reduction and the same QL in both forms
    assert np.max(np.abs(ev_bis - ref) / scale) < 5e-14 and np.all(np.diff(ev_bis, axis=0) >= 0.0)
    assert np.max(np.abs(ev_jac - ref) / scale) < 5e-13
    V = vec.transpose(1, 0, 2)                                 # [k][band][component]
    for i in range(nk):
        s = scale[i]
        assert np.max(np.abs(h[i] @ V[i].T - V[i].T * ev[:, i])) < 2e-13 * s * n
        assert np.max(np.abs(V[i].conj() @ V[i].T - np.identity(n))) < 1e-13 * n


def test_chunked_batches_equal_one_batch():
    """A workspace budget of 1 MiB cuts 2600 matrices of n = 24 into three chunks (the smallest chunk is 1024)."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(7)
    n, nk = 24, 2600
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    with _lib.knob("TBK_QLW_MIN", 0):
        ev1, v1 = _eigh_batch(h)
        with _lib.knob("TBK_QLW_WS_MB", 1), _lib.knob("TBK_QLW_BISECT", 0):
            ev3, v3 = _eigh_batch(h)
            e3, _ = _eigh_batch(h, vectors=False)
    assert np.array_equal(ev1, ev3) and np.array_equal(v1, v3) and np.array_equal(ev1, e3)
    assert np.max(np.abs(ev1 - np.linalg.eigvalsh(h).T)) < 2e-13


def _ribbon(tb, width):
    """Haldane ribbon: `width` cells along a2, periodic along a1 (examples/haldane_fin.py builds the same cut)."""
    m = hp.haldane(tb.tb_model, delta=0.2)
    return hp.quiet(m.cut_piece, width, 1, glue_edgs=False)


@pytest.mark.parametrize("width", [10, 17, 26])                # n = 20, 34, 52: sparse models (no R-grouped table)
def test_ribbon_on_a_k_list_against_oracle(tb, width):
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    m = _ribbon(tb, width)
    n = 2 * width
    k = np.linspace(-0.5, 0.5, 67)
    with _lib.knob("TBK_QLW_MIN", 0):
        ev, vec = m.solve_all(k, eig_vectors=True)
        with _lib.knob("TBK_QLW_BISECT", 0):
            ev_only = m.solve_all(k)
    ev_bis = m.solve_all(k)                                    # the default route of a short eigenvalue-only list: bisection
    oev = orc.solve_all(m, [[x] for x in k])
    assert np.max(np.abs(ev - oev)) < 1e-12 and np.max(np.abs(ev_bis - oev)) < 1e-12
    assert np.array_equal(ev, ev_only)
    for i in range(0, len(k), 5):
        H = m._gen_ham([k[i]])
        V = vec[:, i, :]
        assert np.max(np.abs(H @ V.T - V.T * ev[:, i])) < 1e-12
        assert np.max(np.abs(V.conj() @ V.T - np.identity(n))) < 1e-13 * n


def test_dense_model_on_a_mesh_against_oracle(tb):
    """A dense 3-D model of 24 orbitals (R-grouped table) through solve_on_grid: minimum gaps and Berry phases of the
    lower half of the bands against the oracle's per-k loop (pythtb.py:2499-2511, :3002-3025)."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    rng = np.random.default_rng(5)
    norb = 24
    m = hp.quiet(tb.tb_model, 3, 3, np.identity(3), rng.random((norb, 3)))
    m.set_onsite(np.where(np.arange(norb) < norb // 2, -2.5, 2.5) + 0.2 * rng.standard_normal(norb))
    for R in ([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]):
        for i in range(norb):
            for j in range(norb):
                if R == [0, 0, 0] and j <= i:
                    continue
                m.set_hop(0.08 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    mesh, start = [5, 6, 9], [0.1, -0.2, 0.05]
    with _lib.knob("TBK_QLW_MIN", 0):
        w = tb.wf_array(m, mesh)
        gaps = w.solve_on_grid(start)
        ph = w.berry_phase(range(norb // 2), 2, contin=False)
        fl = w.berry_flux(range(norb // 2), dirs=[0, 1])
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-12
    assert ogaps[norb // 2 - 1] > 0.5
    oph = orc.berry_phase(owfs, 3, list(range(norb // 2)), 2, contin=False)
    d = np.angle(np.exp(1j * (np.asarray(ph) - np.asarray(oph))))
    assert np.max(np.abs(d)) < 1e-9
    ofl = orc.berry_flux(owfs, 3, list(range(norb // 2)), dirs=[0, 1], vectorised=True)
    assert np.max(np.abs(np.asarray(fl) - np.asarray(ofl))) < 1e-9


def test_windows_reproduce_the_whole_mesh(tb):
    """Every point is solved on its own: shard windows are bit-identical to the unsharded array."""
    from pythtb_amd import _lib, shard
    m = hp.quiet(hp.haldane(tb.tb_model, delta=0.3).make_supercell, [[3, 0], [0, 3]])   # 18 orbitals, 2-D
    mesh, start = [13, 9], [-0.5, -0.5]
    with _lib.knob("TBK_QLW_MIN", 0):
        full = tb.wf_array(m, mesh)
        gaps = full.solve_on_grid(start)
        host = full.to_host().copy()
        gmins = []
        for r in range(3):
            row0, nrows = shard.split_rows(mesh[0], 3, r)
            w = tb.wf_array(m, [nrows, mesh[1]])
            gmins.append(w.solve_on_grid_window(start, [row0, 0], mesh))
            assert np.array_equal(w.to_host(), host[row0:row0 + nrows])
    assert np.array_equal(np.min(gmins, axis=0), gaps)
    # periodic images: last row / column = first times the orbital phase (pythtb.py:2733-2736)
    ph0 = np.exp(-2j * np.pi * m._orb[:, 0])
    assert np.max(np.abs(host[-1, 3] - host[0, 3] * ph0)) < 1e-15


def test_direct_vectors_of_17_to_32_states_list_next_to_nothing_and_chunks_in_flight_change_no_bit(tb):
    """Round 6 (k_hh32 + k_ql32_lanes + k_tw32_vectors): on a generic model hardly any matrix goes on the list for the QL replay
    (the context's counter, tbk_ctx_solver_stats); TBK_TW16_GAPTOL=1e300 lists every one; and two or three chunks in flight on the
    side streams (the default from 16384 matrices) leave every bit of the array and of the minimal gaps where it was."""
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    m = hp.random_model(tb.tb_model, 22, 3, 1, seed=8, nhop=120, rmax=1)
    mesh, start = [17, 16, 18], [0.0, 0.1, -0.2]
    npt = int(np.prod(mesh))
    w = tb.wf_array(m, mesh)
    ctx.solver_stats(reset=True)
    gaps = w.solve_on_grid(start)
    host = w.to_host().copy()
    assert ctx.solver_stats(reset=True)["listed_matrices"] <= npt // 100
    for knobs in ({"TBK_QLW_STREAMS": 2}, {"TBK_QLW_STREAMS": 3}, {"TBK_QLW_STREAMS": 1}):
        (kn, kv), = knobs.items()
        with _lib.knob(kn, kv):
            w2 = tb.wf_array(m, mesh)
            g2 = w2.solve_on_grid(start)
        assert np.array_equal(g2, gaps) and np.array_equal(w2.to_host(), host), knobs
    with _lib.knob("TBK_TW16_GAPTOL", "1e300"):
        w3 = tb.wf_array(m, mesh)
        g3 = w3.solve_on_grid(start)
    assert ctx.solver_stats(reset=True)["listed_matrices"] == npt
    assert np.array_equal(g3, gaps)                               # (the eigenvalues come from the same kernel either way)
    k = np.array([start]) + np.array([[3 / 16.0, 5 / 15.0, 7 / 17.0]])
    H = m._gen_ham(k[0])
    for arr in (host, w3.to_host()):
        V = arr[3, 5, 7]
        ev = np.linalg.eigvalsh(H)
        assert np.max(np.abs(H @ V.T - V.T * ev)) < 1e-12 and np.max(np.abs(V.conj() @ V.T - np.eye(22))) < 1e-12


@pytest.mark.parametrize("half", [9, 11, 13, 16])
def test_models_with_paired_levels_at_every_k_take_the_replay_by_themselves(tb, half):
    """Two decoupled identical copies of a random model of `half` orbitals, orbitals interleaved: every level of the model is doubly
    degenerate at every k (the situation of spin-degenerate and Kramers-paired bands).  The twisted factorisation gives both members of
    a pair the same vector, so k_tw32_vectors would list every matrix.  The library notices at upload (one generic k-point solved:
    ModelView::pairs_hint) and such a model takes the rotation replay from the start (Q from k_hh32's reflector record through
    k_tw32_vectors in its Q-only form): nothing is listed, and the eigenvectors are
    eigenvectors, orthonormal inside the degenerate spaces too.  TBK_TW32=3 overrides the hint: every matrix listed, same quality."""
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(4 + half)
    orb = np.repeat(rng.random((half, 3)), 2, axis=0)
    m = hp.quiet(tb.tb_model, 3, 3, np.identity(3), orb)
    m.set_onsite(list(np.repeat(rng.standard_normal(half), 2)))
    seen = set()
    while len(seen) < 5 * half:
        i, j = int(rng.integers(half)), int(rng.integers(half))
        R = tuple(int(x) for x in rng.integers(-1, 2, size=3))
        if (i == j and R == (0, 0, 0)) or (i, j, R) in seen or (j, i, tuple(-r for r in R)) in seen:
            continue
        seen.add((i, j, R))
        amp = complex(rng.standard_normal(), rng.standard_normal())
        for c in (0, 1):
            m.set_hop(amp, 2 * i + c, 2 * j + c, list(R))
    mesh, start = [17, 16, 10], [0.05, -0.1, 0.2]
    npt = int(np.prod(mesh))
    n = 2 * half
    for knob, want in ((1, 0), (3, npt)):
        with _lib.knob("TBK_TW32", knob):
            w = tb.wf_array(m, mesh)
            ctx.solver_stats(reset=True)
            w.solve_on_grid(start)
            host = w.to_host()
            assert ctx.solver_stats(reset=True)["listed_matrices"] == want
        for idx in ((0, 0, 0), (3, 5, 7), (16, 15, 9), (8, 1, 4)):
            k = np.array(start) + np.array(idx) / (np.array(mesh) - 1.0)
            H = m._gen_ham(k)
            V = host[idx]
            ev = np.linalg.eigvalsh(H)
            assert np.max(np.abs(ev[0::2] - ev[1::2])) < 1e-12                       # pairs indeed
            assert np.max(np.abs(H @ V.T - V.T * ev)) < 1e-12 and np.max(np.abs(V.conj() @ V.T - np.eye(n))) < 1e-12


@pytest.mark.parametrize("n", [19, 22, 27, 32])
def test_structured_matrices_of_17_to_32_states(n):
    """profiles/tw32_fuzz.py in small: direct sums of random blocks (T splits), identical blocks (levels degenerate ACROSS blocks: each
    vector must stay inside its own block), blocks coupled by 1e-18 .. 1e-6, the same under a diagonal unitary -- against numpy."""
    rng = np.random.default_rng(300 + n)
    nk = 96

    def herm(k, m):
        a = rng.standard_normal((k, m, m)) + 1j * rng.standard_normal((k, m, m))
        return a + a.conj().transpose(0, 2, 1)
    cut, half = int(rng.integers(3, n - 3)), n // 2
    two = np.zeros((nk, n, n), dtype=complex)
    two[:, :cut, :cut], two[:, cut:, cut:] = herm(nk, cut), herm(nk, n - cut)
    b = herm(nk, half)
    same = np.zeros((nk, n, n), dtype=complex)
    same[:, :half, :half], same[:, half:2 * half, half:2 * half] = b, b
    if n % 2:
        same[:, -1, -1] = rng.standard_normal(nk)
    near = two.copy()
    near[:, cut - 1, cut] = near[:, cut, cut - 1] = 10.0 ** rng.uniform(-18, -6, nk)
    ph = np.exp(2j * np.pi * rng.random((nk, n)))
    from pythtb_amd import _lib
    for h in (two, same, near, ph[:, :, None] * same * ph.conj()[:, None, :]):
        with _lib.knob("TBK_QLW_MIN", 0):
            ev, vec = _eigh_batch(h)
        ref = np.linalg.eigvalsh(h).T
        nrm = np.abs(ref).max(axis=0)
        V = vec.transpose(1, 0, 2)
        assert (np.abs(ev - ref) / nrm).max() < 5e-14
        assert (np.abs(np.einsum("kij,kbj->kbi", h, V) - V * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / nrm).max() < 5e-14
        assert np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).max() < 5e-14


def test_large_batch_takes_the_path_by_default(tb):
    """Above 8 x CUs matrices no knob is needed: 4096 k-points of a 20-orbital ribbon, against LAPACK on H(k)."""
    m = _ribbon(tb, 10)
    k = np.linspace(0.0, 1.0, 4096, endpoint=False)
    ev = m.solve_all(k)
    for i in (0, 1, 777, 4095):
        assert np.max(np.abs(ev[:, i] - np.linalg.eigvalsh(m._gen_ham([k[i]])))) < 1e-12
    # the bulk gap of the ribbon closes only through the two edge bands
    assert ev.shape == (20, 4096)


@pytest.mark.parametrize("n", [3, 4, 9, 12, 16, 20, 40])
def test_small_entries_below_the_subdiagonal_are_not_dropped(n):
    """A strong tridiagonal part plus couplings 1e-9 of it everywhere else (ribbon Hamiltonians near k = 0 look like that:
    real hoppings, imaginary parts ~ k).  The Householder step must decide "nothing to annihilate" on the entries below
    the subdiagonal alone (LAPACK zlarfg); deciding through their sum with the subdiagonal entry loses everything below
    1e-8 of it and shifts eigenvalues by up to 1e-9 (seen on a 20-orbital Haldane ribbon at k = 0.001: 4e-10)."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(40 + n)
    nk = 192
    t = rng.uniform(1.0, 2.0, (nk, n - 1))
    h = np.zeros((nk, n, n), dtype=complex)
    for i in range(nk):
        h[i] = np.diag(rng.standard_normal(n)) + np.diag(t[i], 1) + np.diag(t[i], -1)
    x = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    x = x + x.conj().transpose(0, 2, 1)
    for scale in (1e-9, 1e-12):
        hs = h + scale * x
        with _lib.knob("TBK_QLW_MIN", 0), _lib.knob("TBK_QL16_MIN", 0):
            ev, vec = _eigh_batch(hs)
            ev_only, _ = _eigh_batch(hs, vectors=False)
        ref = np.linalg.eigvalsh(hs).T
        assert np.max(np.abs(ev - ref)) < 1e-13 * n
        assert np.max(np.abs(ev_only - ref)) < 1e-13 * n
        V = vec.transpose(1, 0, 2)
        assert max(np.max(np.abs(hs[i] @ V[i].T - V[i].T * ev[:, i])) for i in range(nk)) < 1e-13 * n


def test_chunked_mesh_equals_one_batch(tb):
    """solve_on_grid of a ribbon on 2601 k-points in three workspace chunks: the same array and gaps as in one."""
    from pythtb_amd import _lib
    m = _ribbon(tb, 10)
    with _lib.knob("TBK_QLW_MIN", 0):
        w1 = tb.wf_array(m, [2601])
        g1 = w1.solve_on_grid([0.0])
        a1 = w1.to_host().copy()
        with _lib.knob("TBK_QLW_WS_MB", 1):
            w3 = tb.wf_array(m, [2601])
            g3 = w3.solve_on_grid([0.0])
            a3 = w3.to_host()
    assert np.array_equal(g1, g3) and np.array_equal(a1, a3)
    ev = m.solve_all(np.linspace(0.0, 1.0, 2601)[:-1])
    assert np.max(np.abs(np.min(ev[1:] - ev[:-1], axis=1) - g1)) < 1e-12


@pytest.mark.parametrize("route", [{"TBK_TW32": 0}, {"TBK_TW16_GAPTOL": "1e300"}],
                         ids=["replay-on-every-matrix", "every-matrix-listed-by-k_tw32_vectors"])
def test_rotation_record_overflow_falls_back_to_jacobi(tb, route):
    """The QL kernel records at most 3 n^2 rotations per matrix (4 x the usual count).  If a matrix needs more, the call is
    repeated on the Jacobi kernels; TBK_QLW_CAP=64 provokes that for every matrix here -- on the replay path of rounds 2-5, and
    (round 6) for the matrices k_tw32_vectors lists, whose record comes from the list-mode launch of the QL kernel."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(12)
    n, nk = 24, 70
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    m = _ribbon(tb, 10)
    k = np.linspace(-0.5, 0.5, 33)
    (rk, rv), = route.items()
    with _lib.knob("TBK_QLW_MIN", 0), _lib.knob("TBK_QLW_CAP", 64), _lib.knob(rk, rv):
        ev, vec = _eigh_batch(h)
        mev, mvec = m.solve_all(k, eig_vectors=True)
        w = tb.wf_array(m, [41])
        gaps = w.solve_on_grid([0.0])
        host = w.to_host()
    with _lib.knob("TBK_QLW", 0):
        ev_j, vec_j = _eigh_batch(h)
        w2 = tb.wf_array(m, [41])
        gaps_j = w2.solve_on_grid([0.0])
    assert np.array_equal(ev, ev_j) and np.array_equal(vec, vec_j)          # literally the Jacobi kernels' output
    assert np.array_equal(gaps, gaps_j) and np.array_equal(host, w2.to_host())
    assert np.max(np.abs(ev - np.linalg.eigvalsh(h).T)) < 1e-12
    for i in (0, 16, 32):
        H = m._gen_ham([k[i]])
        V = mvec[:, i, :]
        assert np.max(np.abs(H @ V.T - V.T * mev[:, i])) < 1e-12


def test_sweep_record_overflow_of_the_16_lane_form_falls_back(tb):
    """n = 9..16 with eigenvectors: the three-kernel form records at most 64 sweeps per matrix; TBK_QLW_CAP=64 leaves room
    for four, every matrix overflows and the batch is repeated on the single kernel -- the same bits as with TBK_QL16_SPLIT=0."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(21)
    n, nk = 14, 300
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    with _lib.knob("TBK_QL16_MIN", 0), _lib.knob("TBK_QL16_SPLIT_MIN", 0), _lib.knob("TBK_TW16", 0):
        ev3, v3 = _eigh_batch(h)                                   # three kernels (the QL-replay form)
        with _lib.knob("TBK_QLW_CAP", 64):
            ev_f, v_f = _eigh_batch(h)                             # overflow -> repeated on the single kernel
        with _lib.knob("TBK_QL16_SPLIT", 0):
            ev1, v1 = _eigh_batch(h)                               # single kernel
    assert np.array_equal(ev_f, ev1) and np.array_equal(v_f, v1)
    ref = np.linalg.eigvalsh(h).T
    assert np.max(np.abs(ev3 - ref)) < 1e-13 and np.max(np.abs(ev1 - ref)) < 1e-13
    V = v3.transpose(1, 0, 2)
    assert max(np.max(np.abs(h[i] @ V[i].T - V[i].T * ev3[:, i])) for i in range(nk)) < 1e-13
    assert max(np.max(np.abs(V[i].conj() @ V[i].T - np.identity(n))) for i in range(nk)) < 1e-13


@pytest.mark.parametrize("tw16", [0, 1])
def test_three_kernel_16_lane_form_on_a_mesh(tb, tw16):
    """cubic16 on a small 3-D mesh forced through the three-kernel forms -- tridiagonalise | lane-per-matrix QL | replay (tw16 = 0)
    and tridiagonalise | eigenvalues | twisted-factorisation vectors (tw16 = 1, tbk_solve_tw16.inl): gaps and Berry phases against
    the oracle, the single kernel's eigenvalue gaps, and bit-identical shard windows (the route is decided on the global mesh)."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib, shard
    m = hp.cubic16(tb.tb_model)
    mesh, start = [7, 6, 9], [0.05, -0.1, 0.2]
    with _lib.knob("TBK_QL16_MIN", 0), _lib.knob("TBK_QL16_SPLIT_MIN", 0), _lib.knob("TBK_TW16", tw16):
        w = tb.wf_array(m, mesh)
        gaps = w.solve_on_grid(start)
        host = w.to_host().copy()
        ph = w.berry_phase(range(8), 2, contin=False)
        for r in range(2):
            row0, nrows = shard.split_rows(mesh[0], 2, r)
            ww = tb.wf_array(m, [nrows, mesh[1], mesh[2]])
            ww.solve_on_grid_window(start, [row0, 0, 0], mesh)
            assert np.array_equal(ww.to_host(), host[row0:row0 + nrows])
        with _lib.knob("TBK_QL16_SPLIT", 0):
            w1 = tb.wf_array(m, mesh)
            gaps1 = w1.solve_on_grid(start)
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-12 and np.max(np.abs(gaps1 - ogaps)) < 1e-12
    oph = orc.berry_phase(owfs, 3, list(range(8)), 2, contin=False)
    assert np.max(np.abs(np.angle(np.exp(1j * (np.asarray(ph) - np.asarray(oph)))))) < 1e-9
    V = host.reshape(-1, 16, 16)
    assert max(np.max(np.abs(v.conj() @ v.T - np.identity(16))) for v in V) < 1e-13


@pytest.mark.parametrize("n,nk", [(65, 6), (100, 5), (137, 4), (256, 3), (300, 2), (513, 2)])
def test_eigenvalues_only_above_64_states(n, nk):
    """Eigenvalue-only solves of 65..1024 states: Householder tridiagonalisation with the matrix in L2, then one thread per
    eigenvalue bisecting on the Sturm count (tbk_solve_trig.inl) -- against numpy.linalg.eigvalsh and the Jacobi solvers."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(500 + n)
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    if nk >= 4:
        h[1] = np.diag(np.arange(n) % 5).astype(complex)                                   # degenerate, diagonal
        h[2] = np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)                   # already tridiagonal
        h[3] = 0.0
    ev, _ = _eigh_batch(h, vectors=False)
    ref = np.linalg.eigvalsh(h).T
    scale = max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(ev - ref)) < 1e-13 * scale
    assert np.all(np.diff(ev, axis=0) >= 0.0)
    with _lib.knob("TBK_TRIG", 0):
        ev_j, _ = _eigh_batch(h, vectors=False)
    assert np.max(np.abs(ev_j - ref)) < 5e-12 * scale


def test_wide_ribbon_band_structure_against_oracle(tb):
    """solve_all without eigenvectors on a Haldane ribbon of 45 cells (90 states, a sparse model): the reference's
    band-structure call (pythtb.py:955-1079 with eig_vectors=False)."""
    from oracle import tb_oracle as orc
    m = _ribbon(tb, 45)
    k = np.linspace(0.0, 1.0, 41)
    ev = m.solve_all(k)
    oev = orc.solve_all(m, [[x] for x in k])
    assert ev.shape == (90, 41)
    assert np.max(np.abs(ev - oev)) < 1e-12
    # near k = 0 the imaginary parts of H(k) are tiny (the case that needs LAPACK's zlarfg decision)
    kz = np.array([0.0, 1e-3, 2.44140625e-4])
    assert np.max(np.abs(m.solve_all(kz) - orc.solve_all(m, [[x] for x in kz]))) < 1e-12


def test_spinful_ribbon_band_structure_against_oracle(tb):
    """A Kane-Mele ribbon of 20 cells (nspin = 2: 80 states, 2 x 2 blocks in the tables) through the eigenvalue-only path, and
    one of 8 cells (32 states) through the tridiagonal path with eigenvectors."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    km = hp.kane_mele(tb.tb_model, "odd")
    wide = hp.quiet(km.cut_piece, 20, 1, glue_edgs=False)
    k = np.linspace(0.0, 1.0, 23)
    ev = wide.solve_all(k)
    assert ev.shape == (80, 23)
    assert np.max(np.abs(ev - orc.solve_all(wide, [[x] for x in k]))) < 1e-12
    narrow = hp.quiet(km.cut_piece, 8, 1, glue_edgs=False)
    with _lib.knob("TBK_QLW_MIN", 0):
        ev2, vec2 = narrow.solve_all(k, eig_vectors=True)
    assert np.max(np.abs(ev2 - orc.solve_all(narrow, [[x] for x in k]))) < 1e-12
    for i in (0, 7, 22):
        H = narrow._gen_ham([k[i]]).reshape(32, 32)
        V = vec2[:, i].reshape(32, 32)
        assert np.max(np.abs(H @ V.T - V.T * ev2[:, i])) < 1e-12
