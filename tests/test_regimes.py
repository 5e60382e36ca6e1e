"""Every regime of the batched Hermitian eigen-solver (launch_wave's dispatch, pythtb_amd/csrc/tbk_solve.hip) forced through
the `knob()` context manager on the SAME supplied matrices and compared with LAPACK -- the reference's _sol_ham is one
numpy.linalg.eigh / eigvalsh per matrix (pythtb.py:927-953).  In the driver's suite, so that a threshold moved in the
dispatch cannot silently route a size onto a kernel nobody tested at that size.

Also here: what happens to input the reference cannot solve either -- NaN in the tables or in a supplied matrix
(np.linalg.eigh raises "Eigenvalues did not converge", pythtb.py:939,944)."""
import contextlib

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
    ctx = _lib.default_context()
    nk, n, _ = h.shape
    ev = np.zeros((n, nk))
    vec = np.zeros((n, nk, n), dtype=complex) if vectors else None
    hc = np.ascontiguousarray(h)
    _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev),
                                       _lib.dptr(vec.view(float)) if vectors else None))
    return ev, vec


def _matrices(n, nk, seed):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = h + h.conj().transpose(0, 2, 1)
    h[0] = np.diag(np.arange(n, dtype=float))                                   # diagonal
    if nk > 2:
        h[1] = 0.0                                                              # zero
        u = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
        d = np.repeat(np.arange((n + 1) // 2, dtype=float), 2)[:n]
        p = u @ np.diag(d) @ u.conj().T
        h[2] = 0.5 * (p + p.conj().T)                                           # exact pairs
    return h


# (n, number of matrices, knobs, what the knobs select)
REGIMES = [
    (3, 500, {}, "per-thread direct solver (Householder + QL)"),
    (4, 500, {}, "per-thread direct solver"),
    (5, 300, {}, "register Jacobi"),
    (8, 300, {}, "register Jacobi"),
    (7, 300, {"TBK_REG": 0}, "wavefront / workgroup LDS Jacobi instead of the register kernel"),
    (9, 200, {}, "small batch: workgroup-per-matrix Jacobi"),
    (16, 200, {}, "small batch: workgroup-per-matrix Jacobi"),
    (12, 200, {"TBK_QL16_MIN": 0}, "direct solver on a DPP row, single kernel"),
    (16, 200, {"TBK_QL16_MIN": 0}, "direct solver on a DPP row, single kernel"),
    (11, 200, {"TBK_QL16_MIN": 0, "TBK_QL16_SPLIT_MIN": 0}, "tridiagonalise | eigenvalues | twisted-factorisation vectors"),
    (16, 200, {"TBK_QL16_MIN": 0, "TBK_QL16_SPLIT_MIN": 0}, "tridiagonalise | eigenvalues | twisted-factorisation vectors"),
    (16, 200, {"TBK_QL16_MIN": 0, "TBK_QL16_SPLIT_MIN": 0, "TBK_TW16_GAPTOL": "1e300"}, "... every matrix re-solved by QL replay"),
    (14, 200, {"TBK_QL16_MIN": 0, "TBK_QL16_SPLIT_MIN": 0, "TBK_TW16": 0}, "tridiagonalise | lane-per-matrix QL | replay"),
    (16, 200, {"TBK_QL16": 0, "TBK_FEW_MAX": 0}, "DPP-row Jacobi (n = 15, 16)"),
    (13, 200, {"TBK_QL16": 0, "TBK_ROW16": 0, "TBK_FEW_MAX": 0}, "wavefront LDS Jacobi"),
    (17, 100, {}, "small batch: workgroup Jacobi"),
    (24, 100, {"TBK_QLW_MIN": 0}, "tridiagonal path (k_hh32: matrix in registers), register replay (32 lanes)"),
    (17, 100, {"TBK_QLW_MIN": 0, "TBK_HH32": 2}, "tridiagonal path, k_hh32<24> forced at 17"),
    (18, 100, {"TBK_QLW_MIN": 0}, "tridiagonal path, k_hh32<24>"),
    (25, 100, {"TBK_QLW_MIN": 0}, "tridiagonal path, k_hh32<32>"),
    (32, 100, {"TBK_QLW_MIN": 0}, "tridiagonal path, k_hh32<32>"),
    (17, 101, {"TBK_QLW_MIN": 0}, "tridiagonal path: k_hh32<20> leaves the reflector record, k_ql32_lanes<20>, k_tw32_vectors<20, record>"),
    (20, 101, {"TBK_QLW_MIN": 0}, "the same at the top of the 20-row forms"),
    (21, 101, {"TBK_QLW_MIN": 0}, "k_hh32<24> / k_ql32_lanes<24> / k_tw32_vectors<24, record>"),
    (26, 101, {"TBK_QLW_MIN": 0}, "k_hh32<28> / k_ql32_lanes<28> / k_tw32_vectors<28, record>"),
    (28, 101, {"TBK_QLW_MIN": 0}, "the same at the top of the 28-row forms"),
    (27, 101, {"TBK_QLW_MIN": 0, "TBK_TW16_GAPTOL": "1e300"}, "k_tw32_vectors<28> lists every matrix: Q out, replay over the list"),
    (18, 101, {"TBK_QLW_MIN": 0, "TBK_TW16_GAPTOL": "1e300"}, "k_tw32_vectors<20> lists every matrix: Q out, replay over the list"),
    (27, 101, {"TBK_QLW_MIN": 0, "TBK_QLW_STREAMS": 3}, "... three chunks in flight on the side streams (default: two from 16384 matrices)"),
    (19, 101, {"TBK_QLW_MIN": 0, "TBK_QLW_STREAMS": 2, "TBK_TW16_GAPTOL": "3e-2"}, "... two chunks in flight, some matrices listed"),
    (20, 101, {"TBK_QLW_MIN": 0, "TBK_QL32": 0}, "... the QL iteration with (d, e) in LDS (k_tridiag_ql_lanes) instead of k_ql32_lanes<24>"),
    (31, 101, {"TBK_QLW_MIN": 0, "TBK_QL32": 0}, "... the QL iteration with (d, e) in LDS (k_tridiag_ql_lanes) instead of k_ql32_lanes<32>"),
    (23, 101, {"TBK_QLW_MIN": 0, "TBK_TW32": 2}, "k_hh32<24> accumulates Z, k_tw32_vectors<24> multiplies by it (matrix cores)"),
    (30, 101, {"TBK_QLW_MIN": 0, "TBK_TW32": 2}, "k_hh32<32> accumulates Z, k_tw32_vectors<32> multiplies by it (matrix cores)"),
    (26, 101, {"TBK_QLW_MIN": 0, "TBK_TW32": 2, "TBK_TW16_GAPTOL": "3e-2"}, "... some matrices listed: Q left in place, replay"),
    (22, 101, {"TBK_QLW_MIN": 0, "TBK_HH32": 0}, "k_tridiag_lds leaves Z, k_tw32_vectors<24> multiplies by it"),
    (24, 100, {"TBK_QLW_MIN": 0, "TBK_TW32": 0}, "tridiagonal path, the rotations replayed on every matrix (rounds 2-5) instead of k_tw32_vectors<24>"),
    (29, 100, {"TBK_QLW_MIN": 0, "TBK_TW32": 0}, "tridiagonal path, the rotations replayed on every matrix instead of k_tw32_vectors<32>"),
    (21, 101, {"TBK_QLW_MIN": 0, "TBK_TW16_GAPTOL": "1e300"}, "k_tw32_vectors<24> lists every matrix: replay over the list"),
    (32, 101, {"TBK_QLW_MIN": 0, "TBK_TW16_GAPTOL": "1e300"}, "k_tw32_vectors<32> lists every matrix: replay over the list"),
    (27, 101, {"TBK_QLW_MIN": 0, "TBK_TW16_GAPTOL": "3e-2"}, "k_tw32_vectors<32>: some matrices listed, some not"),
    (24, 100, {"TBK_QLW_MIN": 0, "TBK_HH32": 0}, "tridiagonal path, LDS workgroup tridiagonalisation (round 2)"),
    (31, 100, {"TBK_QLW_MIN": 0, "TBK_HH32": 0}, "tridiagonal path, LDS workgroup tridiagonalisation (round 2)"),
    (33, 100, {"TBK_QLW_MIN": 0}, "tridiagonal path, LDS replay"),
    (48, 100, {"TBK_QLW_MIN": 0}, "tridiagonal path, register replay (two wavefronts)"),
    (64, 60, {"TBK_QLW_MIN": 0}, "tridiagonal path, register replay (two wavefronts)"),
    (32, 100, {"TBK_QLW_MIN": 0, "TBK_QLW_REPLAY_REG": 0}, "tridiagonal path, LDS replay forced"),
    (40, 100, {"TBK_QLW": 0}, "workgroup LDS Jacobi"),
    (48, 40, {}, "below the QL batch threshold: the direct method of 65+ states"),
    (64, 300, {}, "below the QL batch threshold: the direct method of 65+ states"),
    (40, 700, {}, "below the QL batch threshold: the direct method of 65+ states"),
    (56, 300, {"TBK_TRIGV": 0}, "workgroup LDS Jacobi"),
    (65, 20, {}, "direct method: tridiagonalise | bisection | twisted vectors | Newton-Schulz | back-transformation"),
    (100, 12, {}, "direct method"),
    (230, 5, {}, "direct method, 8-column strips"),
    (300, 3, {}, "direct method"),
    (65, 20, {"TBK_TRIGV": 0}, "workgroup per L2-resident matrix / whole chip"),
    (100, 12, {"TBK_TRIGV": 0, "TBK_BLOCKED": 1}, "block Jacobi"),
    (100, 12, {"TBK_TRIGV": 0, "TBK_BLOCKED": 0}, "workgroup / whole-chip Jacobi"),
    (130, 6, {"TBK_TRIGV": 0, "TBK_BIG_FROM": 65}, "whole-chip Jacobi rounds"),
    (257, 2, {"TBK_TRIGV": 0}, "whole-chip Jacobi rounds"),
    # launch shapes of the paths above (every knob forced somewhere)
    (30, 8, {"TBK_FEW_NT": 512}, "small batch: workgroup LDS Jacobi on 512 threads"),
    (30, 8, {"TBK_FEW_WARM": 0}, "small batch: workgroup LDS Jacobi, always cold"),
    (13, 200, {"TBK_QL16": 0, "TBK_ROW16": 0, "TBK_FEW_MAX": 0, "TBK_WAVE_RUN": 1}, "wavefront LDS Jacobi, chains of one (cold)"),
    (70, 170, {"TBK_TRIGV": 0, "TBK_WG_NT": 512}, "workgroup per L2-resident matrix on 512 threads"),
    (24, 100, {"TBK_QLW_MIN": 0, "TBK_QLW_NT": 64}, "tridiagonal path, 64 threads per matrix"),
    (24, 100, {"TBK_QLW_MIN": 0, "TBK_QLW_NT": 256}, "tridiagonal path, 256 threads per matrix"),
    (100, 12, {"TBK_TRIGV_NC": 4}, "direct method, 4-column strips"),
    (100, 12, {"TBK_TRIGV_NC": 16}, "direct method, 16-column strips"),
    (24, 100, {"TBK_TRIGV_FROM": 17}, "direct method of 65+ states from 17 on"),
    (12, 200, {"TBK_QL16_MIN": 0, "TBK_QL16_SPLIT_MIN": 0, "TBK_TW16_STREAMS": 2}, "fused 9..16 kernel, two chunks in flight"),
    (12, 200, {"TBK_QL16_MIN": 0, "TBK_QL16_SPLIT_MIN": 0, "TBK_TW16_STREAMS": 1, "TBK_E16": 0}, "three-kernel 9..16 form on the context's stream"),
]
EVAL_ONLY = [
    (12, 200, {}, "tridiagonalise + lane-per-matrix QL (any count)"),
    (12, 200, {"TBK_QL16_EVONLY": 0}, "single replicated kernel"),
    (24, 100, {}, "tridiagonalise in LDS + bisection"),
    (24, 100, {"TBK_QLW_BISECT": 0}, "tridiagonalise in LDS + lane-per-matrix QL"),
    (29, 100, {"TBK_QLW_BISECT": 0}, "k_hh32<32, eigenvalues only> + lane-per-matrix QL in registers (k_ql32_lanes)"),
    (23, 100, {"TBK_QLW_BISECT": 0}, "k_hh32<24, eigenvalues only> + k_ql32_lanes<24>"),
    (18, 100, {"TBK_QLW_BISECT": 0}, "k_hh32<20, eigenvalues only> + k_ql32_lanes<20>"),
    (27, 100, {"TBK_QLW_BISECT": 0}, "k_hh32<28, eigenvalues only> + k_ql32_lanes<28>"),
    (20, 100, {"TBK_QLW_BISECT": 0}, "k_hh32<20, eigenvalues only> + k_ql32_lanes<20>"),
    (26, 100, {"TBK_QLW_BISECT": 0, "TBK_QLW_STREAMS": 3}, "... three chunks in flight"),
    (29, 100, {"TBK_QLW_BISECT": 0, "TBK_QL32": 0}, "k_hh32<32, eigenvalues only> + k_tridiag_ql_lanes"),
    (19, 100, {"TBK_HH32": 0}, "k_tridiag_lds + bisection"),
    (48, 100, {"TBK_QLW": 0}, "Jacobi"),
    (90, 10, {}, "tridiagonalise in L2 + bisection"),
    (90, 10, {"TBK_TRIG": 0}, "Jacobi"),
    (90, 10, {"TBK_TRIG": 0, "TBK_BLOCKED": 1}, "block Jacobi, eigenvalues only"),
    (300, 3, {}, "tridiagonalise in L2 + bisection"),
    (90, 10, {"TBK_TRIG_NT": 256}, "tridiagonalise in L2 on 256 threads + bisection"),
    (300, 3, {"TBK_TRIG_NT": 512}, "tridiagonalise in L2 on 512 threads + bisection"),
]


@contextlib.contextmanager
def knobs(d):
    from pythtb_amd import _lib
    with contextlib.ExitStack() as st:
        for k, v in d.items():
            st.enter_context(_lib.knob(k, v))
        yield


@pytest.mark.parametrize("n,nk,kn,what", REGIMES, ids=["%d-%s" % (r[0], "+".join("%s=%s" % kv for kv in r[2].items()) or "default") for r in REGIMES])
def test_forced_regime_with_vectors(n, nk, kn, what):
    h = _matrices(n, nk, 1000 + n)
    with knobs(kn):
        ev, vec = _eigh_batch(h)
    ref = np.linalg.eigvalsh(h)
    scale = np.maximum(np.abs(ref).max(axis=1), 1.0)
    tol = 3e-13 if n <= 64 else 3e-12            # (Jacobi above 64 states: ~24 n eps)
    assert np.max(np.abs(ev.T - ref) / scale[:, None]) < tol, what
    V = vec.transpose(1, 0, 2)
    res = np.abs(np.einsum("kij,kbj->kbi", h, V) - V * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / scale
    orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).reshape(nk, -1).max(axis=1)
    assert res.max() < 10 * tol and orth.max() < 10 * tol, (what, res.max(), orth.max())
    assert np.all(np.diff(ev, axis=0) >= 0.0)


@pytest.mark.parametrize("n,nk,kn,what", EVAL_ONLY, ids=["%d-%s" % (r[0], "+".join("%s=%s" % kv for kv in r[2].items()) or "default") for r in EVAL_ONLY])
def test_forced_regime_eigenvalues_only(n, nk, kn, what):
    h = _matrices(n, nk, 2000 + n)
    with knobs(kn):
        ev, _ = _eigh_batch(h, vectors=False)
    ref = np.linalg.eigvalsh(h)
    scale = np.maximum(np.abs(ref).max(axis=1), 1.0)
    assert np.max(np.abs(ev.T - ref) / scale[:, None]) < (3e-13 if n <= 64 else 3e-12), what
    assert np.all(np.diff(ev, axis=0) >= 0.0)


def test_device_exp_2_pi_i_x_is_accurate(tb):
    """expi2pi (pythtb_amd/csrc/tbk_solve.hip: exact reduction + two polynomials, replaces the library's sincospi) seen through
    a one-orbital chain: E(k) = cos(2 pi k) and the stored component exp(-2 pi i k) -- |error| <= 3e-16 on random arguments,
    quadrant boundaries, huge and tiny ones; exact values where they are representable."""
    m = hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[1.0]])
    m.set_hop(0.5, 0, 0, [1])
    rng = np.random.default_rng(1)
    special = np.concatenate([np.arange(-32, 33) / 16.0, [1e6 + 0.3, -1e9 - 0.125, 0.5 - 2.0 ** -53, 2.0 ** -60, 2.0 ** 52 + 0.5,
                                                            0.125 + 2.0 ** -55, 0.125 - 2.0 ** -56]])
    k = np.concatenate([rng.uniform(-2, 2, 100000), special])
    ev, vec = m.solve_all(k, eig_vectors=True)
    kl = k.astype(np.longdouble)
    ang = 2 * np.pi * (kl - np.rint(kl))
    assert np.max(np.abs(ev[0] - np.cos(ang))) <= 3e-16
    assert np.max(np.abs(vec[0, :, 0].real - np.cos(ang))) <= 3e-16 and np.max(np.abs(vec[0, :, 0].imag + np.sin(ang))) <= 3e-16
    exact = m.solve_all(np.array([0.0, 0.25, 0.5, 0.75, 1.0, -0.25, 3.5]))[0]
    assert np.array_equal(exact, np.array([1.0, 0.0, -1.0, 0.0, 1.0, 0.0, -1.0]))


# ------------------------------------------------------------------ k lists of 2..4 states: one or two k-points per lane
@pytest.mark.parametrize("n,nk", [(2, 1), (2, 255), (2, 513), (2, 70001), (3, 1000), (4, 769), (1, 300)])
def test_two_points_per_lane_is_bit_identical(tb, n, nk):
    """k_solve_small_multi (chip-filling k lists, pythtb_amd/csrc/tbk_solve.hip) forms every point with the same operations in
    the same order as the one-point kernel: the same bits, whatever the list length (points past the end are not stored)."""
    from pythtb_amd import _lib
    m = _chain(tb, n, 0.3) if n > 1 else hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[0.0]])
    if n == 1:
        m.set_hop(0.5, 0, 0, [1])
    k = np.random.default_rng(n * 1000 + nk).uniform(-1.0, 2.0, nk)
    out = {}
    for kpt in (1, 2):
        with _lib.knob("TBK_SMALL_KPT", kpt):
            ev = m.solve_all(k)
            evv, vec = m.solve_all(k, eig_vectors=True)
        out[kpt] = (ev, evv, vec)
    assert np.array_equal(out[1][0], out[2][0])
    assert np.array_equal(out[1][1], out[2][1]) and np.array_equal(out[1][2], out[2][2])
    h = np.array([m._gen_ham([kk]) for kk in k[:50]]) if hasattr(m, "_gen_ham") else None
    if h is not None:
        ref = np.linalg.eigvalsh(h)
        assert np.max(np.abs(out[2][0].T[:50] - ref)) < 1e-13


# ------------------------------------------ 40..64 states, small batches: the direct method on k lists and meshes too
@pytest.mark.parametrize("norb,nk,mesh", [(48, 37, (6, 5)), (64, 20, (4, 4)), (42, 600, (25, 24))])
def test_direct_method_on_small_batches_of_models(tb, norb, nk, mesh):
    """solve_all and solve_on_grid of a 40..64-state model on a batch below the QL threshold take tbk_solve_trigv.inl (kRegimeRules):
    eigenvalues and eigenvectors against LAPACK on the model's own H(k) (pythtb.py:927-953), the array against the list."""
    from pythtb_amd import _lib
    m = hp.random_model(tb.tb_model, norb, 2, 1, seed=norb + nk, nhop=4 * norb, rmax=1)
    assert _lib.lib.tbk_solver_regime(norb, 1, 0, nk, nk, 256, 1, None) == b"trigv"
    k = np.random.default_rng(5).uniform(-0.5, 0.5, (nk, 2))
    ev, vec = m.solve_all(k, eig_vectors=True)
    for i in (0, nk // 2, nk - 1):
        h = m._gen_ham(k[i])
        ref = np.linalg.eigvalsh(h)
        assert np.max(np.abs(ev[:, i] - ref)) < 2e-13 * max(1.0, np.abs(ref).max())
        V = vec[:, i, :]                                   # [band][orbital]
        assert np.max(np.abs(V @ h.T - ev[:, i, None] * V)) < 5e-13 * max(1.0, np.abs(ref).max())
        assert np.max(np.abs(V.conj() @ V.T - np.eye(norb))) < 5e-13
    w = tb.wf_array(m, list(mesh))
    w.solve_on_grid([0.0, 0.0])
    a = w.to_host()
    kk = [0.0 + 2.0 / (mesh[0] - 1), 0.0 + 1.0 / (mesh[1] - 1)]
    ev1, vec1 = m.solve_one(kk, eig_vectors=True)
    # the same point through the mesh kernels: equal up to the phase of each eigenvector
    ov = np.abs(np.einsum("bo,bo->b", a[2, 1].conj(), vec1))
    assert np.max(np.abs(ov - 1.0)) < 1e-10


# ------------------------------------------------- link matrices of 5..8 wide bands: four links per wavefront step or one
@pytest.mark.parametrize("norb,mesh", [(9, (7, 11, 6)), (16, (5, 4, 14)), (12, (3, 9, 10))])
def test_links_four_per_step_equal_one_per_step(tb, norb, mesh):
    """k_chain_links_tile (pythtb_amd/csrc/tbk_berry.hip) against k_chain_links_wave on the same array: string lengths that are
    no multiple of four, odd component counts, every band count 5..8 and every direction -- the same bits (the accumulation
    order of an entry is the same), and the oracle's phases (pythtb.py:3038-3066) to rounding."""
    from pythtb_amd import _lib
    from oracle import tb_oracle as orc
    m = hp.random_model(tb.tb_model, norb, 3, 1, seed=norb, nhop=5 * norb, rmax=1)
    w = tb.wf_array(m, list(mesh))
    w.solve_on_grid([0.1, -0.2, 0.05])
    host = w.to_host()
    for nocc in (5, 6, 7, 8):
        for d in (0, 1, 2):
            got = {}
            with _lib.knob("TBK_CHAIN_PROD", 0):       # (the link-determinant form: the two link kernels)
                for tile in (1, 0):
                    with _lib.knob("TBK_CHAIN_TILE", tile):
                        got[tile] = np.asarray(w.berry_phase(range(nocc), d, contin=False))
            assert np.array_equal(got[1], got[0]), (nocc, d)
            # round 4's default: the string's matrix product on the matrix cores, ONE determinant per string like the reference
            # (pythtb.py:3813-3831) -- the same phases to rounding, run to run the same bits
            prod = np.asarray(w.berry_phase(range(nocc), d, contin=False))
            assert np.array_equal(prod, np.asarray(w.berry_phase(range(nocc), d, contin=False)))
            dphi = (prod - got[1] + np.pi) % (2 * np.pi) - np.pi
            assert np.max(np.abs(dphi)) < 1e-12, (nocc, d, np.max(np.abs(dphi)))
            if nocc in (5, 6, 8) and d in (0, 2):
                ref = np.asarray(orc.berry_phase(host, 3, list(range(nocc)), d, contin=False))
                dphi = (prod - ref + np.pi) % (2 * np.pi) - np.pi
                assert np.max(np.abs(dphi)) < 1e-10
    # bands that are not the lowest ones, in another order (the determinant does not care, the kernels index by occ[])
    sel = [7, 2, 5, 0, 3]
    a = np.asarray(w.berry_phase(sel, 1, contin=False))
    ref = np.asarray(orc.berry_phase(host, 3, sel, 1, contin=False))
    assert np.max(np.abs((a - ref + np.pi) % (2 * np.pi) - np.pi)) < 1e-10


# ---------------------------------------------------------------------------------------------- input nobody can solve
def _chain(tb, n, onsite0):
    m = hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[i / float(n)] for i in range(n)])
    m.set_onsite([onsite0] + [0.1 * i for i in range(1, n)])
    for i in range(n - 1):
        m.set_hop(1.0 + 0.1j * i, i, i + 1, [0])
    m.set_hop(0.7, n - 1, 0, [1])
    return m


@pytest.mark.parametrize("n", [3, 4, 5, 8, 12, 16, 20, 40, 70])
def test_nan_onsite_energy_raises_like_the_reference(tb, n):
    """pythtb.py:939,944: np.linalg.eigh on a Hamiltonian with a NaN raises LinAlgError("Eigenvalues did not converge") for
    n >= 3.  Here every iterative kernel runs into its cap and raises the sticky flag: TbkError, never silent numbers."""
    from pythtb_amd import _lib
    m = _chain(tb, n, float("nan"))
    k = np.linspace(0.0, 1.0, 37, endpoint=False)
    for vec in (False, True):
        with pytest.raises(_lib.TbkError):
            m.solve_all(k, eig_vectors=vec)
    w = tb.wf_array(m, [21])
    with pytest.raises(_lib.TbkError):
        w.solve_on_grid([0.0])
    # the flag does not stick to the next, healthy call
    good = _chain(tb, n, 0.3)
    ev = good.solve_all(k)
    assert np.isfinite(ev).all()


@pytest.mark.parametrize("n", [3, 6, 12, 30])
def test_nan_or_inf_in_a_supplied_matrix(n):
    from pythtb_amd import _lib
    h = _matrices(n, 20, 77)
    bad = h.copy()
    bad[7, 1, 1] = np.nan
    with pytest.raises(_lib.TbkError):
        _eigh_batch(bad)
    worse = h.copy()
    worse[3, 0, 0] = np.inf                     # the reference returns NaN eigenvalues here without raising
    try:
        ev, _ = _eigh_batch(worse)
        assert not np.isfinite(ev[:, 3]).all()
        assert np.isfinite(np.delete(ev, 3, axis=1)).all()
    except _lib.TbkError:
        pass
    ev, vec = _eigh_batch(h)                    # and the next call is clean
    assert np.isfinite(ev).all() and np.isfinite(vec).all()


def test_degenerate_bands_above_64_states_stay_on_the_direct_method(tb):
    """A spinful ribbon-like model without spin-orbit coupling (96 states, every level twice on the whole mesh): the direct method
    for 65..1024 states used to fail its final Gram matrix on the twins (the twisted factorisation gives both the same vector)
    and repeat the whole call on the Jacobi kernels -- 9 x the time, residuals 3e-14.  k_trigv_twisted now repairs the second
    member of a pair by inverse iteration: the call stays on the direct kernels, and the results are at rounding level."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(96)
    norb = 48
    m = hp.quiet(tb.tb_model, 2, 2, [[1.0, 0.0], [0.3, 0.9]], rng.random((norb, 2)), nspin=2)
    m.set_onsite(list(rng.standard_normal(norb)))
    for i in range(norb):
        for j in range(i + 1, min(norb, i + 5)):
            m.set_hop(0.3 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0])
    for R in ([1, 0], [0, 1]):
        for i in range(norb):
            for j in range(max(0, i - 2), min(norb, i + 3)):
                m.set_hop(0.2 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    n = 2 * norb
    ctx = _lib.default_context()
    w = tb.wf_array(m, [13, 11])
    w.solve_on_grid([0.05, -0.1])
    ctx.prof_enable(1)
    ctx.prof_reset()
    gaps = w.solve_on_grid([0.05, -0.1])
    rep = ctx.prof_report()
    ctx.prof_enable(0)
    direct = sum(v["total_ms"] for k, v in rep.items() if k.startswith("trigv_"))
    assert direct > 0.5 * rep["solve_grid"]["total_ms"], rep          # (no second pass on the Jacobi kernels)
    assert np.max(gaps[0::2]) < 1e-12
    host = w.to_host()
    for (i, j) in ((0, 0), (5, 7), (11, 3)):
        H = m._gen_ham([0.05 + i / 12.0, -0.1 + j / 10.0]).reshape(n, n)
        V = host[i, j].reshape(n, n)
        ev = np.einsum("bi,ij,bj->b", V.conj(), H, V).real
        assert np.abs(V @ H.T - ev[:, None] * V).max() < 2e-14 * np.abs(ev).max()
        assert np.abs(V.conj() @ V.T - np.identity(n)).max() < 2e-14
        assert np.max(np.abs(ev - np.linalg.eigvalsh(H))) < 1e-13 * np.abs(ev).max()


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_periodic_image_column_is_stored_by_the_lane_of_column_zero(tb, n):
    """k_grid_rows does not chunk the last column of a closed mesh row: it is the periodic image of the first (pythtb.py:2729-2747)
    and the lane that solves column 0 stores it with the image's orbital phase.  The same bits as solving it on its own
    (TBK_GRID_IMG=0), on row lengths around the chunk size, on 1-D / 2-D / 3-D arrays, with a start point off the origin; min
    gaps equal; windows that do not hold the whole last axis are unaffected (tests/test_tw16_path.py, test_gpu_parity.py)."""
    from pythtb_amd import _lib
    if n == 4:
        m2 = hp.kane_mele(tb.tb_model)
    elif n == 2:
        m2 = hp.haldane(tb.tb_model)
    else:
        m2 = hp.random_model(tb.tb_model, n, 2, 1, seed=40 + n, nhop=4 * n, rmax=1)
    m1 = _chain(tb, n, 0.3) if n > 1 else None
    m3 = hp.random_model(tb.tb_model, n, 3, 1, seed=50 + n, nhop=5 * n, rmax=1)
    cases = [(m2, [65, 65]), (m2, [130, 9]), (m2, [7, 2]), (m2, [5, 129]), (m2, [3, 64]), (m2, [4, 66]), (m3, [5, 6, 66]), (m3, [3, 4, 65])]
    if m1 is not None:
        cases += [(m1, [65]), (m1, [200])]
    for m, mesh in cases:
        start = [0.13, -0.21, 0.37][:len(mesh)]
        out = {}
        for img in (0, 1):
            with _lib.knob("TBK_GRID_IMG", img):
                w = tb.wf_array(m, mesh)
                gaps = w.solve_on_grid(start)
                out[img] = (None if gaps is None else np.array(gaps), w.to_host().copy())
        # n = 3, 4: the orbital phases are folded into the solver's factors and the image is column 0's vector (under column 0's
        # phases) times (image phase conj column-0 phase); the column solved on its own is formed in the same two steps
        # (GridArgs::img_win), so it is the same bits for every n again (round 5 had "to rounding" here)
        assert np.array_equal(out[0][1], out[1][1]), mesh
        if out[0][0] is not None:
            assert np.array_equal(out[0][0], out[1][0]), mesh


@pytest.mark.parametrize("n,rmax", [(2, 3), (2, 4), (3, 3), (4, 3), (4, 4), (4, 6), (1, 5)])
def test_mesh_solve_with_long_hoppings_along_the_last_axis(tb, n, rmax):
    """k_grid_rows<N, PM> is compiled for hopping ranges 0..4 along the last mesh axis and once for any range: every instance
    against the k-list path (term walk, another kernel) on the same points, and the orthonormality of what is stored."""
    m = hp.random_model(tb.tb_model, n, 2, 1, seed=900 + 10 * n + rmax, nhop=6 * n, rmax=rmax)
    m.set_hop(0.21 - 0.13j, 0, n - 1, [1, rmax], mode="add", allow_conjugate_pair=True)      # (the range is reached for sure)
    mesh = [9, 70]
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.1, -0.3])
    host = w.to_host()
    k = np.array([[0.1 + i / (mesh[0] - 1), -0.3 + j / (mesh[1] - 1)] for i in range(mesh[0] - 1) for j in range(mesh[1] - 1)])
    ev, vec = m.solve_all(k, eig_vectors=True)
    V = host[:-1, :-1].reshape(-1, n, n)
    H = np.array([m._gen_ham(kk).reshape(n, n) for kk in k])
    e_mesh = np.einsum("kbi,kij,kbj->kb", V.conj(), H, V).real
    assert np.max(np.abs(e_mesh - ev.T)) < 1e-12
    assert np.max(np.abs(np.einsum("kij,kbj->kbi", H, V) - e_mesh[:, :, None] * V)) < 1e-12
    assert np.max(np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.identity(n))) < 1e-13


@pytest.mark.parametrize("n,rmax,mesh", [(9, 2, [6, 41]), (12, 2, [3, 4, 23]), (16, 2, [5, 4, 19]), (16, 9, [37]), (13, 1, [4, 3, 5, 6])])
def test_mesh_solve_of_9_to_16_states_with_many_lattice_vectors(tb, n, rmax, mesh):
    """round 5: k_e16 on a mesh takes the row's coefficient cells when the model has more than 16 lattice vectors (a dense model of 16
    functions was 10 x slower per point than cubic16, profiles/e16_many_R_probe.py).  Against the reference Hamiltonian of sampled
    points, against the per-vector form (TBK_E16_CELLS=0) to rounding, and a window cut anywhere against the whole array bit for bit
    (wavefronts of four points straddle rows of 19..41 all the time)."""
    from pythtb_amd import _lib
    D = len(mesh)
    m = hp.random_model(tb.tb_model, n, D, 1, seed=1700 + 10 * n + rmax, nhop=40 * n, rmax=rmax)
    start = [0.1, -0.3, 0.2, 0.05][:D]
    with _lib.knob("TBK_E16", 1), _lib.knob("TBK_QL16_MIN", 0), _lib.knob("TBK_QL16_SPLIT_MIN", 0):
        w = tb.wf_array(m, mesh)
        gaps = w.solve_on_grid(start)
        host = w.to_host().copy()
        with _lib.knob("TBK_E16_CELLS", 0):
            w0 = tb.wf_array(m, mesh)
            gaps0 = w0.solve_on_grid(start)
            host0 = w0.to_host().copy()
        sub = [max(2, s - 2) for s in mesh[:-1]] + [max(2, mesh[-1] - 5)]
        lo = [1] * (D - 1) + [min(2, mesh[-1] - sub[-1])]
        ww = tb.wf_array(m, sub)
        ww.solve_on_grid_window(start, lo, mesh)
        assert np.array_equal(ww.to_host(), host[tuple(slice(a, a + b) for a, b in zip(lo, sub))])
    assert np.max(np.abs(gaps - gaps0)) < 1e-13
    assert not np.array_equal(host, host0)                                                     # (another order of the terms: the cells were used)
    idx = np.stack(np.meshgrid(*[np.arange(s - 1) for s in mesh], indexing="ij"), axis=-1).reshape(-1, D)
    k = np.array(start) + idx / (np.array(mesh) - 1.0)
    sel = np.arange(0, len(k), max(1, len(k) // 300))
    V = host[tuple(idx[sel].T)]                                                                # [point][band][component]
    V0 = host0[tuple(idx[sel].T)]
    H = np.array([m._gen_ham(kk).reshape(n, n) for kk in k[sel]])
    ev = np.linalg.eigvalsh(H)
    e_mesh = np.einsum("kbi,kij,kbj->kb", V.conj(), H, V).real
    assert np.max(np.abs(e_mesh - ev)) < 1e-12
    assert np.max(np.abs(np.einsum("kij,kbj->kbi", H, V) - e_mesh[:, :, None] * V)) < 1e-12
    assert np.max(np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.identity(n))) < 1e-13
    ov = np.abs(np.einsum("kbi,kbi->kb", V.conj(), V0))
    gapmin = np.minimum(np.diff(ev, axis=1, prepend=-np.inf), np.diff(ev, axis=1, append=np.inf))
    assert np.all(ov[gapmin > 1e-6] > 1.0 - 1e-9)


@pytest.mark.parametrize("n,rmax,mesh", [(5, 1, [5, 70]), (6, 2, [3, 4, 65]), (8, 3, [7, 129]), (8, 4, [66]), (7, 2, [4, 64])])
def test_mesh_rows_of_5_to_8_states_from_coefficient_cells(tb, n, rmax, mesh):
    """round 5: k_solve_regd on a mesh whose last axis holds >= 64 points sums the row's coefficient cells (reg_assemble_cells: two rows
    per wavefront, lane = slot) instead of every lattice vector per point (TBK_REG_CELLS=0).  Against the reference Hamiltonian of
    every point, against the other form, and window against whole array bit for bit (the order of the terms depends on the model and
    the mesh alone)."""
    from pythtb_amd import _lib
    D = len(mesh)
    m = hp.random_model(tb.tb_model, n, D, 1, seed=1300 + 10 * n + rmax, nhop=7 * n, rmax=rmax)
    R = [0] * (D - 1) + [rmax]
    if D > 1:
        R[0] = 1
    m.set_hop(0.17 + 0.11j, 0, n - 1, R, mode="add", allow_conjugate_pair=True)               # (the range is reached for sure)
    start = [0.1, -0.3, 0.2][:D]
    w = tb.wf_array(m, mesh)
    gaps = w.solve_on_grid(start)
    host = w.to_host().copy()
    with _lib.knob("TBK_REG_CELLS", 0):
        w0 = tb.wf_array(m, mesh)
        gaps0 = w0.solve_on_grid(start)
        host0 = w0.to_host().copy()
    assert np.max(np.abs(gaps - gaps0)) < 1e-13
    idx = np.stack(np.meshgrid(*[np.arange(s - 1) for s in mesh], indexing="ij"), axis=-1).reshape(-1, D)
    k = np.array(start) + idx / (np.array(mesh) - 1.0)
    sel = np.arange(0, len(k), max(1, len(k) // 400))
    V = host[tuple(idx[sel].T)]                                                                # [point][band][component]
    V0 = host0[tuple(idx[sel].T)]
    H = np.array([m._gen_ham(kk).reshape(n, n) for kk in k[sel]])
    ev = np.linalg.eigvalsh(H)
    e_mesh = np.einsum("kbi,kij,kbj->kb", V.conj(), H, V).real
    assert np.max(np.abs(e_mesh - ev)) < 1e-12
    assert np.max(np.abs(np.einsum("kij,kbj->kbi", H, V) - e_mesh[:, :, None] * V)) < 1e-12
    assert np.max(np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.identity(n))) < 1e-13
    ov = np.abs(np.einsum("kbi,kbi->kb", V.conj(), V0))
    gapmin = np.minimum(np.diff(ev, axis=1, prepend=-np.inf), np.diff(ev, axis=1, append=np.inf))
    assert np.all(ov[gapmin > 1e-6] > 1.0 - 1e-9)
    assert not np.array_equal(host, host0)                                                     # (the cells were used: another order of the terms)
    # a window that starts inside a row block of 64 points and ends before the last row: the same bits as the whole array
    lo = [1] * (D - 1) + [3]
    sub = [max(2, s - 2) for s in mesh[:-1]] + [mesh[-1] - 3]
    ww = tb.wf_array(m, sub)
    ww.solve_on_grid_window(start, lo, mesh)
    assert np.array_equal(ww.to_host(), host[tuple(slice(a, a + b) for a, b in zip(lo, sub))])


@pytest.mark.parametrize("n,nocc,mesh", [(6, 3, [23, 37]), (8, 4, [70, 19]), (8, 3, [5, 6, 41]), (9, 4, [12, 11, 9]),
                                         (6, 3, [131, 70]), (8, 4, [67, 200]), (5, 3, [3, 150]), (7, 4, [140, 3]), (4, 4, [9, 9, 70]),
                                         (10, 3, [66, 5, 7]), (6, 5, [40, 70]), (7, 6, [70, 9]), (10, 6, [9, 8, 66]), (12, 5, [131, 6])])
def test_wilson_loops_of_3_and_4_bands_three_routes(tb, n, nocc, mesh):
    """Wilson-loop eigenphases (berry_phase(..., berry_evals=True), pythtb.py:3798-3838) of 3 and 4 bands: a lane per string or per
    link with the occupied vectors staged through LDS (tbk_berry_lanes.inl: TBK_WILSON_REG=3, the default since round 6), the
    string's links, their polar factors and their ordered product in a thread per segment (k_wilson_seg_reg, =1) and the
    workgroup-per-link kernels (=0): the same phases along
    every direction, on string counts and lengths that are no multiples of the wavefront, the tile or the segment -- strings
    across the lanes (S form, contiguous and gathered tiles, ragged last tile), strings along the lanes (L form: the fastest
    axis, several 64-link tiles, few strings), short strings, the occupied bands not the lowest."""
    from pythtb_amd import _lib
    m = hp.random_model(tb.tb_model, n, len(mesh), 1, seed=70 + n + nocc, nhop=4 * n, rmax=1)
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.03, -0.2, 0.1][:len(mesh)])
    occ = list(range(nocc)) if n % 2 == 0 else list(range(n - nocc, n))[::-1]
    for d in range(len(mesh)):
        got = {}
        for route in (1, 0, 3):
            with _lib.knob("TBK_WILSON_REG", route), _lib.knob("TBK_WILSON_MFMA", 2):     # (MFMA=2: wide states stay off the tile kernel)
                got[route] = np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True))
        got["default"] = np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True))
        for route in (0, 3, "default"):
            diff = np.angle(np.exp(1j * (got[1] - got[route])))
            assert np.max(np.abs(diff)) < 1e-10, (d, route)
        # the sum of the eigenphases is the determinant form's phase
        det = np.asarray(w.berry_phase(occ, d, contin=False))
        assert np.max(np.abs(np.angle(np.exp(1j * (got[1].sum(axis=-1) - det))))) < 1e-9


@pytest.mark.parametrize("nocc,mesh,d", [(3, [40, 70], 0), (4, [9, 90], 1)])
def test_wilson_lanes_cayley_pole_is_noticed_and_redone(tb, nocc, mesh, d):
    """The lanes route forms the Cayley transform of the FIRST angle inside its combine kernel (one launch less); the transform has
    a pole at theta = alpha + pi.  With the pole put exactly on one string's eigenphase the call must notice (max |h|), redo
    that string with the next angle through k_wilson_cayley, and return the same phases."""
    from pythtb_amd import _lib
    m = hp.random_model(tb.tb_model, 2 * nocc, 2, 1, seed=91 + nocc, nhop=8 * nocc, rmax=1)
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.05, 0.15])
    occ = list(range(nocc))
    got = np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True))
    ctx = _lib.default_context()
    ctx.prof_enable(1)
    ctx.prof_reset()
    try:
        with _lib.knob("TBK_WILSON_ALPHA", repr(float(-got[5, 1] - np.pi))):
            again = np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True))
        rep = ctx.prof_report()
    finally:
        ctx.prof_enable(0)
    assert rep["wilson_lanes_combine"]["launches"] == 1 and rep["wilson_cayley"]["launches"] == 1     # first angle fused, second alone
    assert np.max(np.abs(np.angle(np.exp(1j * (np.sort(again, axis=1) - np.sort(got, axis=1)))))) < 1e-10


@pytest.mark.parametrize("n,nocc,mesh", [(16, 8, [9, 7, 40]), (16, 5, [6, 33, 5]), (12, 7, [21, 19]), (10, 6, [5, 4, 70])])
def test_wilson_loops_of_5_to_8_wide_bands_on_the_matrix_cores(tb, n, nocc, mesh):
    """Wilson-loop eigenphases of 5..8 bands of states with >= 8 components: k_chain_prod_tile<.., POLAR> forms a string's link
    matrices, iterates each to its polar factor (Newton-Schulz on v_mfma_f64_16x16x4_f64, X and X^T both in the accumulator
    layout) and multiplies the factors in order, one wavefront per (string, segment); against the workgroup-per-link kernels
    (TBK_WILSON_MFMA=0) along every direction, and the sum of the eigenphases against the determinant form."""
    from pythtb_amd import _lib
    m = hp.random_model(tb.tb_model, n, len(mesh), 1, seed=300 + n + nocc, nhop=4 * n, rmax=1)
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.02, 0.11, -0.3][:len(mesh)])
    occ = list(range(nocc))
    for d in range(len(mesh)):
        with _lib.knob("TBK_WILSON_MFMA", 1):
            a = np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True))
        with _lib.knob("TBK_WILSON_MFMA", 0):
            b = np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True))
        assert np.max(np.abs(np.angle(np.exp(1j * (a - b))))) < 1e-9, d
        det = np.asarray(w.berry_phase(occ, d, contin=False))
        assert np.max(np.abs(np.angle(np.exp(1j * (a.sum(axis=-1) - det))))) < 1e-9, d


@pytest.mark.parametrize("n,nspin,occ,mesh", [(2, 1, [0], [9, 8, 70]), (2, 2, [0, 1], [6, 7, 33]), (3, 1, [0, 2], [5, 9, 17]),
                                              (2, 2, [1, 2, 3], [4, 5, 6, 40]), (1, 1, [0], [7, 6, 130])])
def test_flux_of_planes_without_the_fastest_axis(tb, n, nspin, occ, mesh):
    """berry_flux on planes that do not contain the fastest mesh axis takes k_flux_slices (lane = slice, the vectors of 64
    neighbouring planes contiguous in memory): against the row kernel (TBK_FLUX_SLICES=0) for every such pair of directions in
    both orders, on slice counts that are no multiple of the wavefront; planes that do contain the fastest axis are untouched."""
    from pythtb_amd import _lib
    d = len(mesh)
    m = hp.random_model(tb.tb_model, n, d, nspin, seed=11 * n + d, nhop=4 * n, rmax=1)
    w = tb.wf_array(m, mesh)
    w.solve_on_grid([0.01 * (i + 1) for i in range(d)])
    for d0 in range(d - 1):
        for d1 in range(d - 1):
            if d0 == d1:
                continue
            a = np.asarray(w.berry_flux(occ, [d0, d1]))
            with _lib.knob("TBK_FLUX_SLICES", 0):
                b = np.asarray(w.berry_flux(occ, [d0, d1]))
            assert a.shape == b.shape and np.max(np.abs(a - b)) < 1e-11 * max(1.0, np.sqrt(mesh[d0] * mesh[d1])), (d0, d1)
            ind = np.asarray(w.berry_flux(occ, [d0, d1], individual_phases=True))
            assert np.max(np.abs(ind.sum(axis=(-2, -1)) - a)) < 1e-10


# ------------------------------------------ round 5: the direct solvers of 3..8 states on matrices with special structure
def _special_hermitian(n, rng):
    """Supplied matrices that stress the direct small solvers (tridiag_small / ql_deflate_small, the closed-form first-sweep shifts of
    n = 3, 4, k_solve_regd): generic, multiples of the identity, diagonal, block-diagonal (T splits), every level double (Kramers-like),
    nearly double (split by 1e-12 and 1e-7), clustered, graded over 12 orders of magnitude, one huge scale, tiny scale, zero matrix."""
    def conj_by_random_unitary(lev):
        u = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
        a = (u * lev) @ u.conj().T
        return 0.5 * (a + a.conj().T)
    out = []
    for _ in range(40):
        a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        out.append(0.5 * (a + a.conj().T))
    out.append(np.identity(n) * 1.75)
    out.append(np.zeros((n, n)))
    out.append(np.diag(rng.standard_normal(n)).astype(complex))
    blk = np.zeros((n, n), dtype=complex)
    h = n // 2
    for lo, hi in ((0, h), (h, n)):
        a = rng.standard_normal((hi - lo, hi - lo)) + 1j * rng.standard_normal((hi - lo, hi - lo))
        blk[lo:hi, lo:hi] = 0.5 * (a + a.conj().T)
    out.append(blk)
    for split in (0.0, 1e-12, 1e-7):
        lev = np.repeat(np.sort(rng.standard_normal((n + 1) // 2)), 2)[:n] + np.tile([0.0, split], (n + 1) // 2)[:n]
        for _ in range(6):
            out.append(conj_by_random_unitary(lev))
    out.append(conj_by_random_unitary(1.0 + 1e-9 * rng.standard_normal(n)))            # one tight cluster
    out.append(conj_by_random_unitary(10.0 ** np.linspace(-6, 6, n)))                   # graded
    out.append(conj_by_random_unitary(rng.standard_normal(n)) * 1e150)
    out.append(conj_by_random_unitary(rng.standard_normal(n)) * 1e-150)
    t = np.diag(rng.standard_normal(n)).astype(complex)                                 # already tridiagonal, real couplings
    for i in range(n - 1):
        t[i, i + 1] = t[i + 1, i] = rng.standard_normal()
    out.append(t)
    return np.array(out)


@pytest.mark.parametrize("n", [3, 4, 5, 6, 7, 8])
def test_direct_small_solvers_on_special_matrices(tb, n):
    """_sol_ham-type calls (pythtb.py:927-953) through tbk_eigh_batch, eigenvalues alone and with eigenvectors; for 5..8 states also the
    register-Jacobi kernel the direct method replaced (TBK_REG_DIRECT=0) as a second opinion."""
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    h = _special_hermitian(n, np.random.default_rng(700 + n))
    nk = len(h)
    ref = np.linalg.eigvalsh(h)
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-300)

    def run(with_vec):
        ev = np.zeros((n, nk))
        vec = np.zeros((n, nk, n), dtype=complex) if with_vec else None
        hc = np.ascontiguousarray(h)
        _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), nk, _lib.dptr(ev),
                                           _lib.dptr(vec.view(float)) if with_vec else None))
        return ev, vec
    ev0, _ = run(False)
    assert np.all(np.diff(ev0, axis=0) >= 0.0)
    assert np.max(np.abs(ev0.T - ref) / scale[:, None]) < 4e-15
    variants = [("direct", None)] + ([("jacobi", 0)] if n >= 5 else [])
    for name, knob in variants:
        if knob is None:
            ev, vec = run(True)
        else:
            with _lib.knob("TBK_REG_DIRECT", knob):
                ev, vec = run(True)
        V = vec.transpose(1, 0, 2)                      # [k][band][comp]
        assert np.max(np.abs(ev.T - ref) / scale[:, None]) < 4e-15, name
        res = np.abs(np.einsum("kij,kbj->kbi", h, V) - V * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / scale
        orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).reshape(nk, -1).max(axis=1)
        # (the Jacobi kernel's convergence test squares the entries: on the 1e-150 matrix it stops at a residual of 5e-13 |H| -- the
        # direct method, which works on norms with rsqrt, does not have that cliff)
        assert res.max() < (1e-14 if name == "direct" else 2e-12) and orth.max() < 1e-14, (name, res.max(), orth.max())
