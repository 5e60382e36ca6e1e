"""k generators on the device and the DOS histogram of the reference's
examples/haldane.py:96-121 computed without moving eigenvalues to the host."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
import helpers as hp
from oracle import tb_oracle as orc

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "tests", "golden", "reference_tests")


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def _download(lib, ctx, ptr, shape):
    out = np.zeros(shape)
    from pythtb_amd import _lib
    _lib.check(lib.tbk_dev_download(ctx.handle, out.ctypes.data_as(C.c_void_p), ptr, out.nbytes))
    return out


@pytest.mark.parametrize("mesh", [[7], [4, 6], [3, 4, 5], [1, 9], [33, 1, 2]])
def test_k_uniform_mesh_on_device_is_bit_equal(tb, mesh):
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    d = len(mesh)
    m = hp.quiet(tb.tb_model, d, d, np.identity(d), [[0.0] * d])
    want = m.k_uniform_mesh(mesh)
    nk = want.shape[0]
    ptr = C.c_void_p()
    _lib.check(lib.tbk_dev_alloc(ctx.handle, want.nbytes, C.byref(ptr)))
    try:
        _lib.check(lib.tbk_k_uniform_mesh_dev(ctx.handle, d, _lib.iptr(np.array(mesh, dtype=np.int32)), ptr))
        got = _download(lib, ctx, ptr, (nk, d))
    finally:
        _lib.check(lib.tbk_dev_free(ctx.handle, ptr))
    assert np.array_equal(got, want)


def test_k_path_on_device_is_bit_equal(tb):
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    g = hp.graphene(tb.tb_model)
    cases = [(g, [[0.0, 0.0], [2.0 / 3.0, 1.0 / 3.0], [0.5, 0.5], [0.0, 0.0]], 121),
             (g, [[0.0, 0.0], [0.31, -0.77], [0.5, 0.5]], 37),
             (hp.chain3(tb.tb_model, -1.0, 2.0, 0.3), [[-0.5], [0.5]], 31),
             (hp.cubic16(tb.tb_model), [[0, 0, 0], [0.5, 0, 0], [0.5, 0.5, 0], [0.5, 0.5, 0.5], [0, 0, 0]], 203)]
    for model, nodes, nk in cases:
        k_vec, k_dist, k_node = model.k_path(nodes, nk, report=False)
        # node indices as the host code derives them (pythtb.py:1971-1976)
        idx = [0] + [int(round(k_node[s] / k_node[-1] * (nk - 1))) for s in range(1, len(nodes) - 1)] + [nk - 1]
        nd = np.ascontiguousarray(np.array(nodes, dtype=float))
        d = nd.shape[1]
        ptr = C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, nk * d * 8, C.byref(ptr)))
        try:
            _lib.check(lib.tbk_k_path_dev(ctx.handle, d, len(nodes), _lib.dptr(nd), _lib.iptr(np.array(idx, dtype=np.int32)),
                                          nk, ptr))
            got = _download(lib, ctx, ptr, (nk, d))
        finally:
            _lib.check(lib.tbk_dev_free(ctx.handle, ptr))
        assert np.array_equal(got, k_vec)
    with pytest.raises(_lib.TbkError):
        bad = np.array([0, 5, 5, 9], dtype=np.int32)
        lib_ptr = C.c_void_p()
        _lib.check(lib.tbk_dev_alloc(ctx.handle, 10 * 2 * 8, C.byref(lib_ptr)))
        try:
            _lib.check(lib.tbk_k_path_dev(ctx.handle, 2, 4, _lib.dptr(np.zeros((4, 2))), _lib.iptr(bad), 10, lib_ptr))
        finally:
            _lib.check(lib.tbk_dev_free(ctx.handle, lib_ptr))


@pytest.mark.parametrize("builder,mesh", [("haldane", [20, 20]), ("kane_mele", [9, 14]), ("chain3", [57]), ("cubic16", [3, 4, 5])])
def test_solve_all_mesh_equals_solve_all_of_host_mesh(tb, builder, mesh):
    m = {"haldane": lambda: hp.haldane(tb.tb_model, 0.2), "kane_mele": lambda: hp.kane_mele(tb.tb_model, "odd"),
         "chain3": lambda: hp.chain3(tb.tb_model, -1.0, 2.0, 0.3), "cubic16": lambda: hp.cubic16(tb.tb_model)}[builder]()
    k = m.k_uniform_mesh(mesh)
    ev_list, vec_list = m.solve_all(k, eig_vectors=True)
    ev, vec = m.solve_all_mesh(mesh, eig_vectors=True)
    assert ev.shape == ev_list.shape and vec.shape == vec_list.shape
    assert np.array_equal(ev, ev_list)          # same kernel, same k bits -> same bits
    assert np.array_equal(vec, vec_list)
    # eigenvalues alone of up to 4 states: the row kernel k_mesh_evals (separable phases, no list) -- equal to rounding
    evm, evl = m.solve_all_mesh(mesh), m.solve_all(k)
    assert np.max(np.abs(evm - evl)) < 1e-13 and (m._nsta <= 4 or np.array_equal(evm, evl))
    ref = orc.solve_all(orc.Model.from_tables(orc.model_tables(m)), k)
    assert np.max(np.abs(ev - ref)) < 1e-12 * max(1.0, np.abs(ref).max())


def test_dos_reference_example_golden(tb):
    """examples/haldane.py:96-121 (tests/test_examples/haldane/haldane/run.py): 20x20 mesh, hist(evals, 50, (-4,4))."""
    m = hp.haldane(tb.tb_model, 0.2)
    evals_ref = np.load(os.path.join(REF, "haldane", "evals_dos.npy"))          # the reference's flattened eigenvalues
    want, want_edges = np.histogram(evals_ref, 50, range=(-4.0, 4.0))
    got, edges = m.dos_mesh([20, 20], 50, range=(-4.0, 4.0))
    assert got.dtype == np.int64 and got.shape == (50,)
    assert np.array_equal(edges, want_edges)
    assert got.sum() == 800
    assert np.abs(got - want).sum() <= 2          # an eigenvalue within 1e-15 of an edge may change bins
    per_band, _ = m.dos_mesh([20, 20], 50, range=(-4.0, 4.0), per_band=True)
    assert per_band.shape == (2, 50) and np.array_equal(per_band.sum(axis=0), got)


@pytest.mark.parametrize("builder,mesh,bins", [("haldane", [64, 48], 37), ("kane_mele", [31, 17], 64), ("cubic16", [6, 5, 4], 200),
                                               ("chain3", [1000], 8192)])
def test_dos_matches_numpy_histogram_of_device_eigenvalues(tb, builder, mesh, bins):
    m = {"haldane": lambda: hp.haldane(tb.tb_model, 0.0), "kane_mele": lambda: hp.kane_mele(tb.tb_model, "even"),
         "chain3": lambda: hp.chain3(tb.tb_model, -1.3, 2.0, 0.1), "cubic16": lambda: hp.cubic16(tb.tb_model)}[builder]()
    ev = m.solve_all_mesh(mesh)
    # automatic range (min..max of all eigenvalues), bit-equal to numpy on the same values
    got, edges = m.dos_mesh(mesh, bins)
    want, want_edges = np.histogram(ev.flatten(), bins)
    assert np.array_equal(edges, want_edges)
    assert np.array_equal(got, want)
    # explicit range that cuts eigenvalues off on both sides; per band
    lo, hi = np.quantile(ev, 0.2), np.quantile(ev, 0.9)
    got_b, edges_b = m.dos_mesh(mesh, bins, range=(lo, hi), per_band=True)
    for b in range(ev.shape[0]):
        want_b, _ = np.histogram(ev[b], bins, range=(lo, hi))
        assert np.array_equal(got_b[b], want_b)
    # values exactly on edges (incl. the closed last bin): a flat band
    flat = hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[0.0]])
    flat.set_onsite([1.0])
    got_f, edges_f = flat.dos_mesh([10], 4, range=(0.0, 1.0))
    assert np.array_equal(got_f, np.histogram(np.ones(10), 4, range=(0.0, 1.0))[0])
    got_f, edges_f = flat.dos_mesh([10], 4)                       # empty range: numpy widens it by +-0.5
    w, we = np.histogram(np.ones(10), 4)
    assert np.array_equal(got_f, w) and np.array_equal(edges_f, we)


def test_dos_argument_checks(tb):
    m = hp.haldane(tb.tb_model, 0.2)
    with pytest.raises(Exception, match="Incorrect size"):
        hp.quiet(m.dos_mesh, [10], 5)
    with pytest.raises(Exception, match="positive non-zero"):
        m.solve_all_mesh([0, 4])
    with pytest.raises(Exception, match="bins"):
        m.dos_mesh([4, 4], 0)


def test_solve_all_solves_the_list_it_is_given_and_solve_all_mesh_uploads_nothing(tb):
    """configs[1]: `m.solve_all(m.k_uniform_mesh([N, N]))` (examples/haldane.py:96-100) is the LIST path on exactly the array
    passed -- a mesh modified through any alias (np.asarray, a view, .flat) gives the modified list's eigenvalues like the
    reference (pythtb.py:1047-1060) -- and the explicit extension solve_all_mesh generates the same list on
    the device: no 16-bytes-per-k upload, the same eigenvalues (bit for bit through the list kernel)."""
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    ctx = _lib.default_context()
    for m, mo, mesh in ((hp.haldane(tb.tb_model, 0.2), True, [300, 200]),
                        (hp.kane_mele(tb.tb_model, "odd"), True, [64, 48]),
                        (hp.chain3(tb.tb_model, -1.0, 2.0, 0.3), None, [501]), (hp.cubic16(tb.tb_model), None, [6, 5, 7])):
        k = m.k_uniform_mesh(mesh)
        assert type(k) is np.ndarray
        plain = np.array(k)
        m.solve_all(plain[:3])                             # (model tables on the device before the counters are read)
        ctx.prof_enable(1)
        ctx.prof_reset()
        ev_list = m.solve_all(k)
        names = set(ctx.prof_report())
        assert not any(nm.startswith("mesh_evals") or nm == "k_mesh" for nm in names), names     # the list path: no mesh kernels
        ctx.transfer_stats(reset=True)
        ctx.prof_reset()
        ev = m.solve_all_mesh(mesh)
        st = ctx.transfer_stats(reset=True)
        names = set(ctx.prof_report())
        ctx.prof_enable(0)
        assert st["h2d_bytes"] < 1024, st
        assert ("mesh_evals" in names) == (m._nsta <= 4) and not any(nm.startswith("solve_list") for nm in names) == (m._nsta <= 4), names
        # (eigenvalues alone of up to 4 states come from the row kernel k_mesh_evals -- the mesh's separable phases, no k list at
        # all: equal to the list kernel's to rounding; everything else is the list kernel on the generated list: the same bits)
        assert ev.shape == ev_list.shape and np.max(np.abs(ev - ev_list)) < 1e-13
        assert m._nsta <= 4 or np.array_equal(ev, ev_list)
        with _lib.knob("TBK_MESH_ROWS", 0):
            assert np.array_equal(m.solve_all_mesh(mesh), ev_list)
        ev2, vec2 = m.solve_all_mesh(mesh, eig_vectors=True)
        evl, vecl = m.solve_all(k, eig_vectors=True)
        assert np.array_equal(ev2, evl) and vec2.shape == vecl.shape and np.array_equal(vec2, vecl)
        # writes through aliases the array cannot see, at rows no sampling would have looked at: the modified list is what is solved
        nk = len(k)
        rows = [1, nk // 3 + 1, nk - 2]
        np.asarray(k)[rows[0]] += 0.25
        k.view(np.ndarray).reshape(-1)[rows[1] * k.shape[1]] -= 0.125
        memoryview(k)[rows[2], 0] = 0.3125
        plain[rows[0]] += 0.25
        plain[rows[1], 0] -= 0.125
        plain[rows[2], 0] = 0.3125
        ev3 = m.solve_all(k)
        assert np.array_equal(ev3, m.solve_all(plain))
        for r in rows:
            assert not np.array_equal(ev3[:, r], ev_list[:, r])
        if mo is not None:
            assert np.max(np.abs(ev3[:, rows] - orc.solve_all(orc.Model.from_tables(orc.model_tables(m)), plain[rows]))) < 1e-12
