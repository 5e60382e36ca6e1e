"""solve_on_grid + berry_flux in one pass (tbk_wfs_solve_grid_flux_async / wf_array.solve_on_grid_flux, tbk_solve_fused.inl)
against the two calls it replaces (pythtb.py:2421-2532 then :3068-3205) and against the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


CASES = [
    ("haldane", [65, 65], [0]), ("haldane", [130, 77], [1]), ("haldane", [7, 300], [0]), ("haldane", [300, 5], [0]),
    ("haldane", [2, 2], [0]), ("haldane", [33, 64], [0, 1]), ("haldane", [33, 129], [0]), ("haldane", [129, 257], [0]),
    ("kane_mele", [40, 70], [0, 1]), ("kane_mele", [70, 200], [2]), ("kane_mele", [9, 513], [1, 3]),
]


def _model(tb, name):
    return hp.haldane(tb.tb_model, 0.0) if name == "haldane" else hp.kane_mele(tb.tb_model, "odd")


@pytest.mark.parametrize("name,mesh,occ", CASES, ids=["%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], "".join(map(str, c[2]))) for c in CASES])
def test_fused_equals_the_two_calls(tb, name, mesh, occ):
    from pythtb_amd import _lib
    m = _model(tb, name)
    start = [-0.5, 0.13]
    w1 = tb.wf_array(m, mesh)
    g1 = w1.solve_on_grid(start)
    f1 = w1.berry_flux(occ)
    w2 = tb.wf_array(m, mesh)
    g2, f2 = w2.solve_on_grid_flux(start, occ)
    nplaq = (mesh[0] - 1) * (mesh[1] - 1)
    assert np.array_equal(g1, g2)
    assert abs(f1 - f2) < 1e-12 * max(1.0, nplaq ** 0.5)
    a1, a2 = w1.to_host(), w2.to_host()
    assert np.max(np.abs(a1 - a2)) < 2e-15              # two kernels: the compiler contracts a b - c d differently (1 ulp)
    # what a later berry_flux reads back from the fused array is what the fused pass summed
    assert abs(w2.berry_flux(occ) - f2) < 1e-12 * max(1.0, nplaq ** 0.5)
    # the fused kernel's own bits do not depend on its tiling, and a second run repeats them
    for rows, seg in ((1, 1), (2, 3), (7, 2)):
        with _lib.knob("TBK_FUSED_ROWS", rows), _lib.knob("TBK_GRID_SEG", seg):
            w3 = tb.wf_array(m, mesh)
            g3, f3 = w3.solve_on_grid_flux(start, occ)
            assert np.array_equal(a2, w3.to_host()) and np.array_equal(g3, g2) and abs(f3 - f2) < 1e-12 * max(1.0, nplaq ** 0.5)
    g4, f4 = w2.solve_on_grid_flux(start, occ)
    assert f4 == f2 and np.array_equal(g4, g2)


def test_fused_against_the_oracle_and_chern_number(tb):
    from oracle import tb_oracle as orc
    m = hp.haldane(tb.tb_model, 0.0)
    mesh, start = [49, 38], [-0.5, -0.5]
    w = tb.wf_array(m, mesh)
    gaps, flux = w.solve_on_grid_flux(start, [0])
    owfs, ogaps = orc.solve_on_grid(m, mesh, start, vectorised=True)
    oflux = orc.berry_flux(owfs, 2, [0], vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-12 and abs(flux - oflux) < 1e-10
    assert abs(flux / (2 * np.pi) + 1.0) < 1e-10
    w = tb.wf_array(m, [2049, 2049])                         # BASELINE configs[2] at full size
    gaps, flux = w.solve_on_grid_flux(start, [0])
    assert abs(flux / (2 * np.pi) + 1.0) < 1e-10 and abs(gaps[0] - 1.55884812) < 1e-7


def test_fused_slabs_add_up(tb):
    """k-slabs along axis 0 (bench.py --gpus N): every slab runs the fused pass on its rows [row0, row0 + n) of the global
    mesh, halo row recomputed; the partial fluxes add up to the unsharded flux and the min gaps to the global minimum."""
    from pythtb_amd import _lib, shard
    lib = _lib.lib
    m = hp.haldane(tb.tb_model, 0.0)
    g_n0, n1, start = 61, 45, np.array([-0.5, -0.5])
    full = tb.wf_array(m, [g_n0, n1])
    gaps, flux = full.solve_on_grid_flux(start, [0])
    occ = np.array([0], dtype=np.int32)
    pbc = np.ascontiguousarray(np.array([np.exp(-2j * np.pi * m._orb[:, m._per[d]]) for d in range(2)]))
    for world in (2, 3):
        tot, gmin = 0.0, np.inf
        for r in range(world):
            row0, nrows = shard.split_rows(g_n0, world, r)
            w = tb.wf_array(m, [nrows, n1])
            h = w._dev_handle(w._shape())
            _lib.check(lib.tbk_wfs_solve_grid_flux_async(h, m._device_model(), _lib.dptr(start), _lib.dptr(pbc.view(float)), row0, g_n0,
                                                         _lib.iptr(occ), 1))
            g = np.zeros(1)
            _lib.check(lib.tbk_wfs_solve_grid_result(h, _lib.dptr(g)))
            t = np.zeros(1)
            _lib.check(lib.tbk_berry_flux_result(h, _lib.dptr(t), None))
            tot += t[0]
            gmin = min(gmin, g[0])
        assert abs(tot - flux) < 1e-11 and gmin == gaps[0]


def test_fused_falls_back_where_it_does_not_apply(tb):
    """3-D arrays, other state counts, more than two bands: the Python method makes the two calls; the C entry says so."""
    from pythtb_amd import _lib
    m3 = hp.random_model(tb.tb_model, 3, 2, 1, 4)
    w = tb.wf_array(m3, [12, 9])
    gaps, flux = w.solve_on_grid_flux([0.1, 0.2], [0])
    w0 = tb.wf_array(m3, [12, 9])
    g0 = w0.solve_on_grid([0.1, 0.2])
    assert np.array_equal(gaps, g0) and flux == w0.berry_flux([0])
    km = hp.kane_mele(tb.tb_model, "odd")
    w = tb.wf_array(km, [12, 9])
    gaps, flux = w.solve_on_grid_flux([0.1, 0.2], [0, 1, 2])
    assert abs(flux - w.berry_flux([0, 1, 2])) < 1e-12
    h = w._dev_handle(w._shape())
    occ = np.array([0, 1, 2], dtype=np.int32)
    pbc = np.zeros((2, 4), dtype=complex)
    st = np.zeros(2)
    rc = _lib.lib.tbk_wfs_solve_grid_flux_async(h, km._device_model(), _lib.dptr(st), _lib.dptr(pbc.view(float)), 0, 12, _lib.iptr(occ), 3)
    assert rc == 4                                           # TBK_EUNSUPPORTED
