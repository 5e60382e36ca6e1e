"""Every TBK_* knob that selects a path or a launch shape is forced at least once against the default path on the same inputs
("every remaining knob appears in at least one test's _lib.knob(...)").  The eigen-solver knobs are rows of
tests/test_regimes.py's tables; here the Berry-side and transfer-side ones.  tests/test_knob_table_cpu (below, no GPU) checks the
claim itself: the set of names tbk_core.hip parses == the set of names the tests force."""
import os
import re

import numpy as np
import pytest

import helpers as hp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def wrap(d):
    return (np.asarray(d) + np.pi) % (2 * np.pi) - np.pi


def test_every_parsed_knob_is_forced_by_some_test():
    src = open(os.path.join(ROOT, "pythtb_amd", "csrc", "tbk_core.hip")).read()
    # (knobs inside #ifdef TBK_DIAG / TBK_REG_MULTILANE blocks exist in diagnostic builds only)
    prod = re.sub(r"#if(?:def)? +(?:defined\()?TBK_(?:DIAG|REG_MULTILANE)\)?.*?#endif", "", src, flags=re.S)
    parsed = set(re.findall(r'get[ild]\("(TBK_[A-Z0-9_]+)"', prod))
    assert len(parsed) > 40
    forced = set()
    for f in os.listdir(os.path.join(ROOT, "tests")):
        if f.endswith(".py"):
            forced |= set(re.findall(r'"(TBK_[A-Z0-9_]+)"', open(os.path.join(ROOT, "tests", f)).read()))
    assert not sorted(parsed - forced), "knobs no test forces: %s" % sorted(parsed - forced)


gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    pythtb_amd._lib.default_context()
    return pythtb_amd


FLUX_KNOBS = [("TBK_FLUX_TI", 4), ("TBK_FLUX_TI", 24), ("TBK_FLUX_ORDER", 0), ("TBK_FLUX_ORDER", 1), ("TBK_FLUX_FUSED", 1),
              ("TBK_GRID_OCC", 2), ("TBK_GRID_OCC", 4), ("TBK_FUSED_OCC", 2), ("TBK_FUSED_SUM", 0)]


@gpu
@pytest.mark.parametrize("knob,value", FLUX_KNOBS, ids=["%s=%s" % kv for kv in FLUX_KNOBS])
def test_mesh_and_flux_launch_shapes(tb, knob, value):
    """Tile heights, tile order, the in-kernel sum, occupancy caps: the same plaquettes, totals, gaps and (bit for bit) vectors."""
    from pythtb_amd import _lib
    for m, mesh, occ in ((hp.haldane(tb.tb_model, 0.3), [131, 200], [0]), (hp.kane_mele(tb.tb_model, "odd"), [70, 97], [0, 1])):
        out = {}
        for kv in (None, (knob, value)):
            with (_lib.knob(*kv) if kv else _lib.knob("TBK_FLUX_TI", -1)):
                w = tb.wf_array(m, mesh)
                gaps = w.solve_on_grid([0.11, -0.23])
                plaq = w.berry_flux(occ, individual_phases=True)
                tot = w.berry_flux(occ)
                w2 = tb.wf_array(m, mesh)
                g2, f2 = w2.solve_on_grid_flux([0.11, -0.23], occ=occ)
                out[kv is not None] = (np.array(gaps), plaq, tot, w.to_host().copy(), np.array(g2), f2)
        a, b = out[False], out[True]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        assert abs(a[2] - b[2]) < 1e-10 and abs(a[5] - b[5]) < 1e-10 and abs(a[2] - a[5]) < 1e-9


CHAIN_KNOBS = [("TBK_WILSON_REG", 1), ("TBK_CHAIN_WAVE", 0), ("TBK_CHAIN_WAVE_FROM", 3), ("TBK_CHAIN_WS_MB", 1), ("TBK_DET_BIG_FROM", 2), ("TBK_DET_BIG_FROM", 4),
               ("TBK_WILSON_BIG_FROM", 2), ("TBK_WILSON_BIG_FROM", 5), ("TBK_WILSON_FORM", 0), ("TBK_WILSON_FORM", 1),
               ("TBK_WILSON_SEG", 1), ("TBK_WILSON_SEG", 5), ("TBK_WILSON_SWZ", 0)]


@gpu
@pytest.mark.parametrize("knob,value", CHAIN_KNOBS, ids=["%s=%s" % kv for kv in CHAIN_KNOBS])
def test_berry_phase_route_knobs(tb, knob, value):
    """Link determinants and Wilson-loop eigenphases of 2..4 bands of 8- and 16-component states by the route / shape the knob
    forces against the default route: every direction, determinant and eigenphase form."""
    from pythtb_amd import _lib
    cases = ((hp.cubic16(tb.tb_model), [7, 6, 40], [0.1, 0.2, 0.3]),
             (hp.random_model(tb.tb_model, 8, 2, 1, seed=77, nhop=24, rmax=1), [70, 45], [0.05, -0.1]),
             # narrow states (fewer than 8 components): the determinant form of 1..4 bands on the LDS-tile kernels (round 6)
             (hp.random_model(tb.tb_model, 6, 2, 1, seed=78, nhop=20, rmax=1), [131, 45], [0.02, 0.3]),
             (hp.random_model(tb.tb_model, 5, 3, 1, seed=79, nhop=15, rmax=1), [9, 8, 70], [0.1, 0.0, -0.2]))
    for m, mesh, start in cases:
        w = tb.wf_array(m, mesh)
        w.solve_on_grid(start)
        for occ in ([0], [0, 1], [0, 1, 2], [1, 2, 3, 4]):
            for d in range(len(mesh)):
                ref_det = np.asarray(w.berry_phase(occ, d, contin=False))
                ref_ev = np.sort(np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True)), axis=-1)
                with _lib.knob(knob, value):
                    det = np.asarray(w.berry_phase(occ, d, contin=False))
                    ev = np.sort(np.asarray(w.berry_phase(occ, d, contin=False, berry_evals=True)), axis=-1)
                assert np.max(np.abs(wrap(det - ref_det))) < 1e-10, (occ, d)
                # (sorted phases near +-pi may swap ends: compare as sets through the sum and the sorted cosines)
                assert np.max(np.abs(wrap(ev.sum(axis=-1) - ref_ev.sum(axis=-1)))) < 1e-9, (occ, d)
                assert np.max(np.abs(np.sort(np.cos(ev), axis=-1) - np.sort(np.cos(ref_ev), axis=-1))) < 1e-9, (occ, d)


@gpu
def test_solve_all_through_copies_instead_of_mapped_memory(tb):
    """TBK_ZERO_COPY_KB=0: the k list and the results of a small solve_all call cross PCIe as copies, not through mapped host
    memory -- the same bits."""
    from pythtb_amd import _lib
    m = hp.kane_mele(tb.tb_model, "odd")
    k = np.random.default_rng(2).random((700, 2))
    ev, vec = m.solve_all(k, eig_vectors=True)
    with _lib.knob("TBK_ZERO_COPY_KB", 0):
        ev0, vec0 = m.solve_all(k, eig_vectors=True)
        one = m.solve_one(k[5])
    assert np.array_equal(ev, ev0) and np.array_equal(vec, vec0) and np.array_equal(one, ev[:, 5])
