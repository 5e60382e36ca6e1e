"""The reference's own model-equivalence tests (tests/test_tbmodel/test_dimr_dimk_different.py,
test_different_modes.py, test_spin.py), restated against this package: differently written but
physically identical models must give the same Berry phases, band energies and finite-cut
position expectations, here with every number coming from the device kernels."""
import numpy as np
import pytest

from helpers import quiet

pytestmark = pytest.mark.gpu

LAT3 = [[3.0, 0.1, 0.4], [0.1, 3.1, 1.2], [0.8, 0.2, 3.5]]
LAT3B = [[3.0, 0.1, 0.4], [0.8, 0.2, 3.5], [-0.1, -3.1, -1.2]]
LAT2 = [[3.0, 0.4], [0.8, 3.5]]


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def equivalent(tb, models, cut_dirs, occs):
    """The three comparisons of the reference's generic_test_of_models (test_dimr_dimk_different.py:37-80)."""
    phases = []
    for m, occ in zip(models, occs):
        w = tb.wf_array(m, [11, 11])
        w.solve_on_grid([-0.5, -0.5])
        phases.append(w.berry_phase(occ, 1, contin=True))
    for other in phases[1:]:
        assert np.all(np.isclose(phases[0], other))
    energies = [m.solve_one([0.123, 0.523]) for m in models]
    for other in energies[1:]:
        assert np.all(np.isclose(energies[0], other))
    centres = []
    for m, d in zip(models, cut_dirs):
        piece = m.cut_piece(4, d, glue_edgs=False)
        _, vec = piece.solve_one([0.214], eig_vectors=True)
        centres.append(piece.position_expectation(vec, d))
    for other in centres[1:]:
        assert np.all(np.isclose(centres[0], other))


def test_same_model_in_different_embeddings(tb):           # test_dimr_dimk_different.py:9-35
    m0 = quiet(tb.tb_model, 2, 3, LAT3, [[0.3, 0.1, 0.2], [0.1, 0.8, 0.3], [0.2, 0.3, 0.4]], per=[0, 2])
    m0.set_onsite([-2.3, 0.5, 0.1])
    m0.set_hop(0.24, 0, 1, [1, 0, 2])
    m0.set_hop(0.42, 0, 1, [3, 0, 2])
    m0.set_hop(-0.12, 1, 2, [2, 0, 3])
    m0.set_hop(-0.34, 2, 0, [-1, 0, 2])
    m1 = quiet(tb.tb_model, 2, 2, LAT2, [[0.3, 0.2], [0.1, 0.3], [0.2, 0.4]])
    m1.set_onsite([-2.3, 0.5, 0.1])
    m1.set_hop(0.24, 0, 1, [1, 2])
    m1.set_hop(0.42, 0, 1, [3, 2])
    m1.set_hop(-0.12, 1, 2, [2, 3])
    m1.set_hop(-0.34, 2, 0, [-1, 2])
    m2 = quiet(tb.tb_model, 2, 3, LAT3B, [[0.3, 0.2, 0.1], [0.1, 0.3, 0.8], [0.2, 0.4, 0.3]], per=[0, 1])
    m2.set_onsite([-2.3, 0.5, 0.1])
    m2.set_hop(0.24, 0, 1, [1, 2, 0])
    m2.set_hop(0.42, 0, 1, [3, 2, 0])
    m2.set_hop(-0.12, 1, 2, [2, 3, 0])
    m2.set_hop(-0.34, 2, 0, [-1, 2, 0])
    equivalent(tb, [m0, m1, m2], [2, 1, 1], [[0], [0], [0]])


def test_same_model_built_with_different_setter_modes(tb):   # test_different_modes.py:8-55
    amp = -0.34 + 0.3j
    m0 = quiet(tb.tb_model, 2, 3, LAT3, [[0.3, 0.1, 0.2], [0.1, 0.8, 0.3], [0.2, 0.3, 0.4]], per=[0, 2])
    m0.set_onsite([-2.3, 0.5, 0.1])
    m0.set_hop(0.24, 0, 1, [1, 0, 2], mode="set")
    m0.set_hop(0.42, 0, 1, [3, 0, 2])
    m0.set_hop(-0.12, 1, 2, [2, 0, 3])
    m0.set_hop(amp, 2, 0, [-1, 0, 2])
    m1 = quiet(tb.tb_model, 2, 2, LAT2, [[0.3, 0.2], [0.1, 0.3], [0.2, 0.4]])
    m1.set_onsite(-2.3, 0)
    m1.set_onsite(0.5, 1)
    m1.set_onsite(9.1, 2, mode="reset")
    m1.set_onsite(0.07, 2, mode="reset")
    m1.set_onsite(0.03, 2, mode="add")
    m1.set_hop(99.24, 0, 1, [1, 2], mode="set")
    m1.set_hop(0.04, 0, 1, [1, 2], mode="reset")
    m1.set_hop(0.08, 0, 1, [1, 2], mode="add")
    m1.set_hop(0.12, 0, 1, [1, 2], mode="add")
    m1.set_hop(0.42, 0, 1, [3, 2])
    m1.set_hop(-0.12, 1, 2, [2, 3])
    m1.set_hop(amp, 2, 0, [-1, 2])
    m2 = quiet(tb.tb_model, 2, 3, LAT3B, [[0.3, 0.2, 0.1], [0.1, 0.3, 0.8], [0.2, 0.4, 0.3]], per=[0, 1])
    m2.set_onsite([-2.3, 0.5, 0.1])
    m2.set_hop(0.24, 0, 1, [1, 2, 0])
    m2.set_hop(99.42, 0, 1, [3, 2, 0], mode="reset")
    m2.set_hop(0.42, 0, 1, [3, 2, 0], mode="reset")
    m2.set_hop(-0.12, 1, 2, [2, 3, 0])
    m2.set_hop(amp * 0.7, 2, 0, [-1, 2, 0], allow_conjugate_pair=True)
    m2.set_hop(amp.conjugate() * 0.3, 0, 2, [1, -2, 0], allow_conjugate_pair=True)
    equivalent(tb, [m0, m1, m2], [2, 1, 1], [[0], [0], [0]])


def test_spinor_model_equals_its_doubled_scalar_model(tb):    # test_spin.py:8-62
    orb6 = [[0.3, 0.1, 0.2]] * 2 + [[0.1, 0.8, 0.3]] * 2 + [[0.2, 0.3, 0.4]] * 2
    m0 = quiet(tb.tb_model, 2, 3, LAT3, orb6, nspin=1, per=[0, 2])
    m0.set_onsite([-2.3, -2.3, 0.5, 0.5, 0.1, 0.1])

    def both_spins(block, i, j, R):          # (up,up), (dn,dn), (up,dn), (dn,up) of orbital pair (i, j)
        m0.set_hop(block[0][0], 2 * i, 2 * j, R)
        m0.set_hop(block[1][1], 2 * i + 1, 2 * j + 1, R)
        if block[0][1] != 0:
            m0.set_hop(block[0][1], 2 * i, 2 * j + 1, R)
            m0.set_hop(block[1][0], 2 * i + 1, 2 * j, R)

    both_spins([[0.11 + 0.41, 0.21 - 0.31j], [0.21 + 0.31j, 0.11 - 0.41]], 0, 1, [1, 0, 2])
    both_spins([[0.42, 0], [0, 0.42]], 0, 1, [3, 0, 2])
    both_spins([[-0.12, 0], [0, -0.12]], 1, 2, [2, 0, 3])
    both_spins([[-0.34 + 0.29, 0.21 + 0.14j], [0.21 - 0.14j, -0.34 - 0.29]], 2, 0, [-1, 0, 2])
    m1 = quiet(tb.tb_model, 2, 2, LAT2, [[0.3, 0.2], [0.1, 0.3], [0.2, 0.4]], nspin=2)
    m1.set_onsite([-2.3, 0.5, 0.1])
    m1.set_hop([[0.11 + 0.41, 0.21 - 0.31j], [0.21 + 0.31j, 0.11 - 0.41]], 0, 1, [1, 2])
    m1.set_hop(0.42, 0, 1, [3, 2])
    m1.set_hop(-0.12, 1, 2, [2, 3])
    m1.set_hop([-0.34, 0.21, -0.14, 0.29], 2, 0, [-1, 2])       # a0 I + a1 sx + a2 sy + a3 sz
    equivalent(tb, [m0, m1], [2, 1], [[0, 1], [0, 1]])


def test_smallest_models(tb):                               # tests/test_pythtb.py:19-62
    assert isinstance(tb.__version__, str) and tb.__version__
    one = quiet(tb.tb_model, 0, 1, [[1.0]], [[0.0]])
    one.set_onsite([2.5])
    ev = one.solve_all()
    assert ev.shape == (1,) and np.allclose(ev, [2.5])
    assert np.array_equal(one.solve_all(), one.solve_all())
    two = quiet(tb.tb_model, 0, 1, [[1.0]], [[0.0], [0.5]])
    two.set_onsite([0.0, 0.0])
    two.set_hop(3.0, 0, 1)
    ev = np.sort(two.solve_all())
    assert ev.shape == (2,) and np.allclose(ev, [-3.0, 3.0])
    chain = quiet(tb.tb_model, 1, 1, [[1.0]], [[0.0]])
    chain.set_onsite([0.0])
    k_vec, k_dist, k_node = chain.k_path([[0.0], [0.5]], 5, report=False)
    assert k_vec.shape == (5, 1) and k_dist.shape == (5,) and len(k_node) == 2
    assert k_dist[0] == pytest.approx(0.0) and k_dist[-1] > 0.0


def test_direct_table_edits_are_seen(tb):
    """Scripts sometimes write the private tables directly; the device copy must follow."""
    m = quiet(tb.tb_model, 1, 1, [[1.0]], [[0.0], [0.5]])
    m.set_onsite([0.0, 0.0])
    m.set_hop(1.0, 0, 1, [0])
    e0 = m.solve_one([0.0])
    m._site_energies[1] = 3.0                      # no setter involved
    e1 = m.solve_one([0.0])
    assert not np.allclose(e0, e1) and np.allclose(np.sort(e1), np.linalg.eigvalsh(np.array([[0.0, 1.0], [1.0, 3.0]])))
    m._orb[1, 0] = 0.25                            # orbital moved: eigenvector phases change, energies do not
    v0 = m.solve_one([0.3], eig_vectors=True)[1]
    m._orb[1, 0] = 0.75
    v1 = m.solve_one([0.3], eig_vectors=True)[1]
    assert not np.allclose(np.abs(v0[0, 1] / v0[0, 0] - v1[0, 1] / v1[0, 0]), 0.0)
