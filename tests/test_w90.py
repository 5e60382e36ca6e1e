"""Wannier90 importer against what the reference's `w90` class reads and builds
from its silicon example (tests/golden/w90_silicon.npz, made by make_w90_golden.py)."""
import os
import shutil

import numpy as np
import pytest

from conftest import ROOT
from helpers import quiet
from oracle import tb_oracle as orc

from pythtb_amd import w90

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "w90_silicon")
Z = np.load(os.path.join(GOLDEN, "w90_silicon.npz"))
VARIANTS = {
    "full": {},
    "quick": {"min_hopping_norm": 0.01},
    "cut": {"zero_energy": 6.2285135, "min_hopping_norm": 0.002, "max_distance": 7.9, "ignorable_imaginary_part": 1.0e-5},
}


@pytest.fixture(scope="module")
def si():
    return w90(DATA, "silicon")


def test_files_are_read_like_the_reference(si):
    assert si.num_wan == int(Z["num_wan"]) == 8
    assert np.array_equal(si.lat, Z["lat"])
    assert np.array_equal(si.xyz_cen, Z["xyz_cen"])
    assert np.array_equal(si.red_cen, Z["red_cen"])
    keys = np.array(list(si.ham_r.keys()), dtype=np.int32)
    assert np.array_equal(keys, Z["R_keys"])                      # same R vectors in the same (file) order
    assert np.array_equal(np.array([si.ham_r[R]["deg"] for R in si.ham_r]), Z["R_deg"])
    assert np.array_equal(np.array([si.ham_r[R]["h"] for R in si.ham_r]), Z["R_h"])


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_model_tables_match_reference(si, name):
    m = quiet(si.model, **VARIANTS[name])
    assert m._assume_position_operator_diagonal is False
    got = orc.model_tables(m)
    for key in ("dim_k", "dim_r", "nspin", "per", "hop_i", "hop_j", "hop_R"):
        assert np.array_equal(got[key], Z[name + "/" + key]), key
    for key in ("lat", "orb", "site_energies", "hop_amp"):
        assert np.array_equal(got[key], Z[name + "/" + key]), key    # bit for bit: same arithmetic
    with pytest.raises(Exception):
        m.set_hop(0.1, 0, 1, [0, 0, 0])                          # built tables behave like set_hop-made ones


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_oracle_bands_of_imported_model(si, name):
    m = orc.Model.from_tables(orc.model_tables(quiet(si.model, **VARIANTS[name])))
    kpts = Z["band_kpts"][::7]
    assert np.allclose(orc.solve_all(m, kpts), Z[name + "/evals"], rtol=0, atol=1e-11)


def test_dist_hop_shells_bands(si):
    dist, ham = si.dist_hop()
    assert np.array_equal(ham, Z["dist_hop_ham"])
    assert np.array_equal(dist, Z["dist_hop_dist"])
    assert np.array_equal(si.shells(), Z["shells2"])
    assert np.array_equal(si.shells(num_digits=1), Z["shells1"])
    kpts, ene = si.w90_bands_consistency()
    assert np.array_equal(kpts, Z["band_kpts"]) and np.array_equal(ene, Z["band_ene"])
    assert ene.shape == (8, kpts.shape[0])


def test_malformed_input_raises(tmp_path):
    for name in os.listdir(DATA):
        shutil.copy(os.path.join(DATA, name), tmp_path / name)
    win = (tmp_path / "silicon.win").read_text()
    (tmp_path / "silicon.win").write_text(win.lower().replace("begin unit_cell_cart", "begin unit_cell"))
    with pytest.raises(Exception, match="unit_cell_cart"):
        w90(str(tmp_path), "silicon")
    (tmp_path / "silicon.win").write_text(win)
    cen = (tmp_path / "silicon_centres.xyz").read_text().splitlines(True)
    (tmp_path / "silicon_centres.xyz").write_text("".join(cen[:3] + [cen[3].replace("X", "Si", 1)] + cen[4:]))
    with pytest.raises(Exception, match="centres"):
        w90(str(tmp_path), "silicon")
    (tmp_path / "silicon_centres.xyz").write_text("".join(cen))
    hr = (tmp_path / "silicon_hr.dat").read_text().splitlines(True)
    nws = int(hr[2])
    nlines = (nws + 14) // 15
    body = [l for l in hr[3 + nlines:] if not l.split()[:3] == ["-3", "1", "1"]]
    (tmp_path / "silicon_hr.dat").write_text("".join(hr[:3 + nlines] + body))
    if len(body) != len(hr) - 3 - nlines:                            # an R vector lost its partner
        with pytest.raises(Exception, match="negative R"):
            w90(str(tmp_path), "silicon")
    # a unit cell given in bohr is converted with the reference's constant
    (tmp_path / "silicon_hr.dat").write_text("".join(hr))
    lines = win.splitlines(True)
    at = [i for i, l in enumerate(lines) if l.lower().split()[:2] == ["begin", "unit_cell_cart"]][0]
    unit_line = lines[at + 1].strip().lower()
    rows_at = at + 2 if unit_line in ("bohr", "ang", "angstrom") else at + 1
    new = lines[:at + 1] + ["bohr\n"] + lines[rows_at:]
    (tmp_path / "silicon.win").write_text("".join(new))
    scale = 0.5291772108 if unit_line != "bohr" else 1.0
    ref_lat = Z["lat"] if unit_line != "bohr" else Z["lat"]
    got = w90(str(tmp_path), "silicon").lat
    assert np.allclose(got, ref_lat * scale, rtol=1e-15)


# --------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_gpu_bands_of_imported_model(si, name):
    m = quiet(si.model, **VARIANTS[name])
    kpts = Z["band_kpts"]
    ev = m.solve_all(kpts)
    assert np.allclose(ev[:, ::7], Z[name + "/evals"], rtol=0, atol=1e-10)
    if name == "full":            # the untruncated model reproduces Wannier90's own interpolation
        assert np.max(np.abs(ev - Z["band_ene"])) < 5e-5
    ev2, vec = m.solve_all(kpts[:16], eig_vectors=True)
    ham = orc.ham_batch(orc.Model.from_tables(orc.model_tables(m)), kpts[:16])
    for ik in range(16):
        V = vec[:, ik, :]
        assert np.max(np.abs(ham[ik] @ V.T - V.T * ev2[:, ik])) < 1e-11 * np.abs(ev2).max()


@pytest.mark.gpu
def test_gpu_w90_quick_example(si):                 # examples/w90_quick.py
    m = quiet(si.model, min_hopping_norm=0.01)
    path = [[0.5, 0.5, 0.5], [0.0, 0.0, 0.0], [0.5, -0.5, 0.0], [0.375, -0.375, 0.0], [0.0, 0.0, 0.0]]
    k_vec, k_dist, k_node = m.k_path(path, 101, report=False)
    ev = m.solve_all(k_vec)
    ref = orc.solve_all(orc.Model.from_tables(orc.model_tables(m)), k_vec)
    assert ev.shape == (8, 101) and np.max(np.abs(ev - ref)) < 1e-11 * np.abs(ref).max()
    w = quiet(lambda: __import__("pythtb_amd").wf_array(m, [5, 5, 5]))
    gaps = w.solve_on_grid([0.0, 0.0, 0.0])
    _, ogaps = orc.solve_on_grid(orc.Model.from_tables(orc.model_tables(m)), [5, 5, 5], [0.0, 0.0, 0.0], vectorised=True)
    assert np.max(np.abs(gaps - ogaps)) < 1e-9
