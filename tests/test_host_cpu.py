"""CPU-only checks: host logic of the tb_model / wf_array mirror (no kernels are
launched), the C-ABI library loads and exports every symbol include/tbk.h
declares, and the product fails loudly (never falls back) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_tables, load_golden
import helpers as hp
from oracle import tb_oracle as orc

import pythtb_amd as tb
from pythtb_amd import _lib

HAS_GPU = False
try:
    _n = ctypes.c_int(0)
    HAS_GPU = _lib.lib.tbk_device_count(ctypes.byref(_n)) == 0 and _n.value > 0
except Exception:
    pass


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tbk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tbk_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libtbk.so lacks %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib.tbk_version() >= 100


def test_argument_errors_reported_without_gpu():
    assert _lib.lib.tbk_ctx_sync(None) != 0
    assert b"null ctx" in _lib.lib.tbk_last_error()
    assert _lib.lib.tbk_model_info(None, None, None, None) != 0


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    m = hp.haldane(tb.tb_model)
    with pytest.raises(_lib.TbkError, match="no CPU fallback|No CPU|needs an AMD GPU"):
        m.solve_all([[0.0, 0.0]])
    with pytest.raises(_lib.TbkError):
        tb.wf_array(m, [5, 5]).solve_on_grid([0.0, 0.0])
    src = ""
    for f in ("model.py", "wfarray.py", "_lib.py", "__init__.py", "shard.py"):
        src += open(os.path.join(ROOT, "pythtb_amd", f)).read()
    assert "oracle" not in src.replace("no oracle", "")       # the product never imports the checker
    assert "linalg.eigh" not in src and "linalg.eigvalsh" not in src and "linalg.svd" not in src


@pytest.mark.parametrize("name", ["graphene", "haldane0", "km_odd", "chain3", "per02", "molecule", "spin_chain", "cubic16"])
def test_model_tables_match_reference(name):
    """set_onsite/set_hop/_val_to_block store exactly what the reference stores."""
    t = golden_tables(load_golden("point_" + name))
    if name == "graphene":
        m = hp.graphene(tb.tb_model)
    elif name == "haldane0":
        m = hp.haldane(tb.tb_model, 0.0)
    elif name == "km_odd":
        m = hp.kane_mele(tb.tb_model, "odd")        # exercises mode="add" and 4-vector amplitudes
    elif name == "chain3":
        m = hp.chain3(tb.tb_model, -1.0, 2.0, 0.3)
    elif name == "cubic16":
        m = hp.cubic16(tb.tb_model)
    else:
        m = hp.model_from_tables(tb.tb_model, t)
    mine = orc.model_tables(m)
    for key in ("dim_k", "dim_r", "nspin", "orb", "lat", "hop_i", "hop_j", "hop_R", "per"):
        assert np.array_equal(mine[key], t[key]), (name, key)
    assert np.max(np.abs(mine["hop_amp"] - t["hop_amp"])) < 1e-16
    assert np.max(np.abs(mine["site_energies"] - t["site_energies"])) < 1e-16
    # flattened device tables: shapes and the periodic-component selection
    orb_per, onsite, hi, hj, hR, amp = m._flat_tables()
    assert orb_per.shape == (m._norb, m._dim_k) and onsite.shape == (m._norb, m._nspin, m._nspin)
    assert hR.shape == (len(m._hoppings), m._dim_k) and amp.shape == (len(m._hoppings), m._nspin, m._nspin)
    if m._dim_k:
        assert np.array_equal(hR, t["hop_R"][:, list(t["per"])])
        assert np.array_equal(orb_per, t["orb"][:, list(t["per"])])


def test_flattened_tables_rebuild_hamiltonian():
    """The (orb, onsite, hop_*) arrays handed to tbk_model_upload define the same H(k)."""
    for m in (hp.kane_mele(tb.tb_model), hp.random_model(tb.tb_model, 3, 2, 2, 3), hp.random_model(tb.tb_model, 5, 3, 1, 4)):
        orb_per, onsite, hi, hj, hR, amp = m._flat_tables()
        ns, no = m._nspin, m._norb
        k = np.random.default_rng(0).random((5, m._dim_k))
        ref = orc.ham_batch(m, k)
        for ik, kk in enumerate(k):
            H = np.zeros((no, ns, no, ns), dtype=complex)
            for o in range(no):
                H[o, :, o, :] = onsite[o]
            for h in range(len(hi)):
                ph = np.exp(2j * np.pi * kk @ (hR[h] + orb_per[hj[h]] - orb_per[hi[h]]))
                H[hi[h], :, hj[h], :] += amp[h] * ph
                H[hj[h], :, hi[h], :] += (amp[h] * ph).conj().T
            assert np.max(np.abs(H.reshape(no * ns, no * ns) - ref[ik])) < 1e-13


def test_set_hop_and_onsite_rules():
    m = hp.quiet(tb.tb_model, 2, 2, hp.LAT, hp.ORB)
    m.set_hop(1.0, 0, 1, [0, 0])
    with pytest.raises(Exception, match="already specified"):
        m.set_hop(2.0, 0, 1, [0, 0])
    with pytest.raises(Exception, match="implicitely specified"):
        m.set_hop(2.0, 1, 0, [0, 0])
    m.set_hop(2.0, 1, 0, [0, 0], allow_conjugate_pair=True)
    m.set_hop(0.5, 0, 1, [0, 0], mode="add", allow_conjugate_pair=True)
    m.set_hop(0.25, 0, 1, [1, 0], mode="reset")
    assert [h[0] for h in m._hoppings] == [1.5, 2.0, 0.25]
    with pytest.raises(Exception, match="set_onsite instead"):
        m.set_hop(1.0, 0, 0, [0, 0])
    with pytest.raises(Exception, match="out of scope"):
        m.set_hop(1.0, 0, 2, [0, 0])
    with pytest.raises(Exception, match="must equal dim_r"):
        m.set_hop(1.0, 0, 1, [0])
    with pytest.raises(Exception, match="Need to specify ind_R"):
        m.set_hop(1.0, 0, 1)
    with pytest.raises(Exception, match="mode parameter"):
        m.set_hop(1.0, 0, 1, [3, 3], mode="bogus")
    m.set_onsite([1.0, 2.0])
    with pytest.raises(Exception, match="already specified"):
        m.set_onsite([1.0, 2.0])
    m.set_onsite(3.0, 1, mode="reset")
    m.set_onsite(0.5, 1, mode="add")
    assert np.array_equal(m._site_energies, [1.0, 3.5])
    with pytest.raises(Exception, match="imaginary"):
        m.set_onsite(1.0j, 0, mode="reset")
    with pytest.raises(Exception, match="Wrong number"):
        m.set_onsite([1.0])
    s = hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[0.0]], nspin=2)
    s.set_onsite([[0.1, 0.2, 0.3, 0.4]])
    assert np.allclose(s._site_energies[0], [[0.5, 0.2 - 0.3j], [0.2 + 0.3j, -0.3]])
    with pytest.raises(Exception, match="Hermitian"):
        s.set_onsite([[[0, 1], [2, 0]]], mode="reset")
    s.set_hop(2, 0, 0, 1)                                     # 1-D: integer R accepted
    assert np.array_equal(s._hoppings[0][3], [1]) and s._hoppings[0][0].shape == (2, 2)
    with pytest.raises(Exception, match="Wrong format"):
        s.set_hop([1, 2, 3], 0, 0, 2)


def test_constructor_rules():
    with pytest.raises(Exception, match="dim_k out of range"):
        tb.tb_model(5, 5)
    with pytest.raises(Exception, match="dim_r out of range"):
        tb.tb_model(2, 1)
    with pytest.raises(Exception, match="not an integer"):
        tb.tb_model(1.0, 1)
    with pytest.raises(Exception, match="right handed"):
        tb.tb_model(2, 2, [[0, 1], [1, 0]], [[0, 0]])
    with pytest.raises(Exception, match="close to zero"):
        tb.tb_model(2, 2, [[1, 0], [1, 0]], [[0, 0]])
    with pytest.raises(Exception, match="nspin"):
        tb.tb_model(1, 1, [[1.0]], [[0.0]], nspin=3)
    with pytest.raises(Exception, match="periodic"):
        tb.tb_model(1, 2, [[1, 0], [0, 1]], [[0, 0]], per=[0, 1])
    m = hp.quiet(tb.tb_model, 1, 2, None, 3)
    assert m._norb == 3 and np.array_equal(m._lat, np.identity(2)) and m._per == [0]
    assert m.get_num_orbitals() == 3 and m.get_orb().shape == (3, 2) and m.get_lat().shape == (2, 2)


def test_k_generators_match_reference():
    g = load_golden("kgen")
    gr = hp.graphene(tb.tb_model)
    kv, kd, kn = gr.k_path(g["g_path"], 121, report=False)
    assert np.array_equal(kv, g["g_kvec"]) and np.array_equal(kd, g["g_kdist"]) and np.array_equal(kn, g["g_knode"])
    kv, kd, kn = gr.k_path(g["km_path"], 101, report=False)
    assert np.array_equal(kv, g["km_kvec"]) and np.array_equal(kd, g["km_kdist"]) and np.array_equal(kn, g["km_knode"])
    c = hp.chain3(tb.tb_model, -1.0, 2.0, 0.0)
    for key in ("full", "fullc", "half"):
        kv, kd, kn = c.k_path(key, 17, report=False)
        assert np.array_equal(kv, g["c_%s_kvec" % key]) and np.array_equal(kd, g["c_%s_kdist" % key])
        assert np.array_equal(kn, g["c_%s_knode" % key])
    kv, kd, kn = c.k_path([[-0.5], [0.5]], 31, report=False)
    assert np.array_equal(kv, g["c_seg_kvec"]) and np.array_equal(kd, g["c_seg_kdist"])
    p = hp.model_from_tables(tb.tb_model, golden_tables(load_golden("point_per02")))
    kv, kd, kn = p.k_path([[0.0, 0.0], [0.5, 0.0], [0.5, 0.5]], 40, report=False)
    assert np.array_equal(kv, g["p_kvec"]) and np.allclose(kd, g["p_kdist"], atol=1e-15, rtol=0)
    assert np.array_equal(gr.k_uniform_mesh([4, 6]), g["mesh_4_6"])
    assert np.array_equal(hp.cubic16(tb.tb_model).k_uniform_mesh([3, 4, 5]), g["mesh_3_4_5"])
    assert np.array_equal(c.k_uniform_mesh([7]), g["mesh_7"])
    with pytest.raises(Exception, match="more points"):
        gr.k_path([[0, 0], [0.5, 0.5]], 1, report=False)
    with pytest.raises(Exception, match="do not match"):
        gr.k_path([[0.0], [0.5]], 5, report=False)
    with pytest.raises(Exception, match="Incorrect size"):
        gr.k_uniform_mesh([4])
    hp.quiet(gr.k_path, [[0, 0], [0.5, 0.5]], 5)             # report path runs


def test_wf_array_host_side():
    m = hp.haldane(tb.tb_model)
    w = tb.wf_array(m, [4, 5])
    assert w._wfs.shape == (4, 5, 2, 2) and w._wfs.dtype == complex and not w._wfs.any()
    w[1, 2] = [[1, 2], [3, 4j]]
    assert w[1, 2][1, 1] == 4j and w[-3, -3][0, 1] == 2
    with pytest.raises(IndexError):
        w[4, 0]
    with pytest.raises(IndexError):
        w[0, -6]
    with pytest.raises(TypeError):
        w[0.5, 1]
    with pytest.raises(TypeError):
        w[0, 1, 2]
    sub = w.choose_states([1])
    assert sub._nsta_arr == 1 and sub._wfs.shape == (4, 5, 1, 2) and sub._wfs[1, 2, 0, 1] == 4j
    assert w.empty_like()._wfs.shape == (4, 5, 2, 2) and w.empty_like(nsta_arr=3)._wfs.shape == (4, 5, 3, 2)
    assert w._model is not m and w._model._hoppings is not m._hoppings
    w1 = tb.wf_array(hp.chain3(tb.tb_model, -1, 2, 0), [6])
    w1[5] = np.ones((3, 3))
    with pytest.raises(TypeError):
        w1[1, 2]
    ks = tb.wf_array(hp.kane_mele(tb.tb_model), [3, 3])
    assert ks._wfs.shape == (3, 3, 4, 2, 2)
    with pytest.raises(Exception, match="2 or larger"):
        tb.wf_array(m, [4, 1])
    with pytest.raises(Exception, match="not an integer"):
        tb.wf_array(m, [4, 4], nsta_arr=1.5)
    # wf[i,j] is a live, writable view like the reference's (pythtb.py:2662-2666)
    w[1, 2][0] *= 2.0
    v = w[2, 3]
    v[1, 0] = 7.0
    assert w[1, 2][0, 1] == 4 and w._wfs[1, 2, 0, 0] == 2 and w[2, 3][1, 0] == 7 and w._wfs[2, 3, 1, 0] == 7


def test_uniform_mesh_is_a_plain_ndarray():
    """k_uniform_mesh returns what the reference returns (pythtb.py:1792-1861): an ordinary, writable, C-contiguous ndarray --
    no subclass, no hidden note of which mesh it is.  (Round 4 tagged it so that solve_all could skip the upload; a write through
    an untracked alias then returned eigenvalues of a k list the caller no longer held)"""
    m = hp.haldane(tb.tb_model, 0.2)
    ref = np.divide(np.indices((4, 6)).reshape(2, -1).T, np.array([4.0, 6.0]))
    k = m.k_uniform_mesh([4, 6])
    assert type(k) is np.ndarray and k.flags.writeable and k.flags["C_CONTIGUOUS"] and k.flags.owndata is not None
    assert k.dtype == np.float64 and np.array_equal(k, ref) and repr(k) == repr(ref)
    m1 = hp.chain3(tb.tb_model, -1.0, 2.0, 0.3)
    assert type(m1.k_uniform_mesh([7])) is np.ndarray


def test_z2_index_from_the_reference_wannier_centres():
    """Z2 from the `wan_cent` arrays the reference's own kane_mele test holds (tests/test_examples/kane_mele: index 0
    "even", 1 "odd"; centres in units of 2 pi on 41 strings) and from the full-size configs[3] array captured from
    the reference: odd -> 1, even -> 0, either half of the zone."""
    ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_tests", "kane_mele", "kane_mele_wan_cent.npy"))
    for t, want in ((0, 0), (1, 1)):
        for half in ("upper", "lower"):
            assert tb.z2_from_wilson_centres(2 * np.pi * ref[t], half=half) == want
    full = os.path.join(ROOT, "tests", "golden", "full_size.npz")
    if os.path.exists(full):
        assert tb.z2_from_wilson_centres(np.load(full)["D_wan_cent"]) == 1
    # two decoupled copies of a winding band wind twice: trivial
    k = np.linspace(-0.5, 0.5, 41)
    twice = np.stack([2 * np.pi * k + 0.3, 2 * np.pi * k + 0.3 + np.pi], axis=1)
    once = np.stack([np.where(k >= 0, 2 * np.pi * k, -2 * np.pi * k) + 0.1, -np.where(k >= 0, 2 * np.pi * k, -2 * np.pi * k) - 0.1], axis=1)
    assert tb.z2_from_wilson_centres(once) == 1
    with pytest.raises(Exception, match="odd number"):
        tb.z2_from_wilson_centres(twice[:40])


def test_phase_continuity_helpers_match_oracle():
    from pythtb_amd.wfarray import _array_phases_cont, _one_phase_cont
    rng = np.random.default_rng(3)
    for _ in range(20):
        pha = rng.uniform(-np.pi, np.pi, 40)
        clos = rng.uniform(-10, 10)
        assert np.allclose(_one_phase_cont(pha, clos), orc.one_phase_cont(pha, clos), atol=1e-14)
        arr = rng.uniform(-np.pi, np.pi, (15, 4))
        c0 = rng.uniform(-7, 7, 4)
        assert np.allclose(_array_phases_cont(arr, c0), orc.array_phases_cont(arr, c0), atol=1e-14)
    arr = np.array([[0.1, 0.1, 3.0], [0.1, 3.0, 0.1]])        # exact ties follow the reference's last-wins rule
    assert np.allclose(_array_phases_cont(arr, arr[0]), orc.array_phases_cont(arr, arr[0]))


def test_display_report_matches_reference_text(capsys):
    """tb_model.display(): same text as the reference prints (tests/golden/display_reports.json)."""
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "display_reports.json")))
    per02 = hp.quiet(tb.tb_model, 2, 3, [[3.0, 0.1, 0.4], [0.1, 3.1, 1.2], [0.8, 0.2, 3.5]],
                     [[0.3, 0.1, 0.2], [0.1, 0.8, 0.3], [0.2, 0.3, 0.4]], per=[0, 2])
    per02.set_onsite([-2.3, 0.5, 0.1])
    per02.set_hop(0.24, 0, 1, [1, 0, 2])
    per02.set_hop(0.42, 0, 1, [3, 0, 2])
    per02.set_hop(-0.12, 1, 2, [2, 0, 3])
    per02.set_hop(-0.34, 2, 0, [-1, 0, 2])
    mol = hp.quiet(tb.tb_model, 0, 1, [[1.0]], [[0.0], [0.5], [0.8]])
    mol.set_onsite([0.1, -0.4, 0.7])
    mol.set_hop(3.0, 0, 1)
    mol.set_hop(0.5 + 0.25j, 1, 2)
    models = {"haldane02": hp.haldane(tb.tb_model, 0.2), "km_odd": hp.kane_mele(tb.tb_model, "odd"),
              "chain3": hp.chain3(tb.tb_model, -1.0, 2.0, 0.3), "per02": per02, "molecule": mol,
              "haldane_flake_2x2": hp.haldane(tb.tb_model, 0.2).cut_piece(2, 0).cut_piece(2, 1)}
    capsys.readouterr()
    for name, m in models.items():
        m.display()
        assert capsys.readouterr().out == want[name], name


def test_visualize_draws_what_the_reference_draws():
    """tb_model.visualize(): same artists (geometry, sizes, z-order, colours) and axis limits as the
    reference (tests/golden/visualize_artists.npz); eigenstate colouring in all three schemes."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.colors as mcolors
    import matplotlib.pyplot as plt
    g = np.load(os.path.join(ROOT, "tests", "golden", "visualize_artists.npz"))
    models = {"haldane02": (hp.haldane(tb.tb_model, 0.2), (0, 1)), "chain3": (hp.chain3(tb.tb_model, -1.0, 2.0, 0.3), (0, None)),
              "flake": (hp.haldane(tb.tb_model, 0.2).cut_piece(3, 0).cut_piece(2, 1), (1, 0))}
    for name, (m, dirs) in models.items():
        e = g[name + "/eig"]
        for scheme, vec in (("plain", None), ("wheel", e), ("red-blue", e), ("black", e)):
            fig, ax = m.visualize(dirs[0], dirs[1], eig_dr=vec, ph_color="black" if scheme == "plain" else scheme)
            rows = []
            for ln in ax.get_lines():
                x, y = np.asarray(ln.get_xdata(), float), np.asarray(ln.get_ydata(), float)
                pad = np.full(3 - len(x), np.nan)
                rows.append(np.concatenate([np.concatenate([x, pad]), np.concatenate([y, pad]),
                                            [ln.get_markersize(), ln.get_zorder(), ln.get_linewidth()],
                                            mcolors.to_rgba(ln.get_color())]))
            got, want = np.array(rows), g["%s/%s/lines" % (name, scheme)]
            assert got.shape == want.shape, (name, scheme)
            assert np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True), (name, scheme)   # same order, same artists
            assert np.allclose(list(ax.get_xlim()) + list(ax.get_ylim()), g["%s/%s/lims" % (name, scheme)], atol=1e-12)
            plt.close(fig)
    h = hp.haldane(tb.tb_model, 0.2)
    for bad in (lambda: h.visualize(0), lambda: h.visualize(0, 1, ph_color="rainbow"), lambda: h.visualize(0, 1, eig_dr=np.ones(3))):
        with pytest.raises(Exception):
            bad()


def test_k_path_too_few_points_raises_like_reference():
    """Two nodes landing on one path index is a 0/0 in the reference's interpolation loop
    (pythtb.py:1985-1996), which raises ZeroDivisionError; found by tests/golden/diff_fuzz_host.py."""
    from pythtb_amd import tb_model
    m = tb_model(1, 1, [[1.0]], [[0.0]])
    with pytest.raises(ZeroDivisionError):
        m.k_path([[-0.80], [-0.82], [0.35], [0.89]], 5, report=False)


def test_cut_piece_of_an_orbital_free_model_raises_like_reference():
    from pythtb_amd import tb_model
    m = tb_model(1, 1, [[1.0]], [[0.0]]).remove_orb(0)
    with pytest.raises(Exception, match="Wrong orb array rank"):
        m.cut_piece(2, 0)


def _flatten_host(m):
    """tbk_model_flatten_host on a model's tables: (slot, R, amp) term list and info = {pmax, nR, nnz, nslot}."""
    import ctypes as C
    from pythtb_amd import _lib
    orb_per, onsite, hop_i, hop_j, hop_R, hop_amp = m._flat_tables()
    nterm = C.c_int64(0)
    info = np.zeros(4, dtype=np.int32)
    args = (m._dim_k, m._norb, m._nspin, _lib.dptr(orb_per), _lib.dptr(onsite.view(float)), len(hop_i),
            _lib.iptr(hop_i), _lib.iptr(hop_j), _lib.iptr(np.ascontiguousarray(hop_R.reshape(-1))) if hop_R.size else None,
            _lib.dptr(hop_amp.view(float)) if hop_amp.size else None)
    _lib.check(_lib.lib.tbk_model_flatten_host(*args, 0, C.byref(nterm), None, None, None, _lib.iptr(info)))
    n = nterm.value
    slot = np.zeros(max(n, 1), dtype=np.int32)
    R = np.zeros((max(n, 1), 4), dtype=np.int32)
    amp = np.zeros(max(n, 1), dtype=complex)
    _lib.check(_lib.lib.tbk_model_flatten_host(*args, n, C.byref(nterm), _lib.iptr(slot), _lib.iptr(R),
                                               _lib.dptr(amp.view(float)), _lib.iptr(info)))
    return slot[:n], R[:n], amp[:n], info


@pytest.mark.parametrize("name", ["graphene", "haldane02", "km_odd", "chain3", "cubic16", "molecule", "per02", "spin_chain"])
def test_flattened_term_table_reproduces_reference_hamiltonian(name):
    """The host half of tbk_model_upload (merge of hoppings into slot-major (slot, R, amp) terms, pythtb.py:900-924
    restated as S = D H D^+) needs no GPU: rebuild H(k) from the dumped table in NumPy and compare with the H(k)
    the reference's _gen_ham produced for the same model (tests/golden/point_<name>.npz)."""
    import pythtb_amd as tb
    g = load_golden("point_" + name)
    m = hp.model_from_tables(tb.tb_model, golden_tables(g))
    slot, R, amp, info = _flatten_host(m)
    n, dim_k = m._nsta, m._dim_k
    assert info[3] == n * (n + 1) // 2
    assert np.all(np.diff(slot) >= 0)                                       # slot-major
    ab = [(a, b) for a in range(n) for b in range(a, n)]
    orb = np.repeat(m._orb[:, m._per], m._nspin, axis=0) if dim_k else np.zeros((n, 0))
    ks = g["k"][:16] if dim_k else [np.zeros(0)]                            # 0-D model: one Hamiltonian, no k
    for ik, k in enumerate(ks):
        S = np.zeros((n, n), dtype=complex)
        for s_, r_, a_ in zip(slot, R, amp):
            a, b = ab[s_]
            v = a_ * np.exp(2j * np.pi * np.dot(k, r_[:dim_k]))
            S[a, b] += v
            if a != b:
                S[b, a] += np.conj(v)
        e = np.exp(2j * np.pi * orb @ k) if dim_k else np.ones(n)
        H = np.conj(e)[:, None] * S * e[None, :]
        ref = np.asarray(g["ham"][ik]).reshape(n, n)
        assert np.max(np.abs(H - ref)) < 1e-13 * max(1.0, np.abs(ref).max())


def test_knobs_are_parsed_once_and_reloadable():
    from pythtb_amd import _lib
    assert _lib.lib.tbk_build_has_diagnostics() == 0                        # the shipped library carries no ablation code
    with _lib.knob("TBK_GRID_SEG", 3):
        pass
    assert os.environ.get("TBK_GRID_SEG") is None


def test_solver_dispatch_table_routes_the_measured_crossovers():
    """tbk_solver_regime (host only): the dispatch of the batched eigen-solver is one table (tbk_solve.hip, kRegimeRules).
    The rows' boundaries are measurements; this pins the routing at and around them, for a 256-CU chip."""
    from pythtb_amd import _lib

    def regime(n, vec, form, nk, batch=None, rblocks=1):
        return _lib.lib.tbk_solver_regime(n, vec, form, nk, nk if batch is None else batch, 256, rblocks, None).decode()
    LIST, MESH, SUP = 0, 1, 2
    assert regime(2, 1, MESH, 1 << 22) == "closed_form" and regime(4, 1, MESH, 1 << 21) == "ql_small"
    assert regime(5, 1, LIST, 100) == "reg" and regime(8, 0, MESH, 10 ** 6) == "reg"
    # 9..16: chip-filling batches with eigenvectors take the direct solver, small ones keep Jacobi; eigenvalues at any count
    assert regime(16, 1, MESH, 257 ** 3) == "ql16" and regime(16, 1, LIST, 2049) == "ql16"
    assert regime(16, 1, LIST, 2048) == "row16" and regime(12, 1, LIST, 2048) == "wg_lds"
    assert regime(12, 1, MESH, 300, batch=300) == "wg_lds" and regime(12, 1, MESH, 300, batch=10 ** 5) == "ql16"
    assert regime(12, 0, LIST, 7) == "ql16" and regime(16, 1, LIST, 10 ** 5, rblocks=0) == "wave"
    # 17..64
    # (17..32 with eigenvectors: the direct kernels of round 6 from 3 matrices per CU -- 2048 x n=32 1.63 -> 0.51 ms; 33..64 from 8)
    assert regime(32, 1, SUP, 16384) == "qlw" and regime(32, 1, SUP, 2048) == "qlw" and regime(20, 1, SUP, 769) == "qlw"
    assert regime(32, 1, SUP, 768) == "wg_lds" and regime(20, 1, SUP, 768) == "wg_lds" and regime(33, 1, SUP, 2048) == "wg_lds"
    assert regime(64, 0, SUP, 1) == "qlw" and regime(20, 1, LIST, 2049) == "qlw" and regime(33, 1, LIST, 2049) == "qlw"
    # ... and 40..64 with eigenvectors below the QL batch take the direct method of 65+ states instead of workgroup Jacobi
    assert regime(64, 1, SUP, 16) == "trigv" and regime(48, 1, LIST, 2048) == "trigv" and regime(48, 1, LIST, 2049) == "qlw"
    assert regime(47, 1, SUP, 512) == "wg_lds" and regime(47, 1, SUP, 513) == "trigv" and regime(39, 1, SUP, 1024) == "wg_lds"
    assert regime(64, 0, MESH, 100, batch=100) == "wg_lds"       # (eigenvalues only on a mesh: unchanged)
    # above 64 states
    assert regime(300, 0, LIST, 101) == "trig" and regime(513, 0, LIST, 1) == "big" and regime(513, 0, LIST, 2) == "trig"
    # (five 1024-state matrices: too few for a CU each, enough work -- 5 x 1024^2 >= 1.4e6 -- for block Jacobi)
    assert regime(1024, 0, SUP, 5) == "blocked" and regime(1024, 0, SUP, 6) == "trig" and regime(1025, 0, SUP, 100) == "blocked"
    assert regime(800, 0, SUP, 1) == "big" and regime(600, 0, LIST, 1) == "big" and regime(600, 0, LIST, 2) == "trig"
    # with eigenvectors, 65..1024 states: the direct method (tbk_solve_trigv.inl); above, and with TBK_TRIGV=0, the Jacobi solvers
    assert regime(128, 1, LIST, 512) == "trigv" and regime(300, 1, MESH, 101, batch=101) == "trigv" and regime(90, 1, SUP, 1) == "trigv"
    assert regime(800, 1, SUP, 5) == "blocked" and regime(800, 1, SUP, 6) == "trigv" and regime(800, 1, SUP, 1) == "big"
    assert regime(1025, 1, SUP, 3) == "blocked" and regime(2048, 0, SUP, 1) == "blocked"
    with _lib.knob("TBK_TRIGV", 0):
        assert regime(128, 1, LIST, 512) == "blocked" and regime(128, 1, LIST, 64) == "big" and regime(70, 1, LIST, 200) == "wg_global"
        assert regime(95, 1, LIST, 4000) == "wg_global" and regime(230, 1, LIST, 26) == "big" and regime(230, 1, LIST, 27) == "blocked"
        assert regime(300, 1, MESH, 101, batch=101) == "blocked" and regime(90, 1, SUP, 1) == "big"
        assert regime(64, 1, SUP, 16) == "wg_lds"
    # knobs move the boundaries, not the code
    with _lib.knob("TBK_QL16_MIN", 0):
        assert regime(12, 1, LIST, 5) == "ql16"
    with _lib.knob("TBK_QL16", 0):
        assert regime(16, 1, LIST, 10 ** 5) == "row16" and regime(16, 1, MESH, 10 ** 5) == "wave"
    with _lib.knob("TBK_TRIG", 0):
        assert regime(300, 0, LIST, 101) == "blocked"
    with _lib.knob("TBK_BLOCKED", 1), _lib.knob("TBK_TRIGV", 0):
        assert regime(70, 1, LIST, 3) == "blocked"
    with _lib.knob("TBK_BIG_FROM", 65), _lib.knob("TBK_TRIGV", 0):
        assert regime(70, 1, LIST, 1000) == "big"
    note = ctypes.c_char_p()
    _lib.lib.tbk_solver_regime(300, 0, 0, 101, 101, 256, 1, ctypes.byref(note))
    assert b"12.6 ms" in note.value


def test_phase_continuity_in_c_equals_the_reference_loops():
    """tbk_one_phase_cont / tbk_array_phases_cont (host only) against the oracle's restatement of _one_phase_cont and
    _array_phases_cont (pythtb.py:3876-3921): random phase sets, exact ties (Kramers pairs: the LAST index among equal minima),
    jumps of several 2 pi, single-band sets."""
    from oracle import tb_oracle as orc
    from pythtb_amd import wfarray as wa
    rng = np.random.default_rng(11)
    for n in (1, 2, 7, 60):
        pha = rng.uniform(-20.0, 20.0, n)
        for clos in (0.0, 3.0, -7.5, pha[0]):
            assert np.array_equal(wa._one_phase_cont(pha, clos), orc.one_phase_cont(pha, clos))
    for n0, nb in ((1, 1), (5, 1), (9, 2), (30, 4), (12, 7)):
        arr = rng.uniform(-np.pi, np.pi, (n0, nb))
        arr[n0 // 2] = np.sort(arr[n0 // 2])
        if nb >= 2:
            arr[n0 // 3, 1] = arr[n0 // 3, 0]                 # an exact tie
            arr[-1, :2] = [0.3, 0.3]
        arr[0] += 2 * np.pi * rng.integers(-2, 3, nb)
        for clos in (arr[0], np.zeros(nb), rng.uniform(-9, 9, nb)):
            got = wa._array_phases_cont(arr, clos)
            ref = orc.array_phases_cont(arr, np.array(clos, dtype=float))
            assert np.array_equal(got, ref), (n0, nb)


class _FakeWfsLib(object):
    """Stand-in for the seven storage entry points of libtbk that wf_array's host bookkeeping drives (include/tbk.h:145-197):
    the "device copy" is a NumPy array [point][...].  Test infrastructure for the ledger logic only -- no compute."""

    def __init__(self):
        self.dev = {}
        self.calls = {"up_pts": 0, "down_pts": 0, "up": 0, "down": 0}
        self.next = 1

    def _arr(self, ptr, n):
        # (np.ctypeslib.as_array on a pointer leaves a reference CYCLE that keeps the caller's array referenced until the next
        # gc pass; the real library holds nothing, and the ledger's chunk recycling reads reference counts)
        return np.frombuffer((ctypes.c_double * n).from_address(ctypes.addressof(ptr.contents)), dtype=float)

    def tbk_wfs_create(self, ctx, dim, mesh, nsta, ncomp, out):
        m = np.ctypeslib.as_array(mesh, shape=(dim,))
        h = self.next
        self.next += 1
        self.dev[h] = np.zeros((int(np.prod(m)), nsta * ncomp), dtype=complex)
        out._obj.value = h
        return 0

    def _h(self, h):
        return self.dev[h.value if hasattr(h, "value") else h]

    def tbk_wfs_free(self, h):
        return 0

    def tbk_wfs_upload(self, h, p):
        d = self._h(h)
        d[...] = self._arr(p, d.size * 2).view(complex).reshape(d.shape)
        self.calls["up"] += 1
        return 0

    def tbk_wfs_download(self, h, p):
        d = self._h(h)
        self._arr(p, d.size * 2)[...] = d.reshape(-1).view(float)
        self.calls["down"] += 1
        return 0

    def tbk_wfs_upload_points(self, h, idx, n, p):
        d = self._h(h)
        ids = np.ctypeslib.as_array(idx, shape=(n,))
        d[ids] = self._arr(p, n * d.shape[1] * 2).view(complex).reshape(n, -1)
        self.calls["up_pts"] += 1
        return 0

    def tbk_wfs_download_points(self, h, idx, n, p):
        d = self._h(h)
        ids = np.ctypeslib.as_array(idx, shape=(n,))
        self._arr(p, n * d.shape[1] * 2)[...] = d[ids].reshape(-1).view(float)
        self.calls["down_pts"] += 1
        return 0

    def tbk_wfs_impose(self, h, mesh_dir, phase):
        return 0


def _fake_resident(monkeypatch, mesh, small_mirror_bytes=None):
    """A wf_array whose device copy is a _FakeWfsLib array filled with recognisable numbers, device authoritative."""
    import types
    from pythtb_amd import wfarray as wmod, _lib as real
    fake = _FakeWfsLib()
    shim = types.SimpleNamespace(lib=fake, check=real.check, dptr=real.dptr, iptr=real.iptr,
                                 default_context=lambda: types.SimpleNamespace(handle=None))
    monkeypatch.setattr(wmod, "_lib", shim)
    m = hp.haldane(tb.tb_model, 0.3)
    w = tb.wf_array(m, mesh)
    if small_mirror_bytes is not None:
        w._SMALL_MIRROR_BYTES = small_mirror_bytes
    h = w._dev_handle(w._shape())
    d = fake._h(h)
    d[...] = (np.arange(d.size).reshape(d.shape) + 1) * (1.0 + 0.5j)
    w._dev_valid = True

    def device_write(scale):
        d[...] = d * scale
        w._device_wrote()
    return w, fake, d, device_write


@pytest.mark.parametrize("pool", [False, True])
def test_point_ledger_setitem_after_a_device_write_is_not_reverted(monkeypatch, pool):
    """wf[i,j] read -> a kernel rewrites the array -> `wf[i,j] = v` -> next device use.  The assignment skipped
    the (stale) mirror row, so the next synchronisation saw "row differs from snapshot" and uploaded the OLD row over v."""
    w, fake, d, device_write = _fake_resident(monkeypatch, [5, 4], small_mirror_bytes=0 if pool else None)
    held = w[2, 1]
    fi = 2 * 4 + 1
    assert np.array_equal(held.reshape(-1), d[fi])
    device_write(2.0)                               # e.g. a second solve_on_grid / impose_pbc
    assert np.array_equal(held.reshape(-1), d[fi])  # the held array follows the storage
    v = np.full((2, 2), 7.0 - 1.0j)
    w[2, 1] = v
    assert np.array_equal(d[fi], v.reshape(-1)) and np.array_equal(held, v)
    w._ensure_dev()                                 # what every berry_* / position_* call does first
    assert np.array_equal(d[fi], v.reshape(-1)) and np.array_equal(w[2, 1], v)
    held[0, 0] = 3.0                                # and the held array is still live
    w._ensure_dev()
    assert d[fi][0] == 3.0 and d[fi][1] == 7.0 - 1.0j
    assert fake.calls["up"] == 0                    # points only, never the array


@pytest.mark.parametrize("pool", [False, True])
def test_point_ledger_read_loop_is_linear_and_temporaries_are_not_lost(monkeypatch, pool):
    """a plain `for i, j: wf[i, j]` loop compared every point handed out so far on every access (16 s for
    61x61).  Nothing is compared per access now; writes -- also through temporaries that die at once -- reach the device copy
    in ONE batched upload before its next use; asking for a point twice returns the same memory, like the reference's views."""
    import time
    n = 61
    w, fake, d, device_write = _fake_resident(monkeypatch, [n, n], small_mirror_bytes=0 if pool else None)
    t0 = time.perf_counter()
    acc = 0.0
    for i in range(n):
        for j in range(n):
            acc += w[i, j][0, 0].real
    dt = time.perf_counter() - t0
    assert acc == float(np.sum(d[:, 0].real))
    assert dt < 2.0, dt                             # (was quadratic: 16 s; linear: ~10-50 ms here)
    assert fake.calls["up_pts"] == 0
    w[3, 4][0] *= 2.0                               # temporary array, dies at once
    a, b = w[10, 11], w[10, 11]
    a[1, 1] = -5.0
    assert b[1, 1] == -5.0 and np.shares_memory(a, b)
    before = d.copy()
    w._ensure_dev()
    assert fake.calls["up_pts"] == 1 and fake.calls["up"] == 0
    before[3 * n + 4, :2] *= 2.0
    before[10 * n + 11, 3] = -5.0
    assert np.array_equal(d, before)
    w._ensure_dev()
    assert fake.calls["up_pts"] == 1                # nothing changed since: nothing uploaded
    # a device-side rewrite shows up in arrays still held, and their later writes still count
    device_write(-1.0)
    assert np.array_equal(a.reshape(-1), d[10 * n + 11])
    a[0, 0] = 9.0
    assert np.array_equal(w.to_host().reshape(n * n, -1), d) and d[10 * n + 11, 0] == 9.0


def test_point_ledger_gives_way_to_the_whole_mirror_past_its_cap(monkeypatch):
    """More live points than the ledger follows one by one: the whole mirror becomes the live copy (as with `_wfs`), and
    arrays handed out from the pool before that are still honoured."""
    w, fake, d, device_write = _fake_resident(monkeypatch, [20, 20], small_mirror_bytes=0)
    w._PT_TRACK_MAX = 64
    w._PT_TRACK_BYTES = 1 << 40
    first = w[0, 1]
    held = [w[1 + j // 20, j % 20] for j in range(70)]      # arrays the caller keeps: their chunk cannot be recycled
    assert w._host_exported and len(w._pt_copies) == 64 and len(held) == 70
    assert np.shares_memory(w[0, 1], first)         # a pool point stays ONE buffer once the mirror is exported
    first[0, 0] = 11.0
    late = w[15, 15]                                # a view of the exported mirror
    late[1, 0] = 12.0
    w._ensure_dev()
    assert d[1, 0] == 11.0 and d[15 * 20 + 15, 2] == 12.0


@pytest.mark.parametrize("pool", [False, True])
def test_read_loop_past_the_cap_keeps_device_residency(monkeypatch, pool):
    """`for i, j: x = wf[i, j]` over more points than the ledger follows ran the pool to its cap, downloaded
    the whole array and left `_host_exported` set for good, so every later berry_* call re-uploaded the mirror.  Chunks (pool)
    or the mirror (small arrays) that no caller array views any more are forgotten instead; writes made on the way -- also
    through temporaries -- still reach the device; an array the caller keeps across the recycling stays live."""
    w, fake, d, device_write = _fake_resident(monkeypatch, [40, 40], small_mirror_bytes=0 if pool else None)
    w._PT_TRACK_MAX = 64 if not pool else 768       # (pool chunks hold 256 points; two stay referenced below)
    w._PT_TRACK_BYTES = 1 << 40
    # held across the whole loop (small arrays hand out rows of ONE mirror: an array kept there means the mirror stays the
    # live copy past the cap, as before -- at most 32 MB per device use)
    keep = w[0, 0] if pool else None
    want = d.copy()
    acc = 0.0
    for i in range(40):
        for j in range(40):
            if pool:
                x = w[i, j]                         # (the previous point's array is alive during the next access)
                acc += x[0, 0].real
            else:
                acc += w[i, j][0, 0].real
            if (i * 40 + j) % 97 == 0:
                w[i, j][1, 1] = -3.0                # a write through a temporary
                want[i * 40 + j, 3] = -3.0
    assert acc == float(np.sum(want[:, 0].real))
    x = None
    assert not w._host_exported
    assert len(w._pt_copies if pool else w._pt_views) <= w._pt_cap()
    if pool:
        assert fake.calls["down"] == 0              # never the whole array
    ups = fake.calls["up"]
    w._ensure_dev()
    w._ensure_dev()
    assert fake.calls["up"] == ups == 0             # berry_* twice: no full-array upload
    assert np.array_equal(d, want)
    if pool:
        keep[0, 1] = 5.5                            # the array held since before the recycling is still live
        w._ensure_dev()
        assert d[0, 1] == 5.5 and fake.calls["up"] == 0
        assert np.shares_memory(w[0, 0], keep)


def test_band_lists_wrap_like_a_numpy_fancy_index():
    """occ of berry_phase / berry_flux is `_wfs[:, occ, :]` in the reference (pythtb.py:2981, :3141): differential check of
    wf_array._wrap_occ against NumPy's own indexing on random signed lists, incl. the exception class."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 4, 16):
        w = tb.wf_array(hp.haldane(tb.tb_model, 0.0), [3, 3], nsta_arr=n)
        ref = np.arange(n)
        for _ in range(200):
            occ = rng.integers(-n - 2, n + 2, size=int(rng.integers(1, 6)))
            try:
                want = ref[occ]
            except IndexError:
                with pytest.raises(IndexError, match="out of bounds for axis 1 with size %d" % n):
                    w._wrap_occ(w._occ(occ))
                continue
            assert np.array_equal(w._wrap_occ(w._occ(list(occ))), want)
        assert np.array_equal(w._wrap_occ(w._occ(range(-n, 0))), ref)
        assert np.array_equal(w._wrap_occ(w._occ("All")), ref)
