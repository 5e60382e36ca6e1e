#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE (development container only).

    PYTHONPATH=/root/reference python3 tests/golden/make_golden.py [--full]

Writes small .npz fixtures next to this file.  Nothing of the reference's source
travels: only model tables (inputs) and the numbers the reference computes from
them (expected outputs).  `--full` additionally runs the reference at the
BASELINE.json config sizes C and D (about 20 minutes of CPU) and stores the
scalars / small arrays in full_size.npz; `--full-E` does the same for configs[4]
at 16^3 (full_size_E.npz).

Models are built here through the reference's own public API, following the
example scripts cited next to each builder.
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
with contextlib.redirect_stdout(io.StringIO()):
    import pythtb as ref  # noqa: E402

from oracle import tb_oracle as orc  # noqa: E402  (only for model_tables: plain attribute dump)

HERE = os.path.dirname(os.path.abspath(__file__))
LAT = [[1.0, 0.0], [0.5, np.sqrt(3.0) / 2.0]]
ORB = [[1.0 / 3.0, 1.0 / 3.0], [2.0 / 3.0, 2.0 / 3.0]]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def graphene(delta=0.0):                      # examples/graphene.py:14-31
    m = quiet(ref.tb_model, 2, 2, LAT, ORB)
    m.set_onsite([-delta, delta])
    t = -1.0
    m.set_hop(t, 0, 1, [0, 0])
    m.set_hop(t, 1, 0, [1, 0])
    m.set_hop(t, 1, 0, [0, 1])
    return m


def haldane(delta):                           # examples/haldane_bp.py:16-41
    m = quiet(ref.tb_model, 2, 2, LAT, ORB)
    t = -1.0
    t2 = 0.15 * np.exp(1j * np.pi / 2.0)
    t2c = t2.conjugate()
    m.set_onsite([-delta, delta])
    m.set_hop(t, 0, 1, [0, 0])
    m.set_hop(t, 1, 0, [1, 0])
    m.set_hop(t, 1, 0, [0, 1])
    m.set_hop(t2, 0, 0, [1, 0])
    m.set_hop(t2, 1, 1, [1, -1])
    m.set_hop(t2, 1, 1, [0, 1])
    m.set_hop(t2c, 1, 1, [1, 0])
    m.set_hop(t2c, 0, 0, [1, -1])
    m.set_hop(t2c, 0, 0, [0, 1])
    return m


def kane_mele(topological):                   # examples/kane_mele.py:14-67
    m = quiet(ref.tb_model, 2, 2, LAT, ORB, nspin=2)
    esite = 2.5 if topological == "even" else 1.0
    thop = 1.0
    spin_orb = 0.6 * thop * 0.5
    rashba = 0.25 * thop
    m.set_onsite([esite, -esite])
    sx = np.array([0., 1., 0., 0])
    sy = np.array([0., 0., 1., 0])
    sz = np.array([0., 0., 0., 1])
    m.set_hop(thop, 0, 1, [0, 0])
    m.set_hop(thop, 0, 1, [0, -1])
    m.set_hop(thop, 0, 1, [-1, 0])
    m.set_hop(-1.j * spin_orb * sz, 0, 0, [0, 1])
    m.set_hop(1.j * spin_orb * sz, 0, 0, [1, 0])
    m.set_hop(-1.j * spin_orb * sz, 0, 0, [1, -1])
    m.set_hop(1.j * spin_orb * sz, 1, 1, [0, 1])
    m.set_hop(-1.j * spin_orb * sz, 1, 1, [1, 0])
    m.set_hop(1.j * spin_orb * sz, 1, 1, [1, -1])
    r3h = np.sqrt(3.0) / 2.0
    m.set_hop(1.j * rashba * (0.5 * sx - r3h * sy), 0, 1, [0, 0], mode="add")
    m.set_hop(1.j * rashba * (-1.0 * sx), 0, 1, [0, -1], mode="add")
    m.set_hop(1.j * rashba * (0.5 * sx + r3h * sy), 0, 1, [-1, 0], mode="add")
    return m


def chain3(t, delta, lmbd):                   # tests/test_examples/three_site/3site_cycle/run.py
    m = quiet(ref.tb_model, 1, 1, [[1.0]], [[0.0], [1.0 / 3.0], [2.0 / 3.0]])
    m.set_hop(t, 0, 1, [0])
    m.set_hop(t, 1, 2, [0])
    m.set_hop(t, 2, 0, [1])
    m.set_onsite([delta * -np.cos(2.0 * np.pi * (lmbd - i / 3.0)) for i in range(3)])
    return m


def per02():                                  # tests/test_tbmodel/test_dimr_dimk_different.py:10-17
    lat = [[3.0, 0.1, 0.4], [0.1, 3.1, 1.2], [0.8, 0.2, 3.5]]
    orb = [[0.3, 0.1, 0.2], [0.1, 0.8, 0.3], [0.2, 0.3, 0.4]]
    m = quiet(ref.tb_model, 2, 3, lat, orb, per=[0, 2])
    m.set_onsite([-2.3, 0.5, 0.1])
    m.set_hop(0.24, 0, 1, [1, 0, 2])
    m.set_hop(0.42, 0, 1, [3, 0, 2])
    m.set_hop(-0.12, 1, 2, [2, 0, 3])
    m.set_hop(-0.34, 2, 0, [-1, 0, 2])
    return m


def molecule():                               # 0-D model (tests/test_pythtb.py:31-44 style)
    m = quiet(ref.tb_model, 0, 1, [[1.0]], [[0.0], [0.5], [0.8]])
    m.set_onsite([0.1, -0.4, 0.7])
    m.set_hop(3.0, 0, 1)
    m.set_hop(0.5 + 0.25j, 1, 2)
    return m


def spin_chain():                             # 1-D spinor, dim_r=2, 2x2-matrix + 4-vector amplitudes
    m = quiet(ref.tb_model, 1, 2, [[1.0, 0.0], [0.3, 1.4]], [[0.1, 0.0], [0.6, 0.4], [0.35, 0.8]], per=[0], nspin=2)
    m.set_onsite([[0.3, 0.1, -0.2, 0.4], 0.5, [[-0.2, 0.1 - 0.3j], [0.1 + 0.3j, 0.6]]])
    m.set_hop([[0.4, 0.1j], [0.2, -0.3 + 0.1j]], 0, 1, [0, 0])
    m.set_hop([0.3, 0.05, 0.1, -0.2], 1, 2, [0, 0])
    m.set_hop(0.25 - 0.1j, 2, 0, [1, 0])
    m.set_hop([0.0, 0.1j, 0.0, 0.07], 1, 1, [2, 0])
    return m


def cubic16(seed=0):                          # SURVEY.md section 8d recipe
    rng = np.random.default_rng(seed)
    orb = rng.random((16, 3))
    m = quiet(ref.tb_model, 3, 3, np.identity(3), orb)
    m.set_onsite(np.where(np.arange(16) < 8, -2.0, 2.0) + 0.2 * rng.standard_normal(16))
    for i in range(16):
        for j in range(i + 1, 16):
            m.set_hop(0.1 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0, 0])
    for R in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
        for i in range(16):
            for j in range(16):
                m.set_hop(0.1 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    return m


def quad4():                                   # 4-D k-space (dim_k = dim_r = 4), 3 orbitals: the 4-D slicing paths
    rng = np.random.default_rng(3)
    m = quiet(ref.tb_model, 4, 4, np.identity(4), rng.random((3, 4)))
    m.set_onsite([-1.0, 0.2, 1.3])
    r = np.random.default_rng(5)
    for (i, j, R) in [(0, 1, [0, 0, 0, 0]), (1, 2, [0, 0, 0, 0]), (0, 2, [1, 0, 0, 0]), (0, 0, [0, 1, 0, 0]),
                      (1, 1, [0, 0, 1, 0]), (2, 2, [0, 0, 0, 1]), (0, 1, [0, 1, -1, 0]), (1, 2, [1, 0, 0, -1]),
                      (2, 0, [0, 0, 1, 1])]:
        m.set_hop(0.3 * (r.standard_normal() + 1j * r.standard_normal()), i, j, R)
    return m


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def tables(m, prefix="t_"):
    return {prefix + k: v for k, v in orc.model_tables(m).items()}


def pointwise(name, m, nk=64, seed=0):
    """tables + H(k) + eigenvalues at seeded random k (pins kernels 1 and 2)."""
    out = tables(m)
    n = m._nsta
    if m._dim_k == 0:
        out["ham"] = m._gen_ham().reshape(1, n, n)
        out["evals"] = m.solve_all().reshape(n, 1)
    else:
        k = np.random.default_rng(seed).uniform(-1.0, 1.0, size=(nk, m._dim_k))
        out["k"] = k
        out["ham"] = np.array([m._gen_ham(kk).reshape(n, n) for kk in k])
        out["evals"] = m.solve_all(k)
    save("point_" + name, **out)


def grid_case(name, m, mesh, start, occs, dirs_phase, flux_dirs=None):
    """solve_on_grid gaps + berry_flux / berry_phase in every flag combination."""
    w = ref.wf_array(m, mesh)
    gaps = w.solve_on_grid(start)
    out = tables(m)
    out["mesh"] = np.array(mesh)
    out["start_k"] = np.array(start, dtype=float)
    out["min_gaps"] = np.array(gaps if gaps is not None else [])
    for io_, occ in enumerate(occs):
        out["occ%d" % io_] = np.array(occ, dtype=int)
        for dirs in ([[0, 1]] if flux_dirs is None else flux_dirs):
            tag = "occ%d_d%d%d" % (io_, dirs[0], dirs[1])
            out["flux_plaq_" + tag] = np.array(w.berry_flux(occ, dirs=dirs, individual_phases=True))
            out["flux_tot_" + tag] = np.array(w.berry_flux(occ, dirs=dirs))
        for d in dirs_phase:
            for contin in (True, False):
                for be in (False, True):
                    out["phase_occ%d_dir%d_c%d_e%d" % (io_, d, int(contin), int(be))] = \
                        np.array(w.berry_phase(occ, d, contin=contin, berry_evals=be))
    save("grid_" + name, **out)


def kgen_case():
    out = {}
    g = graphene()
    path = [[0., 0.], [2. / 3., 1. / 3.], [.5, .5], [0., 0.]]
    kv, kd, kn = g.k_path(path, 121, report=False)          # examples/graphene.py:42-45
    out.update(g_path=np.array(path), g_kvec=kv, g_kdist=kd, g_knode=kn)
    path = [[0., 0.], [2. / 3., 1. / 3.], [.5, .5], [1. / 3., 2. / 3.], [0., 0.]]
    kv, kd, kn = g.k_path(path, 101, report=False)          # examples/kane_mele.py:61-64
    out.update(km_path=np.array(path), km_kvec=kv, km_kdist=kd, km_knode=kn)
    c = chain3(-1.0, 2.0, 0.0)
    for key in ("full", "fullc", "half"):
        kv, kd, kn = c.k_path(key, 17, report=False)
        out.update({"c_%s_kvec" % key: kv, "c_%s_kdist" % key: kd, "c_%s_knode" % key: kn})
    kv, kd, kn = c.k_path([[-0.5], [0.5]], 31, report=False)
    out.update(c_seg_kvec=kv, c_seg_kdist=kd, c_seg_knode=kn)
    p = per02()
    kv, kd, kn = p.k_path([[0.0, 0.0], [0.5, 0.0], [0.5, 0.5]], 40, report=False)
    out.update(p_kvec=kv, p_kdist=kd, p_knode=kn)
    out["mesh_4_6"] = g.k_uniform_mesh([4, 6])
    out["mesh_3_4_5"] = cubic16().k_uniform_mesh([3, 4, 5])
    out["mesh_7"] = c.k_uniform_mesh([7])
    save("kgen", **out)


def manual_cases():
    """Irregular / parametric wf_arrays filled through [] (cone and 3site_cycle
    run.py of the reference tests), with the inputs the checker needs to rebuild."""
    # cone: tests/test_examples/graphene/cone/run.py
    m = graphene(delta=-0.1)
    n = 31
    ang = 2.0 * np.pi * np.arange(n) / float(n - 1)
    kc = np.stack([np.cos(ang) * 0.05 + 1. / 3., np.sin(ang) * 0.05 + 2. / 3.], axis=1)
    w = ref.wf_array(m, [n])
    for i in range(n):
        w.solve_on_one_point(kc[i], i)
    w[-1] = w[0]
    out = tables(m)
    out["circ_k"] = kc
    out["circ_phase"] = np.array([w.berry_phase([0], 0), w.berry_phase([1], 0), w.berry_phase([0, 1], 0)])
    ws = ref.wf_array(m, [n, n])
    ks = np.zeros((n, n, 2))
    for i in range(n):
        for j in range(n):
            ks[i, j] = [0.1 * (-0.5 + i / float(n - 1)) + 1. / 3., 0.1 * (-0.5 + j / float(n - 1)) + 2. / 3.]
            ws[i, j] = m.solve_one(ks[i, j], eig_vectors=True)[1]
    out["sq_k"] = ks
    out["sq_flux"] = np.array([ws.berry_flux([0]), ws.berry_flux([1]), ws.berry_flux([0, 1])])
    out["sq_plaq"] = ws.berry_flux([0], individual_phases=True)
    save("manual_cone", **out)

    # 3site_cycle: k x lambda array, impose_pbc along k only
    t, delta, steps, nkp = -1.0, 2.0, 21, 31
    lam = np.linspace(0, 1, steps, endpoint=True)
    w = ref.wf_array(chain3(t, delta, 0.0), [nkp, steps])
    for il in range(steps):
        mm = chain3(t, delta, lam[il])
        kv, _, _ = mm.k_path([[-0.5], [0.5]], nkp, report=False)
        _, evec = mm.solve_all(kv, eig_vectors=True)
        for ik in range(nkp):
            w[ik, il] = evec[:, ik, :]
    w.impose_pbc(0, 0)
    save("manual_3site", t=t, delta=delta, lam=lam, nkp=nkp,
         wann=w.berry_phase([0], 0) / (2 * np.pi), flux=np.array(w.berry_flux([0])),
         flux_all=np.array([w.berry_flux([0]), w.berry_flux([1]), w.berry_flux([2]),
                            w.berry_flux([0, 1]), w.berry_flux([0, 1, 2])]))


def hwf_cases():
    """Position-operator / hybrid-Wannier path.  The finite models come from the
    reference's cut_piece/remove_orb (out of scope for the product), so their tables are part of
    the fixture; outputs follow tests/test_examples/slab/cubic_slab_hwf/run.py and
    tests/test_examples/haldane/haldane_hwf/run.py and are cross-checked here against the
    reference's own golden files."""
    tdir = "/root/reference/tests/test_examples"
    # --- cubic slab (17 orbitals, dim_k=2, dim_r=3)
    lat = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    bulk = quiet(ref.tb_model, 3, 3, lat, [[0.0, 0.0, 0.0], [0.5, 0.5, 0.5]])
    bulk.set_onsite([-1.0, 1.0])
    for lvec in ([-1, 0, 0], [0, 0, -1], [-1, -1, 0], [0, -1, -1]):
        bulk.set_hop(0.4, 0, 1, lvec)
    for lvec in ([0, 0, 0], [0, -1, 0], [-1, -1, -1], [-1, 0, -1]):
        bulk.set_hop(0.7, 0, 1, lvec)
    nl = 9
    slab = bulk.cut_piece(nl, 2, glue_edgs=False).remove_orb(2 * nl - 1)
    k1 = np.linspace(0.0, 1.0, 10, endpoint=False)
    kpts = np.array([[kx, ky] for kx in k1 for ky in k1])
    evals = slab.solve_all(kpts)
    nk = 9
    arr = ref.wf_array(slab, [nk, nk])
    arr.solve_on_grid([0.0, 0.0])
    hwf_arr = arr.empty_like(nsta_arr=nl)
    hwfc = np.zeros([nk, nk, nl])
    pexp = np.zeros([nk, nk, nl])
    for ix in range(nk):
        for iy in range(nk):
            val, vec = arr.position_hwf([ix, iy], occ=list(range(nl)), dir=2, hwf_evec=True, basis="orbital")
            hwfc[ix, iy] = val
            hwf_arr[ix, iy] = vec
            pexp[ix, iy] = arr.position_expectation([ix, iy], list(range(nl)), 2)
    hwf_arr.impose_pbc(0, 0)
    hwf_arr.impose_pbc(1, 1)
    px = np.array([hwf_arr.berry_phase(dir=0, occ=[n]) / (2.0 * np.pi) for n in range(nl)])
    xm = arr.position_matrix([2, 5], [0, 3, 4], 2)
    vw, ww = arr.position_hwf([2, 5], [0, 3, 4], 2, hwf_evec=True)           # default basis: wavefunction
    g = os.path.join(tdir, "slab/cubic_slab_hwf/golden_outputs")
    assert np.allclose(evals, np.load(os.path.join(g, "evals.npy")), rtol=1e-8, atol=1e-13)
    assert np.allclose(hwfc, np.load(os.path.join(g, "hwfc.npy")), rtol=1e-8, atol=1e-13)
    assert np.allclose(px, np.load(os.path.join(g, "px.npy")), rtol=1e-8, atol=1e-12)
    out = tables(slab)
    out.update(kpts=kpts, evals=evals, hwfc=hwfc, px=px, pexp=pexp, xmat_2_5=xm, hwfc_2_5=vw,
               hwfproj_2_5=np.abs(ww) ** 2)
    save("hwf_cubic_slab", **out)
    # --- Haldane ribbon (20 orbitals, dim_k=1, dim_r=2)
    hal = quiet(ref.tb_model, 2, 2, LAT, ORB)
    t, t2 = -1.0, 0.05 - 0.15j
    hal.set_onsite([0.2, -0.2])
    hal.set_hop(t, 0, 1, [0, 0])
    hal.set_hop(t, 1, 0, [1, 0])
    hal.set_hop(t, 1, 0, [0, 1])
    hal.set_hop(t2, 0, 0, [1, 0])
    hal.set_hop(t2, 1, 1, [1, -1])
    hal.set_hop(t2, 1, 1, [0, 1])
    hal.set_hop(t2.conjugate(), 1, 1, [1, 0])
    hal.set_hop(t2.conjugate(), 0, 0, [1, -1])
    hal.set_hop(t2.conjugate(), 0, 0, [0, 1])
    rib = hal.cut_piece(10, fin_dir=1, glue_edgs=False)
    kv, _, _ = rib.k_path([0.0, 0.5, 1.0], 100, report=False)
    rev, rvec = rib.solve_all(kv, eig_vectors=True)
    rev = rev - 0.25
    pos_exps = np.array([rib.position_expectation(rvec[:, i], dir=1) for i in range(rvec.shape[1])])
    nocc = np.array([int(np.sum(rev[:, i] < 0.0)) for i in range(rev.shape[1])])
    hw = [rib.position_hwf(rvec[rev[:, i] < 0.0, i], 1) for i in range(rvec.shape[1])]
    g = os.path.join(tdir, "haldane/haldane_hwf/golden_outputs")
    gold = np.load(os.path.join(g, "hwfcs.npy"), allow_pickle=True)
    assert all(np.allclose(a, b, rtol=1e-8, atol=1e-12) for a, b in zip(hw, gold))
    assert np.allclose(rev, np.load(os.path.join(g, "rib_eval.npy"), allow_pickle=True).astype(float), rtol=1e-8, atol=1e-12)
    out = tables(rib)
    out.update(k_vec=kv, rib_eval=rev, pos_exps=pos_exps, nocc=nocc, hwfcs_flat=np.concatenate(hw))
    save("hwf_haldane_ribbon", **out)


def full_size():
    out = {}
    t0 = time.time()
    m = haldane(0.2)
    ev = m.solve_all(m.k_uniform_mesh([256, 256]))            # config B recipe at 256^2 (1024^2: 105 s more)
    out["B256_sum"] = ev.sum(axis=1)
    out["B256_min"] = ev.min(axis=1)
    out["B256_max"] = ev.max(axis=1)
    print("B256", time.time() - t0)
    m = haldane(0.0)
    w = ref.wf_array(m, [2049, 2049])
    out["C_min_gaps"] = w.solve_on_grid([-0.5, -0.5])
    plaq = w.berry_flux([0], individual_phases=True)
    out["C_flux"] = np.array(plaq.sum())                   # == berry_flux([0]) (pythtb.py:3148)
    out["C_flux_row_sums"] = plaq.sum(axis=1)
    out["C_flux_absmax"] = np.array(np.abs(plaq).max())
    del w
    print("C", time.time() - t0, out["C_flux"] / (2 * np.pi))
    m = kane_mele("odd")
    w = ref.wf_array(m, [4097, 513])
    out["D_min_gaps"] = w.solve_on_grid([-0.5, -0.5])
    out["D_wan_cent"] = w.berry_phase([0, 1], dir=0, contin=False, berry_evals=True)
    print("D", time.time() - t0)
    save("full_size", **out)


def full_size_E():
    """BASELINE configs[4]'s recipe at the size the survey measured the reference on (cubic16 at 16^3,
    gap_78 0.345648, berry_phase(range(8), 2) -> (17, 17)); SURVEY.md 8c(7)."""
    t0 = time.time()
    m = cubic16()
    w = ref.wf_array(m, [17, 17, 17])
    out = {"E_min_gaps": w.solve_on_grid([0.0, 0.0, 0.0])}
    out["E_phase"] = w.berry_phase(range(8), 2)                       # (17, 17), contin=True (pythtb.py:3002-3025)
    out["E_phase_dir0_nocontin"] = w.berry_phase(range(8), 0, contin=False)
    out["E_flux01"] = w.berry_flux(range(8), dirs=[0, 1])             # (17,) (pythtb.py:3178-3186)
    ev = m.solve_all(m.k_uniform_mesh([16, 16, 16]))
    out["E_eval_sum"], out["E_eval_min"], out["E_eval_max"] = ev.sum(axis=1), ev.min(axis=1), ev.max(axis=1)
    print("E", time.time() - t0, out["E_min_gaps"][7])
    save("full_size_E", **out)


def display_reports():
    """Text printed by tb_model.display() for a few models (expected output only)."""
    import json
    out = {}
    flake = haldane(0.2).cut_piece(2, 0).cut_piece(2, 1)            # 0-D: hoppings without lattice vectors
    for name, m in (("haldane02", haldane(0.2)), ("km_odd", kane_mele("odd")), ("chain3", chain3(-1.0, 2.0, 0.3)),
                    ("per02", per02()), ("molecule", molecule()), ("haldane_flake_2x2", flake)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m.display()
        out[name] = buf.getvalue()
    with open(os.path.join(HERE, "display_reports.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("display_reports.json", {k: len(v) for k, v in out.items()})


def visualize_artists():
    """What tb_model.visualize() draws (line data, marker sizes, z-orders, colours, axis limits)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.colors as mcolors
    import matplotlib.pyplot as plt
    out = {}
    rng = np.random.default_rng(0)
    for name, m, dirs in (("haldane02", haldane(0.2), (0, 1)), ("chain3", chain3(-1.0, 2.0, 0.3), (0, None)),
                          ("flake", haldane(0.2).cut_piece(3, 0).cut_piece(2, 1), (1, 0))):
        e = rng.normal(size=m._norb) + 1j * rng.normal(size=m._norb)
        e /= np.linalg.norm(e)
        out[name + "/eig"] = e
        for scheme, vec in (("plain", None), ("wheel", e), ("red-blue", e), ("black", e)):
            fig, ax = m.visualize(dirs[0], dirs[1], eig_dr=vec, ph_color="black" if scheme == "plain" else scheme)
            rows = []
            for ln in ax.get_lines():
                x, y = np.asarray(ln.get_xdata(), float), np.asarray(ln.get_ydata(), float)
                pad = np.full(3 - len(x), np.nan)
                rows.append(np.concatenate([np.concatenate([x, pad]), np.concatenate([y, pad]),
                                            [ln.get_markersize(), ln.get_zorder(), ln.get_linewidth()],
                                            mcolors.to_rgba(ln.get_color())]))
            out["%s/%s/lines" % (name, scheme)] = np.array(rows)
            out["%s/%s/lims" % (name, scheme)] = np.array(list(ax.get_xlim()) + list(ax.get_ylim()))
            plt.close(fig)
    np.savez_compressed(os.path.join(HERE, "visualize_artists.npz"), **out)
    print("visualize_artists.npz", len(out))


if __name__ == "__main__":
    if "--visualize" in sys.argv:
        visualize_artists()
        sys.exit(0)
    if "--display" in sys.argv:
        display_reports()
        sys.exit(0)
    if "--quad4" in sys.argv:
        grid_case("quad4_4354", quad4(), [4, 3, 5, 4], [0.1, -0.2, 0.3, 0.05], [[0], [0, 1]], [],
                  flux_dirs=[[0, 1], [2, 3], [3, 1]])
        sys.exit(0)
    if "--full-E" in sys.argv:
        full_size_E()
        sys.exit(0)
    if "--full" in sys.argv:
        full_size()
        sys.exit(0)
    if "--hwf" in sys.argv:
        hwf_cases()
        sys.exit(0)
    pointwise("graphene", graphene())
    pointwise("haldane0", haldane(0.0))
    pointwise("haldane02", haldane(0.2))
    pointwise("km_odd", kane_mele("odd"))
    pointwise("km_even", kane_mele("even"))
    pointwise("chain3", chain3(-1.0, 2.0, 0.3))
    pointwise("per02", per02())
    pointwise("molecule", molecule())
    pointwise("spin_chain", spin_chain())
    pointwise("cubic16", cubic16(), nk=16)
    kgen_case()
    grid_case("haldane0_33", haldane(0.0), [33, 33], [-0.5, -0.5], [[0], [1], [0, 1]], [0, 1], [[0, 1], [1, 0]])
    grid_case("haldane02_20x28", haldane(0.2), [20, 28], [0.0, 0.1], [[0], [0, 1]], [0, 1])
    grid_case("km_odd_65x33", kane_mele("odd"), [65, 33], [-0.5, -0.5], [[0, 1], [0], [1, 2, 3]], [0, 1])
    grid_case("km_even_41", kane_mele("even"), [41, 41], [-0.5, -0.5], [[0, 1]], [1])
    grid_case("chain3_41", chain3(-1.0, 2.0, 0.3), [41], [0.0], [[0], [0, 1], [0, 1, 2]], [0], flux_dirs=[])
    grid_case("per02_11", per02(), [11, 11], [-0.5, -0.5], [[0], [0, 1]], [0, 1])
    grid_case("spin_chain_25", spin_chain(), [25], [-0.5], [[0, 1], [0, 1, 2]], [0], flux_dirs=[])
    grid_case("cubic16_9", cubic16(), [9, 9, 9], [0.0, 0.0, 0.0], [list(range(8)), [0, 3]], [0, 1, 2],
              [[0, 1], [1, 2], [2, 0]])
    manual_cases()
