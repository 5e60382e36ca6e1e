#!/usr/bin/env python3
"""Development-container only: differential fuzz of the HOST-side API against the imported reference
(random models, random setter sequences, random transform chains; tables and exceptions must agree).

    python3 tests/golden/diff_fuzz_host.py [seed] [cases]
"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
with contextlib.redirect_stdout(io.StringIO()):
    import pythtb as ref  # noqa: E402
import pythtb_amd as mine  # noqa: E402
from oracle import tb_oracle as orc  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(*a, **k)


def both(fn_name, objs, *a, **k):
    """Call the same method on both objects; return (results, raised) with raised a pair of bools."""
    res, exc = [], []
    for o in objs:
        try:
            res.append(quiet(getattr(o, fn_name), *a, **k))
            exc.append(False)
        except Exception:
            res.append(None)
            exc.append(True)
    return res, exc


def same_tables(a, b):
    ta, tb_ = orc.model_tables(a), orc.model_tables(b)
    for key in ta:
        if np.shape(ta[key]) != np.shape(tb_[key]) or not np.allclose(ta[key], tb_[key], rtol=0, atol=1e-13):
            return key
    return None


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncase):
    dim_r = int(rng.integers(1, 4))
    dim_k = int(rng.integers(1, dim_r + 1))
    nspin = int(rng.integers(1, 3))
    norb = int(rng.integers(1, 5))
    lat = np.identity(dim_r) + 0.2 * rng.standard_normal((dim_r, dim_r))
    if np.linalg.det(lat) < 0:
        lat[0] *= -1
    orb = rng.uniform(-0.3, 1.3, size=(norb, dim_r))
    per = sorted(rng.choice(dim_r, size=dim_k, replace=False).tolist())
    pair = [quiet(cls, dim_k, dim_r, lat, orb, per=per, nspin=nspin) for cls in (ref.tb_model, mine.tb_model)]
    log = []
    for _ in range(int(rng.integers(2, 12))):        # random setter calls, legal and illegal
        if rng.random() < 0.3:
            val = float(rng.standard_normal()) if nspin == 1 or rng.random() < 0.5 else list(rng.standard_normal(4))
            args = (val, int(rng.integers(-1, norb + 1)))
            kw = {"mode": str(rng.choice(["set", "reset", "add", "bogus"]))}
            _, exc = both("set_onsite", pair, *args, **kw)
            log.append(("set_onsite", args, kw))
        else:
            amp = complex(rng.standard_normal(), rng.standard_normal())
            if nspin == 2 and rng.random() < 0.5:
                amp = list(rng.standard_normal(4))
            R = [int(x) for x in rng.integers(-2, 3, size=dim_r)]
            for d in range(dim_r):
                if d not in per:
                    R[d] = 0
            args = (amp, int(rng.integers(0, norb)), int(rng.integers(0, norb)), R)
            kw = {"mode": str(rng.choice(["set", "reset", "add"])), "allow_conjugate_pair": bool(rng.integers(0, 2))}
            _, exc = both("set_hop", pair, *args, **kw)
            log.append(("set_hop", args, kw))
        if exc[0] != exc[1]:
            print("case", case, "EXCEPTION MISMATCH", log[-1], exc)
            bad += 1
    for _ in range(int(rng.integers(0, 4))):         # random transform chain
        m_ref, m_mine = pair
        choice = str(rng.choice(["cut_piece", "reduce_dim", "make_supercell", "remove_orb", "change_nonperiodic_vector"]))
        if choice == "cut_piece":
            args, kw = (int(rng.integers(1, 4)), int(rng.integers(0, dim_r))), {"glue_edgs": bool(rng.integers(0, 2))}
        elif choice == "reduce_dim":
            args, kw = (int(rng.integers(0, dim_r)), float(rng.uniform(-1, 1))), {}
        elif choice == "make_supercell":
            S = np.identity(m_ref._dim_r, dtype=int)
            for a in m_ref._per:
                for b in m_ref._per:
                    S[a, b] = int(rng.integers(-1, 3)) if a != b else int(rng.integers(1, 3))
            args, kw = (S.tolist(),), {"to_home": bool(rng.integers(0, 2)), "to_home_suppress_warning": bool(rng.integers(0, 2))}
        elif choice == "remove_orb":
            args, kw = (int(rng.integers(0, m_ref._norb + 1)),), {}
        else:
            args, kw = (int(rng.integers(0, dim_r)),), {"to_home": bool(rng.integers(0, 2)), "to_home_suppress_warning": True}
        res, exc = both(choice, pair, *args, **kw)
        if exc[0] != exc[1]:
            print("case", case, "EXCEPTION MISMATCH in", choice, args, kw, exc)
            bad += 1
            break
        if exc[0]:
            continue
        if choice == "make_supercell" and np.abs(np.linalg.det(np.array(args[0]))) > 6:
            pass
        pair = res
        key = same_tables(pair[0], pair[1])
        if key:
            print("case", case, "TABLE MISMATCH after", choice, args, kw, "in", key)
            bad += 1
            break
        if pair[0]._norb > 40:
            break
    key = same_tables(pair[0], pair[1])
    if key:
        print("case", case, "FINAL TABLE MISMATCH in", key)
        bad += 1
    if pair[0]._dim_k in (1, 2, 3) and case % 3 == 0:     # k generators
        mesh = [int(rng.integers(1, 5)) for _ in range(pair[0]._dim_k)]
        res, exc = both("k_uniform_mesh", pair, mesh)
        if exc[0] != exc[1] or (not exc[0] and not np.array_equal(res[0], res[1])):
            print("case", case, "k_uniform_mesh mismatch")
            bad += 1
        nodes = rng.uniform(-1, 1, size=(int(rng.integers(2, 5)), pair[0]._dim_k))
        res, exc = both("k_path", pair, nodes.tolist(), int(rng.integers(2, 40)), report=False)
        if exc[0] != exc[1] or (not exc[0] and not all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))):
            print("case", case, "k_path mismatch")
            bad += 1
    # signed band lists (round 6): the reference indexes `_wfs[..., occ, :]` with NumPy in berry_phase / berry_flux
    # (pythtb.py:2981, :2989-2996, :3141) -- which lists raise IndexError, and which bands the others select, must agree with
    # wf_array._wrap_occ.  (The reference runs its Berry call on an array of random vectors; mine only wraps: no GPU here.)
    if pair[0]._dim_k in (1, 2) and 1 <= pair[0]._nsta <= 12 and case % 2 == 0:       # (a model all of whose orbitals were removed has no bands to index)
        mesh = [3] * pair[0]._dim_k
        wr, wm = ref.wf_array(pair[0], mesh), mine.wf_array(pair[1], mesh)
        wr._wfs[...] = rng.standard_normal(wr._wfs.shape) + 1j * rng.standard_normal(wr._wfs.shape)
        nst = pair[0]._nsta
        for _ in range(4):
            occ = [int(x) for x in rng.integers(-nst - 2, nst + 2, size=int(rng.integers(1, 4)))]
            try:
                quiet(wr.berry_phase, occ, 0, contin=False)
                r_exc = None
            except Exception as e:                      # noqa: BLE001
                r_exc = type(e)
            try:
                got = wm._wrap_occ(wm._occ(occ))
                m_exc = None
            except Exception as e:                      # noqa: BLE001
                got, m_exc = None, type(e)
            if (r_exc is IndexError) != (m_exc is IndexError) or (m_exc is None and not np.array_equal(got, np.arange(nst)[occ])):
                print("case", case, "occ", occ, "reference", r_exc, "mine", m_exc, got, "nsta", pair[0]._nsta, pair[1]._nsta, wr._nsta_arr, wm._nsta_arr, pair[0]._norb, pair[0]._nspin)
                bad += 1
print("cases", ncase, "mismatches", bad)

# the Wannier90 importer with random options (tables and exceptions must agree)
w90_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "w90_silicon")
if os.path.isdir(w90_dir):
    readers = [quiet(mod.w90, w90_dir, "silicon") for mod in (ref, mine)]
    wbad = 0
    for case in range(12):
        kw = {"zero_energy": float(rng.uniform(-5, 5)),
              "min_hopping_norm": None if rng.random() < 0.3 else float(10 ** rng.uniform(-4, 0)),
              "max_distance": None if rng.random() < 0.3 else float(rng.uniform(1, 12)),
              "ignorable_imaginary_part": None if rng.random() < 0.5 else float(10 ** rng.uniform(-9, -3))}
        res, exc = both("model", readers, **kw)
        if exc[0] != exc[1] or (not exc[0] and same_tables(res[0], res[1])):
            print("w90 case", case, kw, "MISMATCH", exc)
            wbad += 1
    res, exc = both("dist_hop", readers)
    if not all(np.array_equal(x, y) for x, y in zip(res[0], res[1])):
        print("w90 dist_hop MISMATCH")
        wbad += 1
    print("w90 cases 12 mismatches", wbad)
