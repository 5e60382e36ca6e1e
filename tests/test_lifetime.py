"""Lifetime stress for the host-side resource code of libtbk (after the blob-pool bug of 1fe3375): a long
seeded random sequence of the operations that allocate, park, reuse and free device and mapped-host memory -- model edits
(blobs parked in the context's pool), deep copies, small solves through the mapped buffer and large ones through scratch,
growth of the mapped buffer, wf_array creation / solve / Berry calls / frees, a second context created and destroyed -- with
every result checked against the oracle (or against an earlier result of the same call) at every step."""
import copy
import gc

import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def _models(tb):
    return [hp.haldane(tb.tb_model, 0.2), hp.kane_mele(tb.tb_model, "odd"), hp.chain3(tb.tb_model, -1.0, 2.0, 0.3),
            hp.random_model(tb.tb_model, 3, 2, 1, 5), hp.random_model(tb.tb_model, 5, 1, 1, 6), hp.random_model(tb.tb_model, 2, 2, 2, 7)]


@pytest.mark.parametrize("seed", [0, 1])
def test_random_lifetime_sequence_against_the_oracle(tb, seed):
    from oracle import tb_oracle as orc
    from pythtb_amd import _lib
    rng = np.random.default_rng(1000 + seed)
    models = _models(tb)
    arrays = []                                   # (wf_array, oracle's array, mesh, occupied bands) kept alive for a while
    GAPPED = {0: 1, 1: 2, 2: 1}                   # haldane: lowest band; kane_mele: lowest two; 3-site chain: lowest
    nsteps = 1000
    done = {"edit": 0, "copy": 0, "small": 0, "large": 0, "vec": 0, "grid": 0, "berry": 0, "free": 0, "ctx": 0, "ham": 0}
    for step in range(nsteps):
        op = rng.integers(0, 10)
        mi = int(rng.integers(0, len(models)))
        m = models[mi]
        d = m._dim_k
        if op == 0:                               # edit a model: its device blob is replaced, the old one parked
            if mi in GAPPED:                      # (the textbook models keep their gaps: a small shift of one on-site energy)
                if m._nspin == 1:
                    m.set_onsite(0.02 * float(rng.standard_normal()), int(rng.integers(0, m._norb)), mode="add")
                else:
                    m.set_onsite([0.02 * float(rng.standard_normal()), 0.0, 0.0, 0.0], int(rng.integers(0, m._norb)), mode="add")
            elif m._nspin == 1:
                m.set_onsite(float(rng.standard_normal()), int(rng.integers(0, m._norb)), mode="reset")
            else:
                m.set_onsite([float(rng.standard_normal()), 0.0, 0.0, 0.1], int(rng.integers(0, m._norb)), mode="reset")
            done["edit"] += 1
        elif op == 1:                             # deep copy replaces the model (the copy owns no device blob yet)
            models[mi] = copy.deepcopy(m)
            done["copy"] += 1
        elif op in (2, 3):                        # small solve (mapped host buffer), eigenvalues or with vectors
            nk = int(rng.integers(1, 40))
            k = rng.random((nk, d))
            if op == 2:
                ev = m.solve_all(k)
                assert np.max(np.abs(ev - orc.solve_all_vec(m, k))) < 1e-11, step
                done["small"] += 1
            else:
                ev, vec = m.solve_all(k, eig_vectors=True)
                assert np.max(np.abs(ev - orc.solve_all_vec(m, k))) < 1e-11, step
                H = np.array([orc.gen_ham(m, kk) for kk in k]).reshape(nk, m._nsta, m._nsta)
                V = vec.reshape(m._nsta, nk, m._nsta)
                res = np.abs(np.einsum("kij,bkj->bki", H, V) - ev[:, :, None] * V).max()
                assert res < 1e-10, (step, res)
                done["vec"] += 1
        elif op == 4:                             # large solve: grows scratch / the mapped buffer past its first size
            nk = int(rng.integers(3000, 40000))
            k = rng.random((nk, d))
            ev = m.solve_all(k)
            samp = rng.integers(0, nk, 16)
            assert np.max(np.abs(ev[:, samp] - orc.solve_all_vec(m, k[samp]))) < 1e-11, step
            done["large"] += 1
        elif op == 5:                             # one Hamiltonian through the mapped buffer
            kk = rng.random(d)
            assert np.max(np.abs(m._gen_ham(kk).reshape(m._nsta, m._nsta) - orc.gen_ham(m, kk).reshape(m._nsta, m._nsta))) < 1e-12, step
            done["ham"] += 1
        elif op == 6 and d >= 1:                  # a new wf_array, solved
            mesh = [int(rng.integers(8, 15)) for _ in range(d)]
            start = list(rng.random(d))
            w = tb.wf_array(m, mesh)
            gaps = w.solve_on_grid(start)
            owfs, ogaps = orc.solve_on_grid(w._model, mesh, start, vectorised=True)
            if gaps is not None:
                assert np.max(np.abs(gaps - ogaps)) < 1e-10, step
            # Berry quantities are compared on the gapped textbook models only (occupied set = GAPPED[mi] lowest bands): between
            # the coarse points of a random model neighbouring occupied subspaces can be nearly orthogonal and the phase of a
            # vanishing determinant means nothing
            if mi in GAPPED and float(np.min(ogaps[GAPPED[mi] - 1])) > 0.05:
                arrays.append((w, owfs, mesh, GAPPED[mi]))
            done["grid"] += 1
        elif op == 7 and arrays:                  # Berry quantities of an array made earlier (its model may have been edited since:
            w, owfs, mesh, nocc = arrays[int(rng.integers(0, len(arrays)))]   # the array holds its own deep copy)
            nd = len(mesh)
            occ = list(range(nocc))
            dr = int(rng.integers(0, nd))
            got = np.asarray(w.berry_phase(occ, dr if nd > 1 else None, contin=False))
            ref = np.asarray(orc.berry_phase(owfs, nd, occ, dr, contin=False))
            assert np.max(np.abs((got - ref + np.pi) % (2 * np.pi) - np.pi)) < 1e-8, step
            if nd >= 2:
                f = np.asarray(w.berry_flux(occ, [0, 1], individual_phases=True))
                fr = np.asarray(orc.berry_flux(owfs, nd, occ, [0, 1], individual_phases=True, vectorised=True))
                assert np.max(np.abs((f - fr + np.pi) % (2 * np.pi) - np.pi)) < 1e-8, step
            done["berry"] += 1
        elif op == 8 and arrays:                  # free an array (its device buffers go), sometimes all of them
            if rng.random() < 0.2:
                arrays.clear()
            else:
                arrays.pop(int(rng.integers(0, len(arrays))))
            gc.collect()
            done["free"] += 1
        elif op == 9 and step % 7 == 0:           # a second context comes and goes (its own pools and mapped buffer)
            import ctypes as C
            lib = _lib.lib
            c2 = _lib.Context(0)
            assert c2.info()["compute_units"] > 0
            orb, onsite, hi, hj, hR, amp = m._flat_tables()
            k = np.ascontiguousarray(rng.random((5, max(d, 1))))
            ref = m.solve_all(k if d else None) if d else None
            for _ in range(3):                                   # upload / small solve / free on the second context: its own pool
                hm = C.c_void_p()
                _lib.check(lib.tbk_model_upload(c2.handle, m._dim_k, m._norb, m._nspin, _lib.dptr(orb), _lib.dptr(onsite.view(float)), len(hi),
                                                _lib.iptr(hi), _lib.iptr(hj), _lib.iptr(hR.reshape(-1)) if hR.size else None,
                                                _lib.dptr(amp.view(float)) if amp.size else None, C.byref(hm)))
                if d:
                    ev = np.zeros((m._nsta, 5))
                    _lib.check(lib.tbk_solve_list(hm, _lib.dptr(k), 5, _lib.dptr(ev), None))
                    assert np.array_equal(ev, ref), step
                _lib.check(lib.tbk_model_free(hm))
            c2.sync()
            _lib.check(lib.tbk_ctx_destroy(c2.handle))
            done["ctx"] += 1
        if len(arrays) > 12:
            arrays.pop(0)
    assert min(done[k] for k in ("edit", "copy", "small", "large", "vec", "grid", "berry", "free")) > 20 and done["ctx"] > 3, done
