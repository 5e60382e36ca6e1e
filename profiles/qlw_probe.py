"""A/B of the three-kernel tridiagonal path for n = 17..64 (tbk_solve_qlw.inl) against the Jacobi kernels (TBK_QLW=0):
accuracy against numpy.linalg.eigh on supplied matrices (random, zero, diagonal-degenerate, repeated +-1, banded) and
timings of batches of supplied matrices and of ribbon models on k lists / meshes."""
import contextlib, io, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child():
    import pythtb_amd as tb
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    out = {"qlw": os.environ.get("TBK_QLW", "1")}
    rng = np.random.default_rng(3)
    sizes = [int(s) for s in os.environ.get("QLW_SIZES", "17,24,31,32,33,40,48,63,64").split(",")]
    for n in sizes:
        nk = 4096
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = h + h.conj().transpose(0, 2, 1)
        h[5] = 0.0
        h[6] = np.diag(np.arange(n) % 3).astype(complex)
        h[7] = np.kron(np.eye(n // 2 + 1), [[0, 1], [1, 0]])[:n, :n]
        h[8] = np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)          # already tridiagonal
        h[9] = np.diag(np.full(n - 2, 1j), 2) + np.diag(np.full(n - 2, -1j), -2)  # two decoupled chains
        hh = np.ascontiguousarray(h)
        rec = {}
        if not os.environ.get("QLW_TIMING_ONLY"):
            ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
            _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hh.view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
            ref = np.linalg.eigvalsh(h).T
            V = vec.transpose(1, 0, 2)
            idx = list(range(0, 16)) + list(range(16, nk, 37))
            res = max(np.max(np.abs(h[i] @ V[i].T - V[i].T * ev[:, i])) for i in idx)
            orth = max(np.max(np.abs(V[i].conj() @ V[i].T - np.eye(n))) for i in idx)
            ev2 = np.zeros((n, nk))
            _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hh.view(float)), nk, _lib.dptr(ev2), None))
            rec = dict(eval_err=float(np.max(np.abs(ev - ref))), resid=float(res), orth=float(orth),
                       evalonly_err=float(np.max(np.abs(ev2 - ref))))
        # device-resident timings
        nkt = 16384 if n <= 40 else 8192
        ht = np.ascontiguousarray(np.tile(hh[16:16 + 1024], (nkt // 1024, 1, 1)))
        for vecs in (False, True):
            evt = np.zeros((n, nkt)); vt = np.zeros((n, nkt, n), dtype=complex) if vecs else None
            args = (ctx.handle, n, _lib.dptr(ht.view(float)), nkt, _lib.dptr(evt), _lib.dptr(vt.view(float)) if vecs else None)
            _lib.check(lib.tbk_eigh_batch(*args))
            ctx.prof_enable(1); ctx.prof_reset()
            _lib.check(lib.tbk_eigh_batch(*args))
            rec["kern_%d_%s_ms" % (nkt, "vec" if vecs else "val")] = round(ctx.prof_report()["eigh_batch"]["total_ms"], 3)
            ctx.prof_enable(0)
        out["n%d" % n] = rec
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in ("1", "0"):
            env = dict(os.environ, TBK_QLW=v)
            subprocess.run([sys.executable, __file__, "child"], env=env)
