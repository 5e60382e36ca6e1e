"""A/B of the direct n = 9..16 solver (Householder + implicit QL in registers, tbk_solve_ql16.inl) against the
Jacobi kernels (TBK_QL16=0): accuracy against numpy.linalg.eigh and timings on config E's 64^3 sub-mesh."""
import contextlib, io, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))

def child():
    import ctypes as C
    import pythtb_amd as tb
    from pythtb_amd import _lib
    import helpers as hp
    lib, ctx = _lib.lib, _lib.default_context()
    out = {"ql16": os.environ.get("TBK_QL16", "1")}
    rng = np.random.default_rng(3)
    for n in (9, 12, 13, 15, 16):
        nk = 4096
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = h + h.conj().transpose(0, 2, 1)
        h[5] = 0.0                                  # zero matrix
        h[6] = np.diag(np.arange(n) % 3).astype(complex)   # degenerate, already diagonal
        h[7] = np.kron(np.eye(n // 2 + 1), [[0, 1], [1, 0]])[:n, :n]   # repeated +-1
        ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
        _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(np.ascontiguousarray(h).view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
        ref = np.linalg.eigvalsh(h).T
        V = vec.transpose(1, 0, 2)                  # [k][band][comp]
        res = max(np.max(np.abs(h[i] @ V[i].T - V[i].T * ev[:, i])) for i in range(0, nk, 7))
        orth = max(np.max(np.abs(V[i].conj() @ V[i].T - np.eye(n))) for i in range(0, nk, 7))
        ev2 = np.zeros((n, nk))
        _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(np.ascontiguousarray(h).view(float)), nk, _lib.dptr(ev2), None))
        out["n%d" % n] = dict(eval_err=float(np.max(np.abs(ev - ref))), resid=float(res), orth=float(orth),
                              evalonly_err=float(np.max(np.abs(ev2 - ref))))
    with contextlib.redirect_stdout(io.StringIO()):
        m = hp.cubic16(tb.tb_model)
    w = tb.wf_array(m, [65, 65, 65])
    w.solve_on_grid([0, 0, 0]); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin(); g = w.solve_on_grid([0, 0, 0]); best = min(best, ctx.timer_end())
    out["E64_solve_ms"] = best
    out["gap78"] = float(g[7])
    t0 = time.perf_counter(); ph = w.berry_phase(range(8), 2, contin=False); out["E64_phase_ms"] = (time.perf_counter() - t0) * 1e3
    out["phase_checksum"] = float(np.sum(np.cos(ph)))
    k = np.random.default_rng(0).random((262144, 3))
    for vecs in (False, True):
        m.solve_all(k[:4096], eig_vectors=vecs)
        t0 = time.perf_counter(); r = m.solve_all(k, eig_vectors=vecs); out["list262144_%s_ms" % ("vec" if vecs else "val")] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in ("1", "0"):
            env = dict(os.environ, TBK_QL16=v)
            subprocess.run([sys.executable, __file__, "child"], env=env)
