"""A/B of the three-kernel form of the n = 9..16 direct solver (TBK_QL16_SPLIT=1: tridiagonalise | lane-per-matrix QL with a
rotation record | replay) against the single kernel: accuracy against numpy and timings on config E's 64^3 sub-mesh."""
import contextlib, io, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child():
    import pythtb_amd as tb
    from pythtb_amd import _lib
    import helpers as hp
    lib, ctx = _lib.lib, _lib.default_context()
    out = {"split": os.environ.get("TBK_QL16_SPLIT", "0")}
    rng = np.random.default_rng(3)
    for n in (9, 12, 13, 16):
        nk = 9000
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = h + h.conj().transpose(0, 2, 1)
        h[5] = 0.0
        h[6] = np.diag(np.arange(n) % 3).astype(complex)
        h[7] = np.kron(np.eye(n // 2 + 1), [[0, 1], [1, 0]])[:n, :n]
        h[8] = np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
        ev = np.zeros((n, nk)); vec = np.zeros((n, nk, n), dtype=complex)
        _lib.check(lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(np.ascontiguousarray(h).view(float)), nk, _lib.dptr(ev), _lib.dptr(vec.view(float))))
        ref = np.linalg.eigvalsh(h).T
        V = vec.transpose(1, 0, 2)
        idx = list(range(0, 12)) + list(range(12, nk, 97)) + [nk - 1]
        res = max(np.max(np.abs(h[i] @ V[i].T - V[i].T * ev[:, i])) for i in idx)
        orth = max(np.max(np.abs(V[i].conj() @ V[i].T - np.eye(n))) for i in idx)
        out["n%d" % n] = dict(eval_err=float(np.max(np.abs(ev - ref))), resid=float(res), orth=float(orth))
    with contextlib.redirect_stdout(io.StringIO()):
        m = hp.cubic16(tb.tb_model)
    w = tb.wf_array(m, [65, 65, 65])
    w.solve_on_grid([0, 0, 0]); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin(); g = w.solve_on_grid([0, 0, 0]); best = min(best, ctx.timer_end())
    out["E64_solve_ms"] = best
    out["gap78"] = float(g[7])
    ph = w.berry_phase(range(8), 2, contin=False)
    out["phase_checksum"] = float(np.sum(np.cos(ph)))
    k = np.random.default_rng(0).random((262144, 3))
    m.solve_all(k[:16384], eig_vectors=True)
    t0 = time.perf_counter(); r = m.solve_all(k, eig_vectors=True); out["list262144_vec_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in ("1", "0"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TBK_QL16_SPLIT=v))
