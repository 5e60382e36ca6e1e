"""A/B of the eigenvalue-only path for n = 65..1024 (tridiagonalise with A in L2 + bisection, tbk_solve_trig.inl) against the
Jacobi solvers (TBK_TRIG=0): accuracy against numpy.linalg.eigvalsh and device-resident batch times."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child():
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    out = {"trig": os.environ.get("TBK_TRIG", "1")}
    rng = np.random.default_rng(3)
    cases = [(65, 512), (96, 512), (128, 512), (200, 256), (300, 101), (400, 64), (512, 16), (800, 8), (300, 1), (512, 1)]
    if os.environ.get("TRIG_CASES"):
        cases = [tuple(int(x) for x in c.split("x")) for c in os.environ["TRIG_CASES"].split(",")]
    for n, nk in cases:
        h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
        h = h + h.conj().transpose(0, 2, 1)
        if nk > 4:
            h[1] = np.diag(np.arange(n) % 5).astype(complex)
            h[2] = np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
            h[3] = 0.0
        hh = np.ascontiguousarray(h)
        ev = np.zeros((n, nk))
        args = (ctx.handle, n, _lib.dptr(hh.view(float)), nk, _lib.dptr(ev), None)
        _lib.check(lib.tbk_eigh_batch(*args))
        ref = np.linalg.eigvalsh(h).T
        err = float(np.max(np.abs(ev - ref)) / max(1.0, np.max(np.abs(ref))))
        ctx.prof_enable(1); ctx.prof_reset()
        _lib.check(lib.tbk_eigh_batch(*args))
        ms = ctx.prof_report()["eigh_batch"]["total_ms"]
        ctx.prof_enable(0)
        out["%dx%d" % (n, nk)] = dict(rel_err=err, kern_ms=round(ms, 3))
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in ("1", "0"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TBK_TRIG=v))
