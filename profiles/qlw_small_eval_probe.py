"""Eigenvalue-only solves of n = 17..64 at small and large batch sizes: Jacobi (TBK_QLW=0), tridiagonalise + lane-per-matrix QL
(TBK_QLW_BISECT=0), tridiagonalise + bisection (TBK_QLW_BISECT=1).  Device-resident kernel time per batch and accuracy."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child():
    from pythtb_amd import _lib
    lib, ctx = _lib.lib, _lib.default_context()
    out = {"cfg": os.environ.get("CFG")}
    rng = np.random.default_rng(3)
    for n in (20, 32, 48, 64):
        for nk in (1, 128, 1024, 4096, 16384):
            base = rng.standard_normal((min(nk, 256), n, n)) + 1j * rng.standard_normal((min(nk, 256), n, n))
            base = base + base.conj().transpose(0, 2, 1)
            hh = np.ascontiguousarray(np.tile(base, (max(1, nk // 256), 1, 1))[:nk])
            ev = np.zeros((n, nk))
            args = (ctx.handle, n, _lib.dptr(hh.view(float)), nk, _lib.dptr(ev), None)
            _lib.check(lib.tbk_eigh_batch(*args))
            err = float(np.max(np.abs(ev[:, :min(nk, 256)] - np.linalg.eigvalsh(base).T)))
            ctx.prof_enable(1); ctx.prof_reset()
            _lib.check(lib.tbk_eigh_batch(*args))
            ms = ctx.prof_report()["eigh_batch"]["total_ms"]
            ctx.prof_enable(0)
            out["%dx%d" % (n, nk)] = [round(ms, 3), float("%.1e" % err)]
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for cfg, env in (("jacobi", {"TBK_QLW": "0"}), ("ql", {"TBK_QLW_BISECT": "0"}), ("bisect", {"TBK_QLW_BISECT": "1"})):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CFG=cfg, **env))
