"""Round 6: fuzz of the 17..32-state path on supplied matrices with STRUCTURE -- direct sums of random Hermitian blocks (T splits),
identical blocks (levels degenerate across blocks), blocks coupled by 1e-18 .. 1e-6 (nearly split), low-rank perturbations of the
identity, and random diagonal similarity by phases; eigenvalues against numpy.linalg.eigvalsh, residual and orthonormality of the
vectors, and the number of matrices that went to the replay list.   python profiles/tw32_fuzz.py [matrices per case = 600]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pythtb_amd import _lib
ctx = _lib.default_context()
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(2026)
def herm(k, n):
    a = rng.standard_normal((k, n, n)) + 1j * rng.standard_normal((k, n, n))
    return a + a.conj().transpose(0, 2, 1)
def solve(h):
    n = h.shape[1]
    ev = np.zeros((n, len(h))); vec = np.zeros((n, len(h), n), dtype=complex)
    hc = np.ascontiguousarray(h)
    _lib.check(_lib.lib.tbk_eigh_batch(ctx.handle, n, _lib.dptr(hc.view(float)), len(h), _lib.dptr(ev), _lib.dptr(vec.view(float))))
    return ev, vec
worst = {}
with _lib.knob("TBK_QLW_MIN", 0):
    for n in range(17, 33):
        cases = {}
        cut = int(rng.integers(3, n - 3))
        h = np.zeros((nk, n, n), dtype=complex); h[:, :cut, :cut] = herm(nk, cut); h[:, cut:, cut:] = herm(nk, n - cut)
        cases["two blocks"] = h
        half = n // 2
        b = herm(nk, half)
        h = np.zeros((nk, n, n), dtype=complex); h[:, :half, :half] = b; h[:, half:2 * half, half:2 * half] = b
        if n % 2: h[:, -1, -1] = rng.standard_normal(nk)
        cases["identical blocks"] = h
        h = cases["two blocks"].copy()
        eps = 10.0 ** rng.uniform(-18, -6, nk)
        h[:, cut - 1, cut] = eps; h[:, cut, cut - 1] = eps
        cases["nearly split"] = h
        u = rng.standard_normal((nk, n, 2)) + 1j * rng.standard_normal((nk, n, 2))
        cases["identity + rank 2"] = np.eye(n)[None] * 2.5 + u @ u.conj().transpose(0, 2, 1)
        ph = np.exp(2j * np.pi * rng.random((nk, n)))
        cases["identical blocks, phases"] = ph[:, :, None] * cases["identical blocks"] * ph.conj()[:, None, :]
        for name, h in cases.items():
            ctx.solver_stats(reset=True)
            ev, vec = solve(h)
            listed = ctx.solver_stats(reset=True)["listed_matrices"]
            ref = np.linalg.eigvalsh(h).T
            nrm = np.maximum(np.abs(ref).max(axis=0), 1e-300)
            V = vec.transpose(1, 0, 2)
            de = (np.abs(ev - ref) / nrm).max()
            res = (np.abs(np.einsum("kij,kbj->kbi", h, V) - V * ev.T[:, :, None]).reshape(nk, -1).max(axis=1) / nrm).max()
            orth = np.abs(np.einsum("kbi,kci->kbc", V.conj(), V) - np.eye(n)).reshape(nk, -1).max(axis=1).max()
            w = worst.setdefault(name, [0.0, 0.0, 0.0, 0])
            w[0] = max(w[0], de); w[1] = max(w[1], res); w[2] = max(w[2], orth); w[3] += listed
for name, w in worst.items():
    print("%-26s eigenvalues %.1e  residual %.1e  orthonormality %.1e  listed %d of %d" % (name, w[0], w[1], w[2], w[3], nk * 16))
