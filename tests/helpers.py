"""Model builders for the tests, written against the public tb_model API (the same
calls the reference's example scripts make), plus a rebuild-from-fixture helper."""
import contextlib
import io

import numpy as np

LAT = [[1.0, 0.0], [0.5, np.sqrt(3.0) / 2.0]]
ORB = [[1.0 / 3.0, 1.0 / 3.0], [2.0 / 3.0, 2.0 / 3.0]]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def model_from_tables(cls, t):
    """Rebuild a model through set_onsite/set_hop from fixture tables."""
    dim_k, dim_r, nspin = int(t["dim_k"]), int(t["dim_r"]), int(t["nspin"])
    m = quiet(cls, dim_k, dim_r, t["lat"], t["orb"], per=[int(p) for p in t["per"]], nspin=nspin)
    if nspin == 1:
        m.set_onsite([float(np.real(x)) for x in t["site_energies"]])
    else:
        m.set_onsite([np.array(x) for x in t["site_energies"]])
    for h in range(len(t["hop_i"])):
        amp = complex(t["hop_amp"][h][0, 0]) if nspin == 1 else np.array(t["hop_amp"][h])
        if dim_k == 0:
            m.set_hop(amp, int(t["hop_i"][h]), int(t["hop_j"][h]), allow_conjugate_pair=True)
        else:
            m.set_hop(amp, int(t["hop_i"][h]), int(t["hop_j"][h]), [int(x) for x in t["hop_R"][h]],
                      allow_conjugate_pair=True)
    return m


def graphene(cls, delta=0.0, t=-1.0):
    m = quiet(cls, 2, 2, LAT, ORB)
    m.set_onsite([-delta, delta])
    m.set_hop(t, 0, 1, [0, 0])
    m.set_hop(t, 1, 0, [1, 0])
    m.set_hop(t, 1, 0, [0, 1])
    return m


def haldane(cls, delta=0.0, t=-1.0, t2abs=0.15):
    m = quiet(cls, 2, 2, LAT, ORB)
    t2 = t2abs * np.exp(1j * np.pi / 2.0)
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


def kane_mele(cls, topological="odd"):
    m = quiet(cls, 2, 2, LAT, ORB, nspin=2)
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


def chain3(cls, t, delta, lmbd):
    m = quiet(cls, 1, 1, [[1.0]], [[0.0], [1.0 / 3.0], [2.0 / 3.0]])
    m.set_hop(t, 0, 1, [0])
    m.set_hop(t, 1, 2, [0])
    m.set_hop(t, 2, 0, [1])
    m.set_onsite([delta * -np.cos(2.0 * np.pi * (lmbd - i / 3.0)) for i in range(3)])
    return m


def cubic16(cls, seed=0):
    rng = np.random.default_rng(seed)
    orb = rng.random((16, 3))
    m = quiet(cls, 3, 3, np.identity(3), orb)
    m.set_onsite(np.where(np.arange(16) < 8, -2.0, 2.0) + 0.2 * rng.standard_normal(16))
    for i in range(16):
        for j in range(i + 1, 16):
            m.set_hop(0.1 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, [0, 0, 0])
    for R in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
        for i in range(16):
            for j in range(16):
                m.set_hop(0.1 * (rng.standard_normal() + 1j * rng.standard_normal()), i, j, R)
    return m


def random_model(cls, norb, dim_k, nspin, seed, nhop=None, rmax=2):
    """Seeded random hermitian model with a mix of amplitude formats."""
    rng = np.random.default_rng(seed)
    dim_r = max(dim_k, 1)
    lat = np.identity(dim_r) + 0.1 * rng.random((dim_r, dim_r))
    if np.linalg.det(lat) < 0:
        lat[0] *= -1
    orb = rng.random((norb, dim_r))
    m = quiet(cls, dim_k, dim_r, lat, orb, nspin=nspin)
    if nspin == 1:
        m.set_onsite(list(rng.standard_normal(norb)))
    else:
        m.set_onsite([list(rng.standard_normal(4)) for _ in range(norb)])
    nhop = nhop if nhop is not None else 3 * norb
    seen = set()
    tries = 0
    while len(seen) < nhop and tries < 50 * nhop:
        tries += 1
        i, j = int(rng.integers(norb)), int(rng.integers(norb))
        R = tuple(int(x) for x in rng.integers(-rmax, rmax + 1, size=dim_r)) if dim_k > 0 else ()
        if dim_k > 0:
            R = tuple(R[d] if d < dim_k else 0 for d in range(dim_r))
        if i == j and all(r == 0 for r in R):
            continue
        key, ckey = (i, j, R), (j, i, tuple(-r for r in R))
        if key in seen or ckey in seen:
            continue
        seen.add(key)
        if nspin == 1:
            amp = complex(rng.standard_normal(), rng.standard_normal())
        else:
            amp = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
        if dim_k == 0:
            m.set_hop(amp, i, j)
        else:
            m.set_hop(amp, i, j, list(R))
    return m
