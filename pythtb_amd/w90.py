"""w90: Wannier90 -> tb_model importer (reference pythtb.py:3208-3759).

Reads prefix.win (unit cell), prefix_hr.dat (H(R) with Wigner-Seitz degeneracies),
prefix_centres.xyz (Wannier centres) and optionally prefix_band.kpt / prefix_band.dat, and
builds the large, long-ranged hopping tables (hundreds to 1e5 terms) that the device kernels
stream per k-point.  Host-side parsing only; the attribute names (`lat`, `num_wan`, `ham_r`,
`xyz_cen`, `red_cen`) and the order of the generated hoppings follow the reference so that the
resulting tb_model tables are identical (tests/test_w90.py compares them with tables captured
from the reference on its silicon example).
"""
import numpy as np

from .model import tb_model

__all__ = ["w90"]

_BOHR = 0.5291772108      # Angstrom, the constant the reference uses (pythtb.py:3323)


def _lines(path):
    with open(path, "r") as f:
        return f.readlines()


class w90(object):
    """Wannier90 output `path/prefix*` read into memory; `model()` turns it into a tb_model."""

    def __init__(self, path, prefix):
        self.path = path
        self.prefix = prefix
        stem = self.path + "/" + self.prefix                    # same concatenation as the reference
        self._read_win(_lines(stem + ".win"))
        self._read_hr(_lines(stem + "_hr.dat"))
        self._read_centres(_lines(stem + "_centres.xyz"))

    # ---- prefix.win: the unit_cell_cart block, optional unit line (pythtb.py:3311-3340)
    def _read_win(self, ln):
        self.lat = np.zeros((3, 3), dtype=float)
        for i, line in enumerate(ln):
            sp = line.split()
            if len(sp) >= 2 and sp[0].lower() == "begin" and sp[1].lower() == "unit_cell_cart":
                unit = ln[i + 1].strip().lower()
                if unit == "bohr":
                    scale, first = _BOHR, i + 2
                elif unit in ("ang", "angstrom"):
                    scale, first = 1.0, i + 2
                else:
                    scale, first = 1.0, i + 1
                for j in range(3):
                    row = ln[first + j].split()
                    for c in range(3):
                        self.lat[j, c] = float(row[c]) * scale
                return
        raise Exception("Unable to find unit_cell_cart block in the .win file.")

    # ---- prefix_hr.dat: header, degeneracies, then "R1 R2 R3 i j re im" (pythtb.py:3342-3404)
    def _read_hr(self, ln):
        self.num_wan = int(ln[1])
        num_ws = int(ln[2])
        deg = []
        body = None
        for j in range(3, len(ln)):
            deg.extend(int(s) for s in ln[j].split())
            if len(deg) == num_ws:
                body = j + 1
                break
            if len(deg) > num_ws:
                raise Exception("Too many degeneracies for WS points!")
        if body is None:
            raise Exception("Too few degeneracies for WS points!")
        deg = np.array(deg, dtype=int)
        self.ham_r = {}               # ham_r[(R1,R2,R3)] = {"h": <i|H|j+R> matrix, "deg": WS degeneracy}
        seen = 0
        for line in ln[body:]:
            sp = line.split()
            if len(sp) < 7:
                if len(sp) == 0:
                    continue
                raise Exception("Malformed line in the _hr.dat file: " + line)
            key = (int(sp[0]), int(sp[1]), int(sp[2]))
            if key not in self.ham_r:
                self.ham_r[key] = {"h": np.zeros((self.num_wan, self.num_wan), dtype=complex), "deg": deg[seen]}
                seen += 1
            self.ham_r[key]["h"][int(sp[3]) - 1, int(sp[4]) - 1] = float(sp[5]) + 1.0j * float(sp[6])
        for R in self.ham_r:          # every R must come with exactly one -R
            if R != (0, 0, 0) and (-R[0], -R[1], -R[2]) not in self.ham_r:
                raise Exception("Did not find negative R for R = " + str(R) + "!")

    # ---- prefix_centres.xyz: two header lines, then "X x y z" per Wannier function (pythtb.py:3406-3424)
    def _read_centres(self, ln):
        cen = []
        for i in range(2, 2 + self.num_wan):
            sp = ln[i].split()
            if sp[0] != "X":
                raise Exception("Inconsistency in the centres file.")
            cen.append([float(sp[1]), float(sp[2]), float(sp[3])])
        self.xyz_cen = np.array(cen, dtype=float)
        to_red = np.linalg.inv(np.array([self.lat[0], self.lat[1], self.lat[2]]).T)       # pythtb.py:3925-3938
        self.red_cen = np.zeros_like(self.xyz_cen, dtype=float)
        for i in range(len(self.xyz_cen)):
            self.red_cen[i] = np.dot(to_red, self.xyz_cen[i])

    # ---- geometry helpers
    def _cart_of_R(self, R):
        return self.lat[0] * R[0] + self.lat[1] * R[1] + self.lat[2] * R[2]                # pythtb.py:3940-3947

    def _distances(self, R):
        """|-r_i + r_j + R| for all (i, j), with the reference's operation order."""
        vecR = np.zeros(3, dtype=float)
        vecR[:] = self._cart_of_R(R)
        n = self.num_wan
        out = np.zeros((n, n), dtype=float)
        for i in range(n):
            for j in range(n):
                v = -self.xyz_cen[i] + self.xyz_cen[j] + vecR
                out[i, j] = np.sqrt(np.dot(v, v))
        return out

    def model(self, zero_energy=0.0, min_hopping_norm=None, max_distance=None, ignorable_imaginary_part=None):
        """tb_model of the Wannier Hamiltonian (pythtb.py:3426-3560): energies shifted by
        `zero_energy`; hoppings weaker than `min_hopping_norm` or longer than `max_distance`
        dropped; imaginary parts below `ignorable_imaginary_part` zeroed.  One of each
        (R, -R) pair is kept (first non-zero component positive); for R = 0 the upper triangle."""
        tb = tb_model(3, 3, self.lat, self.red_cen)
        tb._assume_position_operator_diagonal = False
        home = self.ham_r[(0, 0, 0)]
        onsite = np.zeros(self.num_wan, dtype=float)
        for i in range(self.num_wan):
            val = home["h"][i, i] / float(home["deg"])
            onsite[i] = val.real
            if np.abs(val.imag) > 1.0E-9:
                raise Exception("Onsite terms should be real!")
        tb.set_onsite(onsite - zero_energy)
        n = self.num_wan
        hops = []
        for R in self.ham_r:
            first_nonzero = next((c for c in R if c != 0), 0)
            if first_nonzero < 0:
                continue                                   # its partner -R carries these terms
            home_cell = first_nonzero == 0
            amp = self.ham_r[R]["h"] / float(self.ham_r[R]["deg"])
            keep = np.triu(np.ones((n, n), dtype=bool), 1) if home_cell else np.ones((n, n), dtype=bool)
            if max_distance is not None:
                keep &= ~(self._distances(R) > max_distance)
            if min_hopping_norm is not None:
                keep &= ~(np.abs(amp) < min_hopping_norm)
            if ignorable_imaginary_part is not None:
                amp = np.where(np.abs(amp.imag) < ignorable_imaginary_part, amp.real + 0.0j, amp)
            for i, j in zip(*np.nonzero(keep)):            # row-major: the reference's (i, j) loop order
                hops.append([amp[i, j], int(i), int(j), np.array(list(R))])
        # every (i, j, R) above is distinct and no conjugate partner is present, so appending is
        # what set_hop(mode="set") would have done -- without its O(n_hop) scan per insertion
        tb._hoppings.extend(hops)
        tb._tbk_epoch += 1
        return tb

    def dist_hop(self):
        """(distances, amplitudes) of every term of H(R) except the on-site ones, in file order
        of R then (i, j) (pythtb.py:3562-3611)."""
        dist, ham = [], []
        n = self.num_wan
        for R in self.ham_r:
            amp = self.ham_r[R]["h"] / float(self.ham_r[R]["deg"])
            d = self._distances(R)
            keep = ~np.eye(n, dtype=bool) if R == (0, 0, 0) else np.ones((n, n), dtype=bool)
            ham.extend(amp[keep])
            dist.extend(d[keep])
        return (np.array(dist), np.array(ham))

    def shells(self, num_digits=2):
        """Sorted distinct inter-centre distances, rounded to `num_digits` (pythtb.py:3613-3651)."""
        found = set()
        for R in self.ham_r:
            for d in self._distances(R).flatten():
                found.add(round(d, num_digits))
        return np.sort(list(found))

    def w90_bands_consistency(self):
        """(kpts, ene) of Wannier90's own interpolated bands: kpts (nk,3) reduced coordinates from
        prefix_band.kpt, ene (num_wan, nk) from prefix_band.dat (pythtb.py:3653-3759)."""
        kpts = np.loadtxt(self.path + "/" + self.prefix + "_band.kpt", skiprows=1)
        kpts = kpts[:, :3]
        ene = np.loadtxt(self.path + "/" + self.prefix + "_band.dat")
        ene = ene[:, 1]
        ene = ene.reshape((self.num_wan, kpts.shape[0]))
        return (kpts, ene)
