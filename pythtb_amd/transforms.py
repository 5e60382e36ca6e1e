"""Model -> model transforms of PythTB's tb_model: cut_piece, reduce_dim,
change_nonperiodic_vector, make_supercell, remove_orb (pythtb.py:1105-1789).

Pure host code: each runs once, in O(n_hop), and hands the hot path nothing but another
hopping table.  The functions are written against the public setters so that the stored
tables (orbital order, hopping order, merged amplitudes) come out exactly as the reference
builds them; tests/test_transforms.py compares them table by table with fixtures captured
from the reference.
"""
import copy

import numpy as np


def _is_int(a):
    return np.issubdtype(type(a), np.integer)


def cut_piece(self, num, fin_dir, glue_edgs=False):
    """Finite piece of `num` cells along periodic lattice vector `fin_dir`: a (dim_k-1)-periodic
    model whose orbital i of cell n is numbered i + norb*n (pythtb.py:1105-1231)."""
    if self._dim_k == 0:
        raise Exception("\n\nModel is already finite")
    if not _is_int(num):
        raise Exception("\n\nArgument num not an integer")
    if num < 1:
        raise Exception("\n\nArgument num must be positive!")
    if num == 1 and glue_edgs == True:  # noqa: E712
        raise Exception("\n\nCan't have num==1 and glueing of the edges!")
    no = self._norb
    orbs = np.tile(self._orb, (num, 1))
    orbs[:, fin_dir] += np.repeat(np.arange(num, dtype=float), no)
    onsite = np.array([self._site_energies[j] for _ in range(num) for j in range(no)])
    per = copy.deepcopy(self._per)
    if per.count(fin_dir) != 1:
        raise Exception("\n\nCan not make model finite along this direction!")
    per.remove(fin_dir)
    out = type(self)(self._dim_k - 1, self._dim_r, copy.deepcopy(self._lat), orbs if no else [], per, self._nspin)
    out._assume_position_operator_diagonal = self._assume_position_operator_diagonal
    out.set_onsite(onsite, mode="reset")
    total = no * num
    for cell in range(num):
        for hop in self._hoppings:
            R = copy.deepcopy(hop[3])
            jump = R[fin_dir]                      # cells travelled along the cut direction
            if out._dim_k != 0:
                R[fin_dir] = 0
            hi = hop[1] + cell * no
            hj = hop[2] + (cell + jump) * no
            if glue_edgs == False:  # noqa: E712
                if hj < 0 or hj >= total:
                    continue                       # the bond leaves the piece
            else:
                hj = int(hj) % int(total)
            if out._dim_k == 0:
                out.set_hop(hop[0], hi, hj, mode="add", allow_conjugate_pair=True)
            else:
                out.set_hop(hop[0], hi, hj, R, mode="add", allow_conjugate_pair=True)
    return out


def reduce_dim(self, remove_k, value_k):
    """Fix one k component at `value_k`: the Bloch phase of that component is folded into the
    amplitudes, pure-phase self-hoppings become on-site terms (pythtb.py:1233-1311)."""
    if self._dim_k == 0:
        raise Exception("\n\nCan not reduce dimensionality even further!")
    out = copy.deepcopy(self)
    out._per.remove(remove_k)
    out._dim_k = len(out._per)
    if out._dim_k != self._dim_k - 1:
        raise Exception("\n\nSpecified wrong dimension to reduce!")
    out._hoppings = []
    out._tbk_epoch += 1
    for hop in self._hoppings:
        amp = complex(hop[0]) if self._nspin == 1 else np.array(hop[0], dtype=complex)
        i, j = hop[1], hop[2]
        R = np.array(hop[3], dtype=int)
        rv = (-out._orb[i, :] + out._orb[j, :] + np.array(R, dtype=float))[remove_k]
        phase = np.exp((2.0j) * np.pi * (value_k * rv))
        if i == j and np.all(np.array(R[out._per], dtype=int) == 0):
            if R[remove_k] == 0:
                out.set_onsite(amp * phase, i, mode="add")
            elif self._nspin == 1:
                out.set_onsite(amp * phase + (amp * phase).conj(), i, mode="add")
            else:
                out.set_onsite(amp * phase + (amp.T * phase).conj(), i, mode="add")
        else:
            R[remove_k] = 0
            out.set_hop(amp * phase, i, j, R, mode="add", allow_conjugate_pair=True)
    return out


def _shift_to_home(self, to_home_suppress_warning=False):
    """Bring orbitals back to the home cell along periodic directions (pythtb.py:1639-1715).

    Reproduced as the reference (v1.8.0) actually executes it: the shift of the orbital and of
    the lattice vectors of its hoppings sits after the orbital loop and inside the
    `to_home_suppress_warning == False` branch (:1682-1715), so only the LAST orbital is moved,
    and only when the warning is not suppressed.  Spectra are unaffected either way; orbital
    positions (hence Berry-phase conventions) are, so a drop-in keeps the same behaviour."""
    flagged = [[] for _ in range(self._dim_r)]
    disp = np.zeros(self._dim_r, dtype=int)
    last = -1
    for i in range(self._norb):
        disp = np.zeros(self._dim_r, dtype=int)
        for k in range(self._dim_r):
            shift = np.floor(self._orb[i, k] + 1.0E-6).astype(int)
            if k in self._per:
                disp[k] = shift
            elif shift != 0:
                flagged[k] = flagged[k] + [i]
        last = i
    if to_home_suppress_warning == False:  # noqa: E712
        lines = ""
        for k in range(self._dim_r):
            if flagged[k] != []:
                lines += "  * Direction %1d : Orbitals " % k + ', '.join(str(e) for e in flagged[k]) + "\n"
        if lines != "":
            bar = '  ' + 69 * '-' + '\n'
            print(bar + "  WARNING from '_shift_to_home' (called by 'change_nonperiodic_vector'\n"
                  "  or 'make_supercell'): Orbitals are not \"shifted to home\" along\n"
                  "  non-periodic directions (PythTB 1.7.3 and newer do not shift there).\n"
                  "  The following orbitals would have been assigned different coordinates\n"
                  "  in PythTB 1.7.2 and older:\n  *\n" + lines +
                  "  *\n  To prevent printing this warning, pass 'to_home_suppress_warning=True'.\n" + bar)
        if last >= 0:
            self._orb[last] -= disp
            if self._dim_k != 0:
                for hop in self._hoppings:
                    if hop[1] == last:
                        hop[3] -= disp
                    if hop[2] == last:
                        hop[3] += disp
            self._tbk_epoch += 1


def change_nonperiodic_vector(self, np_dir, new_latt_vec=None, to_home=True, to_home_suppress_warning=False):
    """Replace one non-periodic lattice vector (by default with its component perpendicular to the
    periodic ones), keeping Cartesian orbital positions (pythtb.py:1313-1438)."""
    if self._per.count(np_dir) == 1:
        print("\nnp_dir =", np_dir)
        raise Exception("Selected direction is not nonperiodic")
    if new_latt_vec is None:
        periodic = np.zeros_like(self._lat)
        for d in self._per:
            periodic[d] = self._lat[d]
        coeffs = np.linalg.lstsq(periodic.T, self._lat[np_dir], rcond=None)[0]
        new_vec = self._lat[np_dir] - np.dot(self._lat.T, coeffs)
    else:
        new_vec = np.array(new_latt_vec)
        if new_vec.shape != (self._dim_r,):
            raise Exception("\n\nNonperiodic vector has wrong length")
    lat = copy.deepcopy(self._lat)
    lat[np_dir] = new_vec
    orb = [np.linalg.solve(lat.T, np.dot(self._lat.T, o)) for o in self._orb]
    out = copy.deepcopy(self)
    out._lat = np.array(lat, dtype=float)
    out._orb = np.array(orb, dtype=float)
    out._tbk_epoch += 1
    if new_latt_vec is None:
        for d in out._per:
            if np.abs(np.dot(out._lat[d], out._lat[np_dir])) > 1.0E-6:
                raise Exception("\n\nThis shouldn't happen.  New nonperiodic vector \n"
                                "is not perpendicular to periodic vectors!?")
    for i in range(self._orb.shape[0]):
        if np.max(np.abs(np.dot(self._lat.T, self._orb[i]) - np.dot(out._lat.T, out._orb[i]))) > 1.0E-6:
            raise Exception("\n\nThis shouldn't happen. New choice of nonperiodic vector\n"
                            "somehow changed Cartesian coordinates of orbitals.")
    if np.abs(np.linalg.det(out._lat)) < 1.0E-6:
        raise Exception("\n\nLattice with new choice of nonperiodic vector has zero volume?!")
    if to_home == True:  # noqa: E712
        out._shift_to_home(to_home_suppress_warning)
    return out


def make_supercell(self, sc_red_lat, return_sc_vectors=False, to_home=True, to_home_suppress_warning=False):
    """Super-cell spanned by the integer combinations `sc_red_lat` of the lattice vectors
    (pythtb.py:1440-1637)."""
    if self._dim_r == 0:
        raise Exception("\n\nMust have at least one periodic direction to make a super-cell")
    S = np.array(sc_red_lat)
    if S.shape != (self._dim_r, self._dim_r):
        raise Exception("\n\nDimension of sc_red_lat array must be dim_r*dim_r")
    if S.dtype != int:
        raise Exception("\n\nsc_red_lat array elements must be integers")
    for i in range(self._dim_r):
        for j in range(self._dim_r):
            if i == j and i not in self._per and S[i, j] != 1:
                raise Exception("\n\nDiagonal elements of sc_red_lat for non-periodic directions must equal 1.")
            if i != j and (i not in self._per or j not in self._per) and S[i, j] != 0:
                raise Exception("\n\nOff-diagonal elements of sc_red_lat for non-periodic directions must equal 0.")
    if np.abs(np.linalg.det(S)) < 1.0E-6:
        raise Exception("\n\nSuper-cell lattice vectors length/area/volume too close to zero, or zero.")
    if np.linalg.det(S) < 0.0:
        raise Exception("\n\nSuper-cell lattice vectors need to form right handed system.")
    if self._dim_r > 4:
        raise Exception("\n\nWrong dimensionality of dim_r!")
    St = np.array(S.T, dtype=float)

    def to_red_sc(v):
        return np.linalg.solve(St, np.array(v, dtype=float))

    # original-lattice vectors inside the super-cell, in the reference's enumeration order
    max_R = np.max(np.abs(S)) * self._dim_r
    eps = np.sqrt(2.0) * 1.0E-8
    sc_vec = []
    for idx in np.ndindex(*([2 * max_R + 1] * self._dim_r)):
        vec = np.array(idx) - max_R
        red = to_red_sc(vec)
        if np.all(red > -eps) and np.all(red <= 1.0 - eps):
            sc_vec.append(vec)
    if int(round(np.abs(np.linalg.det(S)))) != len(sc_vec):
        raise Exception("\n\nSuper-cell generation failed! Wrong number of super-cell vectors found.")
    sc_orb = [to_red_sc(o + v) for v in sc_vec for o in self._orb]
    out = type(self)(self._dim_k, self._dim_r, np.dot(S, self._lat), sc_orb, per=self._per, nspin=self._nspin)
    out._assume_position_operator_diagonal = self._assume_position_operator_diagonal
    for c in range(len(sc_vec)):
        for j in range(self._norb):
            out.set_onsite(self._site_energies[j], c * self._norb + j)
    lookup = {tuple(int(x) for x in v): p for p, v in enumerate(sc_vec)}
    for c, v in enumerate(sc_vec):
        for hop in self._hoppings:
            R = copy.deepcopy(hop[3])
            sc_part = np.array(np.floor(to_red_sc(R + v)), dtype=int)      # round down!
            inside = R + v - np.dot(sc_part, S)
            p = lookup.get(tuple(int(x) for x in inside))
            if p is None:
                raise Exception("\n\nDid not find super cell vector!")
            out.set_hop(hop[0], hop[1] + c * self._norb, hop[2] + p * self._norb, sc_part,
                        mode="add", allow_conjugate_pair=True)
    if to_home == True:  # noqa: E712
        out._shift_to_home(to_home_suppress_warning)
    if return_sc_vectors == False:  # noqa: E712
        return out
    return (out, sc_vec)


def remove_orb(self, to_remove):
    """Model without the listed orbitals; higher indices move down (pythtb.py:1718-1789)."""
    drop = [to_remove] if _is_int(to_remove) else copy.deepcopy(to_remove)
    for o in drop:
        if o < 0 or o > self._norb - 1 or (not _is_int(o)):
            raise Exception("\n\nSpecified wrong orbitals to remove!")
    if len(set(drop)) != len(drop):
        raise Exception("\n\nSpecified duplicate orbitals to remove!")
    out = copy.deepcopy(self)
    out._norb -= len(drop)
    out._nsta -= len(drop) * self._nspin
    for o in sorted(drop, reverse=True):
        out._orb = np.delete(out._orb, o, 0)
        out._site_energies = np.delete(out._site_energies, o, 0)
        out._site_energies_specified = np.delete(out._site_energies_specified, o)
        kept = []
        for hop in out._hoppings:
            if hop[1] == o or hop[2] == o:
                continue
            if hop[1] > o:
                hop[1] -= 1
            if hop[2] > o:
                hop[2] -= 1
            kept.append(hop)
        out._hoppings = kept
    out._tbk_epoch += 1
    return out


def install(cls):
    """Attach the transforms to the tb_model class."""
    for fn in (cut_piece, reduce_dim, _shift_to_home, change_nonperiodic_vector, make_supercell, remove_orb):
        setattr(cls, fn.__name__, fn)
