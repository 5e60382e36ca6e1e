"""tb_model.display (pythtb.py:562-634): the plain-text report of a model.  Host-side convenience,
kept so that scripts written for the reference run unchanged; the text layout follows the
reference's report (tests/test_host_cpu.py compares it with captured output)."""
import numpy as np


def _num(x, width=7, digits=4):
    return str(round(x, digits)).rjust(width)


def _idx(i):
    return str(i).rjust(2)


def _cplx(z):
    z = complex(z)
    return "%s %s %s i" % (_num(z.real), "-" if z.imag < 0.0 else "+", _num(abs(z.imag)))


def _vector(values, fmt):
    return "[ " + " , ".join(fmt(v) for v in values) + " ]"


def _target(hop):
    """'j' or 'j + [ R ]' for a stored hopping [amp, i, j(, R)]."""
    text = _idx(hop[2])
    if len(hop) == 4:
        text += " + " + _vector(hop[3], _idx)
    return text


def report_lines(m):
    bar = "-" * 39
    out = [bar, "report of tight-binding model", bar,
           "k-space dimension           = %s" % m._dim_k,
           "r-space dimension           = %s" % m._dim_r,
           "number of spin components   = %s" % m._nspin,
           "periodic directions         = %s" % (m._per,),
           "number of orbitals          = %s" % m._norb,
           "number of electronic states = %s" % m._nsta,
           "lattice vectors:"]
    out += [" # %s  ===>  %s" % (_idx(i), _vector(v, _num)) for i, v in enumerate(m._lat)]
    out.append("positions of orbitals:")
    out += [" # %s  ===>  %s" % (_idx(i), _vector(v, _num)) for i, v in enumerate(m._orb)]
    out.append("site energies:")
    for i, e in enumerate(m._site_energies):
        shown = _num(e) if m._nspin == 1 else str(e).replace("\n", " ")
        out.append(" # %s  ===>   %s" % (_idx(i), shown))
    out.append("hoppings:")
    for hop in m._hoppings:
        shown = _cplx(hop[0]) if m._nspin == 1 else str(hop[0]).replace("\n", " ")
        out.append("< %s | H | %s >     ===>  %s" % (_idx(hop[1]), _target(hop), shown))
    out.append("hopping distances:")
    for hop in m._hoppings:
        start = np.dot(m._orb[hop[1]], m._lat)
        end = np.dot(m._orb[hop[2]], m._lat)
        if len(hop) == 4:
            end = end + np.dot(hop[3], m._lat)
        out.append("|  pos( %s )  - pos( %s ) |  =   %s" % (_idx(hop[1]), _target(hop), _num(np.linalg.norm(end - start))))
    out.append("")
    return out


def display(self):
    """Print a summary of the model: dimensions, lattice, orbitals, site energies, hoppings."""
    print("\n".join(report_lines(self)))
