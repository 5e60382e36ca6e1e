"""tb_model.visualize (pythtb.py:636-860): a 2-D sketch of the model -- cell vectors, orbitals,
hopping bonds and optionally one eigenstate.  Host-side convenience (matplotlib is imported on
use), kept so that scripts written for the reference run unchanged."""
import colorsys

import numpy as np

_PLAIN = dict(cell="b", orbital="r", image=[0.85, 0.65, 0.65], bond="g")
_GREY = dict(cell=[0.4, 0.4, 0.4], orbital=[0.0, 0.0, 0.0], image=[0.6, 0.6, 0.6], bond=[0.0, 0.0, 0.0])


def _phase_colour(phase, scheme):
    if scheme == "black":
        return "k"
    if scheme == "red-blue":                       # 0 -> red, +-pi -> blue
        t = abs(phase) / np.pi
        return [1.0 - t, 0.0, t]
    hue = (phase % (2.0 * np.pi)) / (2.0 * np.pi)   # "wheel": red, yellow, green, cyan, blue, magenta in steps of pi/3
    return list(colorsys.hsv_to_rgb(hue, 1.0, 1.0))


def visualize(self, dir_first, dir_second=None, eig_dr=None, draw_hoppings=True, ph_color="black"):
    """Draw the model projected on Cartesian axes (dir_first, dir_second).  With `eig_dr` (one complex
    amplitude per orbital) the state is drawn as discs of area ~ |amplitude|^2 coloured by phase
    according to `ph_color` ("black", "red-blue" or "wheel").  Returns (fig, ax)."""
    if eig_dr is not None and np.shape(eig_dr) != (self._norb,):
        raise Exception("\n\nWrong format of eig_dr! Must be array of size norb.")
    if ph_color not in ("black", "red-blue", "wheel"):
        raise Exception("\n\nWrong value of ph_color parameter!")
    if dir_second is None and self._dim_r > 1:
        raise Exception("\n\nNeed to specify index of second coordinate for projection!")
    import matplotlib.pyplot as plt

    def flat(red):                                 # reduced coordinates -> point(s) in the drawing plane
        cart = np.atleast_2d(np.dot(red, self._lat))
        second = np.zeros(len(cart)) if dir_second is None else cart[:, dir_second]
        return np.column_stack([cart[:, dir_first], second])

    side = plt.rcParams["figure.figsize"][0]
    fig = plt.figure(figsize=[side, side])
    ax = fig.add_subplot(111, aspect="equal")
    col = _PLAIN if (eig_dr is None or ph_color == "black") else _GREY
    dot = dict(mec="w", mew=0.0)
    ax.plot([0.0], [0.0], "o", c=col["cell"], zorder=7, ms=4.5, **dot)
    for d in self._per:
        tip = flat(np.eye(self._dim_r)[d])[0]
        ax.plot([0.0, tip[0]], [0.0, tip[1]], "-", c=col["cell"], lw=1.5, zorder=7)
    home = flat(self._orb)
    for x, y in home:
        ax.plot([x], [y], "o", c=col["orbital"], zorder=10, ms=4.0, **dot)
    if draw_hoppings == True:  # noqa: E712
        for hop in self._hoppings:
            shift = np.zeros(self._dim_r)
            if self._dim_k != 0:
                shift[self._per] = np.array(hop[3])[self._per]
            # the bond and its mirror image: i -> j+R drawn from the home cell, and i-R -> j
            for a_red, b_red in ((self._orb[hop[1]], self._orb[hop[2]] + shift), (self._orb[hop[1]] - shift, self._orb[hop[2]])):
                a, b = flat(a_red)[0], flat(b_red)[0]
                span = b - a
                length = np.sqrt(np.dot(span, span))
                bow = np.array([span[1], -span[0]]) / length            # unit normal: bonds are bowed by 5 %
                mid = 0.5 * (a + b) + 0.05 * length * bow
                ax.plot([a[0], mid[0], b[0]], [a[1], mid[1], b[1]], "-", c=col["bond"], lw=0.75, zorder=8)
                for x, y in (a, b):
                    ax.plot([x], [y], "o", c=col["image"], zorder=9, ms=4.0, **dot)
    if eig_dr is not None:
        amp = np.asarray(eig_dr)
        weight = (amp * amp.conjugate()).real
        for (x, y), wgt, z in zip(home, weight, amp):
            ax.plot([x], [y], "o", c=_phase_colour(np.angle(z), ph_color), ms=2.0 * wgt * float(self._norb),
                    zorder=11, alpha=0.8, **dot)
    (x0, x1), (y0, y1) = ax.set_xlim(), ax.set_ylim()
    half = max(x1 - x0, y1 - y0) * 0.55                                  # square window with a 5 % margin
    ax.set_xlim(0.5 * (x0 + x1) - half, 0.5 * (x0 + x1) + half)
    ax.set_ylim(0.5 * (y0 + y1) - half, 0.5 * (y0 + y1) + half)
    return (fig, ax)
