"""Scalar point / line sources (`scalsource`, src/modscalsource.f90:379-483; files read by `createscals`, :283-375).

The sources do not depend on time or on the flow, so the host evaluates them once -- the reference's expressions, on this
rank's slab -- and hands each scalar's source field to the device as the smallest box that holds it
(`udc_set_scalar_source`); the device adds it to svp every substep where the reference calls `scalsource`
(src/program.f90:181).
"""
import math
import os

import numpy as np

from .namoptions import _read_table

PI = 3.141592653589793116          # src/modglobal.f90:270


def read_sources(deck):
    """{'p': [per scalar: rows (xS yS zS SS sigS)], 'l': [per scalar: rows (xSb ySb zSb xSe ySe zSe SS sigS)]}."""
    base = os.path.dirname(os.path.abspath(deck.path))
    exp = deck.get("RUN", "iexpnr")
    nsv = int(deck.get("SCALARS", "nsv"))
    out = {"p": [[] for _ in range(nsv)], "l": [[] for _ in range(nsv)]}
    for kind, flag, cnt, ncol in (("p", "lscasrc", "nscasrc", 5), ("l", "lscasrcl", "nscasrcl", 8)):
        n = int(deck.get("SCALARS", cnt))
        if not deck.get("SCALARS", flag) or nsv == 0 or n <= 0:
            continue
        for m in range(nsv):
            out[kind][m] = _read_table(os.path.join(base, f"scalarsource{kind}.inp.{m + 1}.{exp:03d}"), ncol, n)
    return out


def source_field(g, points, lines, j0=0, nyl=None):
    """Source tendency [nz, nyl, nx] (interior cells of the slab starting at global row j0) of one scalar."""
    nyl = g.ny if nyl is None else nyl
    nx, nz = g.nx, g.nz
    dxi, dyi = 1. / g.dx, 1. / g.dy
    px = ((np.arange(1, nx + 1) - 0.5) * g.dx)[None, None, :]
    py = ((np.arange(1, nyl + 1) + j0 - 0.5) * g.dy)[None, :, None]
    pz = np.asarray(g.zf[1:nz + 1])[:, None, None]
    dzfi = (1. / np.asarray(g.dzf[1:nz + 1]))[:, None, None]
    s = np.zeros((nz, nyl, nx))
    for xS, yS, zS, SS, sigS in points:
        ra2 = (px - xS) ** 2 + (py - yS) ** 2 + (pz - zS) ** 2
        s = s + np.where(ra2 <= 9 * sigS ** 2, dxi * dyi * dzfi * SS * np.exp(-ra2 / (2 * sigS ** 2)), 0.)
    erf = np.vectorize(math.erf)
    for xSb, ySb, zSb, xSe, ySe, zSe, SS, sigS in lines:
        lsx, lsy, lsz = xSe - xSb, ySe - ySb, zSe - zSb
        dot = ((px - xSb) * lsx + (py - ySb) * lsy + (pz - zSb) * lsz) / (lsx * lsx + lsy * lsy + lsz * lsz)
        # nearest point of the segment: its begin, its end, or the projection (:450-456)
        rb = (px - xSb) ** 2 + (py - ySb) ** 2 + (pz - zSb) ** 2
        re = (px - xSe) ** 2 + (py - ySe) ** 2 + (pz - zSe) ** 2
        rp = (px - (xSb + dot * lsx)) ** 2 + (py - (ySb + dot * lsy)) ** 2 + (pz - (zSb + dot * lsz)) ** 2
        ra2 = np.where(dot < 0.0, rb, np.where(dot > 1.0, re, rp))
        inside = ra2 <= 9 * sigS ** 2
        arg = np.where(inside, (9 * sigS ** 2 - ra2) / (2 * sigS ** 2), 0.)
        s = s + np.where(inside, dxi * dyi * dzfi * math.sqrt(2.0 * PI) * SS * sigS * np.exp(-ra2 / (2 * sigS ** 2))
                         * erf(np.sqrt(arg)), 0.)
    return s


def apply_sources(core, deck, j0=0):
    """Evaluate the deck's sources on this rank's slab and register them with the device.  Returns the number of scalars
    that have one."""
    src = read_sources(deck)
    done = 0
    for n in range(core.nsv):
        if not src["p"][n] and not src["l"][n]:
            continue
        s = source_field(core.g, src["p"][n], src["l"][n], j0=j0, nyl=core.nyl)
        core.set_scalar_source(n, s)
        done += 1
    return done
