"""Immersed boundary, host side: the reference's sparse input files -> udc_set_ibm_points / udc_ibm_commit.

The reference's preprocessing writes, per grid g in u, v, w, c, ``solid_g.txt`` and ``fluid_boundary_g.txt`` (one header line,
then rows of global 1-based ``i j k``) and the counts into &WALLS (nsolpts_g, nbndpts_g); initibm reads them with
read_sparse_ijk (src/modibm.f90:131-186, src/readinput.f90:31-130).  This slice has the corrections that need no facet
data, and the hand-over of the facet section tables (udcore/facets.py) for the wall functions.
"""
from __future__ import annotations

import os

import numpy as np

GRIDS = ("u", "v", "w", "c")


def read_points(path, n):
    """n rows of `i j k` after one header line (read_sparse_ijk)."""
    if n == 0:
        return np.zeros((0, 3), dtype=np.int32)
    a = np.loadtxt(path, skiprows=1, dtype=np.int64, ndmin=2)
    if a.shape[0] < n or a.shape[1] < 3:
        raise ValueError(f"{path}: expected {n} rows of i j k")
    return np.ascontiguousarray(a[:n, :3], dtype=np.int32)


def read_ibm(deck):
    """{grid: (solid[n,3], boundary[n,3])} from the deck's directory, counts from &WALLS."""
    base = os.path.dirname(os.path.abspath(deck.path))
    out = {}
    need_c = int(deck.get("SCALARS", "nsv")) > 0 or bool(deck.get("PHYSICS", "ltempeq")) or bool(deck.get("PHYSICS", "lmoist"))
    iwallmom = int(deck.get("WALLS", "iwallmom"))
    for g in GRIDS:
        if g == "c" and not need_c:
            continue
        ns, nb = int(deck.get("WALLS", f"nsolpts_{g}")), int(deck.get("WALLS", f"nbndpts_{g}"))
        if g != "c" and iwallmom == 1:
            # without wall functions initibm never reads the fluid-boundary points of the velocity grids (src/modibm.f90:166-179:
            # initibmwallfun for u, v, w sits under `iwallmom > 1`), so ibmwallfun's diffu/v/w_corr loop over no points: the
            # subgrid fluxes through solid faces stay in the momentum tendencies.  The c grid's list is read whatever iwallmom (:180-186).
            nb = 0
        out[g] = (read_points(os.path.join(base, f"solid_{g}.txt"), ns), read_points(os.path.join(base, f"fluid_boundary_{g}.txt"), nb))
    return out


def apply_ibm(core, deck):
    """Register the deck's immersed boundary with the device core (no-op unless &RUN libm)."""
    if not deck.get("RUN", "libm"):
        return None
    iwallmom = int(deck.get("WALLS", "iwallmom"))
    if iwallmom not in (1, 2, 3):
        raise ValueError("&WALLS iwallmom must be 1 (no wall functions), 2 (stability functions) or 3 (neutral)")
    if iwallmom == 2 and not deck.get("PHYSICS", "ltempeq"):
        raise ValueError("libm with iwallmom = 2: the stability functions read the air temperature (ltempeq); use iwallmom = 3 for neutral walls")
    # wallfunheat (src/modibm.f90:1436): the sensible part from the facet temperatures (iwalltemp = 2) or prescribed (1; all-zero fluxes
    # add exactly nothing: adiabatic walls, no tables needed); the latent part on the vegetated facets prescribed (iwallmoist = 1; zero:
    # impermeable walls) or from the facets' humidity (2: needs the heat transfer coefficient of iwalltemp = 2).
    bc = lambda n: float(deck.get("BC", n))      # noqa: E731
    iwalltemp = int(deck.get("WALLS", "iwalltemp"))
    if deck.get("PHYSICS", "ltempeq") and iwalltemp not in (1, 2):
        raise ValueError("libm with ltempeq: iwalltemp must be 1 (prescribed wall heat fluxes bctf*) or 2 (from the facet temperatures)")
    fluxes = iwalltemp == 1 and any(bc(n) != 0. for n in ("bctfxm", "bctfxp", "bctfym", "bctfyp", "bctfz"))
    lmoist = bool(deck.get("PHYSICS", "lmoist"))
    iwallmoist = int(deck.get("WALLS", "iwallmoist")) if lmoist else 0
    if lmoist and iwallmoist not in (1, 2):
        raise ValueError("libm with lmoist: iwallmoist must be 1 (prescribed wall moisture fluxes bcqf*) or 2 (from the facets' humidity)")
    qfluxes = iwallmoist == 1 and any(bc(n) != 0. for n in ("bcqfxm", "bcqfxp", "bcqfym", "bcqfyp", "bcqfz"))
    if iwallmoist == 2 and not (deck.get("PHYSICS", "ltempeq") and iwalltemp == 2):
        raise ValueError("libm with iwallmoist = 2: the aerodynamic resistance comes from the heat transfer coefficient of iwalltemp = 2 "
                         "(undefined in the reference otherwise, src/modibm.f90:1571-1574)")
    if qfluxes and not deck.get("PHYSICS", "ltempeq"):
        raise ValueError("libm with wall moisture fluxes: wallfunheat runs with the temperature equation (ltempeq)")
    core.set_ibm_conservative(bool(deck.get("PHYSICS", "lconservativeibm")))
    lists = read_ibm(deck)
    # the masks' ghost cells as the reference run of this deck has them: wrapped only in a direction it splits over ranks
    core.set_ibm_mask_wrap(int(deck.get("RUN", "nprocx")) > 1, int(deck.get("RUN", "nprocy")) > 1)
    for q, g in enumerate(GRIDS):
        if g in lists:
            core.set_ibm_points(q, *lists[g])
    core.ibm_commit()
    latent = iwallmoist == 2 or qfluxes
    heat = bool(deck.get("PHYSICS", "ltempeq")) and (iwalltemp == 2 or fluxes or latent)
    if iwallmom > 1 or heat:      # facet wall functions (wallfunmom, wallfunheat): facets and section tables to the device
        from .facets import c_mask, read_facets, temperature_masks, wall_sections
        g = core.g
        facets = read_facets(deck)
        wrapx, wrapy = int(deck.get("RUN", "nprocx")) > 1, int(deck.get("RUN", "nprocy")) > 1
        jtot = int(deck.get("DOMAIN", "jtot"))
        gg = g if g.ny == jtot else type(g).from_deck(deck)
        mask = c_mask(gg.nx, gg.ny, gg.nz, lists["c"][0] if "c" in lists else [], wrapx, wrapy)
        lnorec = bool(deck.get("WALLS", "lnorec"))      # &WALLS lnorec: no reconstruction points (src/modibm.f90:380, 1345, 1481)
        core.set_ibm_wallfun(iwallmom, float(deck.get("WALLS", "prandtlturb")), gg.zf[1:gg.nz + 2], gg.zh[1:gg.nz + 2])
        if iwallmom > 1:
            for q, gr in enumerate("uvw"):
                S = wall_sections(deck, gg, gr, lists[gr][1], facets, lnorec)
                core.set_ibm_sections(q, S, facets, temperature_masks(gr, S, mask))
        if heat:
            S = wall_sections(deck, gg, "c", lists["c"][1], facets, lnorec)
            nf = np.asarray(S["fac"]) - 1
            if latent:      # per section: vegetated?, then the prescribed flux or the facet's humidity and resistances
                from .facets import prescribed_moisture_fluxes
                if iwallmoist == 1:
                    wm = (facets["lgr"][nf], prescribed_moisture_fluxes(deck, S, facets), None, None, None)
                else:
                    wm = (facets["lgr"][nf], facets["qsat"][nf], facets["hurel"][nf], facets["resc"][nf], facets["ress"][nf])
            if iwalltemp == 1:      # prescribed fluxes ride in the slot of the facet temperature
                from .facets import prescribed_fluxes
                facets = {"norm": facets["norm"][nf], "z0": facets["z0"][nf], "z0h": facets["z0h"][nf],
                          "tsurf": prescribed_fluxes(deck, S, facets)}      # one entry per section
                S = dict(S, fac=np.arange(1, S["n"] + 1, dtype=np.int32))
            core.set_ibm_sections(3, S, facets, np.ones((S["n"], 2)))
            core.set_ibm_wallheat(iwalltemp)
            if latent:
                core.set_ibm_wallmoist(iwallmoist, *wm)
    return lists
