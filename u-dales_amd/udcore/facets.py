"""Facet wall functions of the immersed boundary, host side (src/initfac.f90 readfacetfiles, src/modibm.f90:273-644
initibmwallfun): the facets (normal, roughness lengths, surface temperature) and, per velocity grid, the facet sections -- which
boundary cell a piece of wall acts on, over what area, from what distance, and where the velocity is taken for the log law
(the boundary point itself, or a reconstruction point further out along the facet normal when the log law is ill-defined
that close to the wall).  The tables go to the device once (udc_set_ibm_sections); `wallfunmom` (:1286-1433) runs there.

Input files, as the reference's pre-processing writes them next to the deck: facets.inp.NNN (type, normal), factypes.inp.NNN
(id, lGR, z0, z0h, ...), Tfacinit.inp.NNN (iwallmom = 2), facet_sections_{u,v,w}.txt (facet, area, boundary point, distance)
and the fluid_boundary_{u,v,w}.txt lists the section rows point into.
"""
from __future__ import annotations

import os

import numpy as np

EPS1 = 1.e-10                                                 # src/modglobal.f90:318


def _rows(path, nskip, n, ncol):
    a = np.loadtxt(path, skiprows=nskip, ndmin=2, max_rows=n)
    if a.shape[0] < n or a.shape[1] < ncol:
        raise ValueError(f"{path}: expected {n} rows of at least {ncol} columns")
    return a


def qsat(T):
    """src/initfac.f90:406-412 (Bolton 1980; Murphy & Koop 2005 at 1000 hPa)."""
    gres = 611.00 * np.exp(17.27 * (T - 273.15) / (T - 35.85))
    return 0.62198 * 0.01 * gres / (1000 - 0.01 * gres)


def read_facets(deck):
    """facnorm[nfcts, 3], z0[nfcts], z0h[nfcts], tsurf[nfcts] (facT(:, 1); zeros unless iwallmom = 2 / iwalltemp = 2 / iwallmoist = 2),
    lgr[nfcts] (vegetated: factypes' lGR column, src/initfac.f90:216) and, with iwallmoist = 2, the facets' humidity as initfac leaves
    it without an energy balance (:353-356): qsat at the facet temperature, fachurel of the vegetated ones from &ENERGYBALANCE wsoil, wfc."""
    base = os.path.dirname(os.path.abspath(deck.path))
    iexp = int(deck.get("RUN", "iexpnr"))
    nfcts = int(deck.get("WALLS", "nfcts"))
    if nfcts <= 0:
        raise ValueError("&WALLS nfcts: the facet wall functions (iwallmom > 1) need the facets of the pre-processing")
    fac = _rows(os.path.join(base, f"facets.inp.{iexp:03d}"), 1, nfcts, 4)
    with open(os.path.join(base, f"factypes.inp.{iexp:03d}")) as f:
        types = [ln.split() for ln in f.readlines()[3:] if ln.strip()]
    z0 = {int(float(t[0])): float(t[2]) for t in types}
    z0h = {int(float(t[0])): float(t[3]) for t in types}
    lgr = {int(float(t[0])): abs(float(t[1]) - 1.00) < 1.0e-5 for t in types}
    ftype = fac[:, 0].astype(int)
    out = {"norm": np.ascontiguousarray(fac[:, 1:4]), "z0": np.array([z0[t] for t in ftype]), "z0h": np.array([z0h[t] for t in ftype]),
           "tsurf": np.zeros(nfcts), "lgr": np.array([lgr[t] for t in ftype])}
    moist2 = int(deck.get("WALLS", "iwallmoist")) == 2
    if int(deck.get("WALLS", "iwallmom")) == 2 or int(deck.get("WALLS", "iwalltemp")) == 2 or moist2:      # src/initfac.f90:299
        out["tsurf"] = _rows(os.path.join(base, f"Tfacinit.inp.{iexp:03d}"), 1, nfcts, 1)[:, 0].copy()
    if moist2:
        wsoil, wfc = float(deck.get("ENERGYBALANCE", "wsoil")), float(deck.get("ENERGYBALANCE", "wfc"))
        out["qsat"] = qsat(out["tsurf"])
        out["hurel"] = np.where(out["lgr"], 0.5 * (1. - np.cos(3.14159 * wsoil / wfc)), 0.)
        out["resc"], out["ress"] = np.full(nfcts, 200.), np.full(nfcts, 50.)      # facf(:, 4), facf(:, 5): src/initfac.f90:134
    return out


def alignment(n):
    """src/modibm.f90:1683-1706."""
    for a, e in ((1, (1., 0., 0.)), (2, (0., 1., 0.)), (3, (0., 0., 1.)), (-1, (-1., 0., 0.)), (-2, (0., -1., 0.)), (-3, (0., 0., -1.))):
        if np.all(np.abs(np.asarray(n) - np.asarray(e)) < EPS1):
            return a
    return 0


def _plane_line(norm, v0, p0, p1):
    """src/modibm.f90:647-694 -> (point, check, dist)."""
    w, u = p0 - v0, p1 - p0
    d, n = float(np.dot(norm, u)), -float(np.dot(norm, w))
    if abs(d) < EPS1:
        return np.zeros(3), (2 if abs(n) < EPS1 else 0), 0.
    s = n / d
    pt = p0 + s * u
    return pt, (3 if (s < 0. or s > 1.) else 1), float(np.linalg.norm(pt - p0))


def _findloc(x, grid):
    """findloc(x >= grid, .true., back = .true.) for grid(1:n): the last 1-based index with grid <= x, 0 if none."""
    return int(np.searchsorted(grid, x, side="right"))


def wall_sections(deck, g, grid, bnd_pts, facets, lnorec=False):
    """The section table of one grid ('u', 'v', 'w'; 'c' for the heat wall function) in the file's order, skipped sections left out.
    bnd_pts: fluid_boundary_<grid>.txt rows (global 1-based i, j, k)."""
    base = os.path.dirname(os.path.abspath(deck.path))
    nsec = int(deck.get("WALLS", f"nfctsecs_{grid}"))
    rows = _rows(os.path.join(base, f"facet_sections_{grid}.txt"), 1, nsec, 4) if nsec else np.zeros((0, 4))
    nx, ny, nz, dx, dy = g.nx, g.ny, g.nz, g.dx, g.dy
    # (ib : itot + ih), (jb : jtot + jh), (kb : ktot + kh): index 0 of these arrays = reference index 1
    xh, xf = np.arange(nx + 1) * dx, (np.arange(nx + 1) + 0.5) * dx
    yh, yf = np.arange(ny + 1) * dy, (np.arange(ny + 1) + 0.5) * dy
    zf, zh = g.zf[1:nz + 2].copy(), g.zh[1:nz + 2].copy()
    dz1 = g.dzf[1]
    dir_align = {"u": 1, "v": 2, "w": 3, "c": 0}[grid]
    xg, yg, zg = {"u": (xh, yf, zf), "v": (xf, yh, zf), "w": (xf, yf, zh), "c": (xf, yf, zf)}[grid]
    xhat, yhat, zhat = np.eye(3)
    T = {k: [] for k in ("cell", "area", "dist", "fac", "comprec", "recpt", "recids")}
    for fac, area, bid, dst in rows:
        fac, bid = int(fac), int(bid)
        norm = facets["norm"][fac - 1]
        z0 = facets["z0"][fac - 1]
        if (dir_align != 0 and dir_align == alignment(norm)) or z0 < EPS1:            # :366-373
            continue
        i, j, k = (int(v) for v in bnd_pts[bid - 1])
        if not (1 <= i <= nx and 1 <= j <= ny):      # a boundary point no rank owns: its sections act nowhere (lfctsecsrank, :340-350)
            continue
        comprec, recpt, recids = True, np.zeros(3), np.zeros((4, 3), dtype=np.int32)
        if not (np.log(dst / z0) > 1. or lnorec):                    # :375-378: reconstruct
            comprec = False
            xc, yc, zc = xg[i - 1], yg[j - 1], zg[k - 1]
            p0 = np.array([xc, yc, zc])
            p1 = p0 + norm * np.sqrt(3.) * (dx * dy * dz1) ** (1. / 3.)
            planes = [(xhat, np.array([xc - dx / 2., yc, zc])), (xhat, np.array([xc + dx / 2., yc, zc])),
                      (yhat, np.array([xc, yc - dy / 2., zc])), (yhat, np.array([xc, yc + dy / 2., zc])),
                      (zhat, np.array([xc, yc, zc - dz1 / 2.])), (zhat, np.array([xc, yc, zc + dz1 / 2.]))]
            hits = [_plane_line(n, v, p0, p1) for n, v in planes]
            ok = [q for q, h in enumerate(hits) if h[1] == 1]
            if not ok:
                raise ValueError("initibmwallfun: no intersection found")
            pos = min(ok, key=lambda q: (hits[q][2], q))              # minloc: the first of equal minima
            recpt = hits[pos][0]
            grids = ((xh, yf, zf), (xf, yh, zf), (xf, yf, zh), (xf, yf, zf))      # u, v, w, c
            skip = False
            for q, (gx, gy, gz) in enumerate(grids):
                recids[q] = (_findloc(recpt[0], gx), _findloc(recpt[1], gy), _findloc(recpt[2], gz))
            for q in range(4):                                          # :447-482: the cell and its upper neighbour inside the arrays
                a, b, c = recids[q]
                if a < 1 or a + 1 > nx + 1 or b < 1 or b + 1 > ny + 1 or c < 1 or c + 1 > nz + 1:
                    skip = True
                    break
            if skip:
                continue
        T["cell"].append((i, j, k)); T["area"].append(area); T["dist"].append(dst); T["fac"].append(fac)
        T["comprec"].append(1 if comprec else 0); T["recpt"].append(recpt); T["recids"].append(recids)
    n = len(T["cell"])
    return {"n": n, "cell": np.array(T["cell"], dtype=np.int32).reshape(n, 3), "area": np.array(T["area"]), "dist": np.array(T["dist"]),
            "fac": np.array(T["fac"], dtype=np.int32), "comprec": np.array(T["comprec"], dtype=np.int32),
            "recpt": np.array(T["recpt"]).reshape(n, 3), "recids": np.array(T["recids"], dtype=np.int32).reshape(n, 4, 3)}


def c_mask(nx, ny, nz, solid_c, wrapx=False, wrapy=False):
    """mask_c of initibm (src/modibm.f90:188-191): 1 fluid / 0 solid on the c grid, [nz+2, ny+2, nx+2] with the ghost ring
    (index == the reference's k, j, i); the plane below the floor is 0; ghost cells are fluid unless the direction wraps."""
    m = np.ones((nz + 2, ny + 2, nx + 2))
    m[0] = 0.
    if len(solid_c):
        i, j, k = np.asarray(solid_c).T
        m[k, j, i] = 0.
    if wrapx:
        m[:, :, 0], m[:, :, -1] = m[:, :, -2], m[:, :, 1]
    if wrapy:
        m[:, 0], m[:, -1] = m[:, -2], m[:, 1]
    return m


def temperature_masks(grid, S, mask_c):
    """The two c-grid masks interp_temperature_<grid> (:1794-1830) reads per section: the boundary cell and its lower neighbour
    along the grid's direction."""
    di, dj, dk = {"u": (1, 0, 0), "v": (0, 1, 0), "w": (0, 0, 1)}[grid]
    out = np.ones((S["n"], 2))
    for s in range(S["n"]):
        i, j, k = (int(c) for c in S["cell"][s])
        out[s] = (mask_c[k, j, i], mask_c[k - dk, j - dj, i - di])
    return out


def prescribed_moisture_fluxes(deck, S, facets):
    """wallfunheat's latent part with iwallmoist = 1 (src/modibm.f90:1557-1568): +x bcqfxp, -x bcqfxm, +y bcqfyp, -y bcqfym, +z bcqfz on
    the vegetated facets; any other normal is undefined in the reference: refused.  -> flux per section (0 where not vegetated)."""
    bc = lambda n: float(deck.get("BC", n))      # noqa: E731
    table = {1: bc("bcqfxp"), -1: bc("bcqfxm"), 2: bc("bcqfyp"), -2: bc("bcqfym"), 3: bc("bcqfz")}
    out = np.zeros(S["n"])
    for s in range(S["n"]):
        f = int(S["fac"][s]) - 1
        if not facets["lgr"][f]:
            continue
        a = alignment(facets["norm"][f])
        if a not in table:
            raise ValueError("iwallmoist = 1 with non-zero wall moisture fluxes: the reference defines the flux for facets facing +-x, +-y, +z only")
        out[s] = table[a]
    return out


def prescribed_fluxes(deck, S, facets):
    """wallfunheat with iwalltemp = 1 (src/modibm.f90:1508-1524): one wall heat flux per facet direction -- +x bctfxp, -x bctfxm,
    +y bctfyp, -y bctfxm (the reference's own assignment), +z bctfz.  Any other normal keeps whatever `flux` held before in the
    reference (undefined): refused.  -> flux per section."""
    bc = lambda n: float(deck.get("BC", n))      # noqa: E731
    table = {1: bc("bctfxp"), -1: bc("bctfxm"), 2: bc("bctfyp"), -2: bc("bctfxm"), 3: bc("bctfz")}
    out = np.zeros(S["n"])
    for s in range(S["n"]):
        a = alignment(facets["norm"][int(S["fac"][s]) - 1])
        if a not in table:
            raise ValueError("iwalltemp = 1 with non-zero wall heat fluxes: the reference defines the flux for facets facing +-x, +-y, +z only")
        out[s] = table[a]
    return out
