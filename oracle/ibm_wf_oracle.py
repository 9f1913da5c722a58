"""TEST INFRASTRUCTURE -- numpy restatement of the reference's momentum wall function on facet sections, NOT the product.

wallfunmom (src/modibm.f90:1286-1433) with its helpers: interp_velocity_u/v/w (:1738-1777), interp_temperature_u/v/w
(:1794-1830), trilinear_interp_var (:1610-1680), local_coords (:1833-1853), mom_transfer_coef_stability (:1856-1904, Uno et
al. 1995) and mom_transfer_coef_neutral (:1907-1917).  Pinned on the reference's own compiled routines: oracle/_ref/udales_ref
holds them (oracle/extract_modibm.sh, oracle/extract_initfac.sh) and the fixtures k_ibm_wf3_*, k_ibm_wf2_*, run_ibm_wf2_* carry
the tendencies before and after its `ibmwallfun`.

Fields are m-arrays [nz+2, ny+2, nx+2] (index == the reference's k, j, i with one ghost ring).  Sections: the table of
udcore.facets.wall_sections (file order, skipped sections left out), `facets` of udcore.facets.read_facets.
"""
import numpy as np

EPS1, GRAV, FKAR = 1.e-10, 9.81, 0.41


def _trilinear(var, cell, xg, yg, zg, p):
    i, j, k = (int(c) for c in cell)
    c = [var[k, j, i], var[k, j, i + 1], var[k, j + 1, i], var[k, j + 1, i + 1],
         var[k + 1, j, i], var[k + 1, j, i + 1], var[k + 1, j + 1, i], var[k + 1, j + 1, i + 1]]
    x0, y0, z0, x1, y1, z1 = xg[i - 1], yg[j - 1], zg[k - 1], xg[i], yg[j], zg[k]
    xd, yd, zd = (p[0] - x0) / (x1 - x0), (p[1] - y0) / (y1 - y0), (p[2] - z0) / (z1 - z0)
    return (c[0] * (1 - xd) * (1 - yd) * (1 - zd) + c[1] * xd * (1 - yd) * (1 - zd) + c[2] * (1 - xd) * yd * (1 - zd) + c[3] * xd * yd * (1 - zd)
            + c[4] * (1 - xd) * (1 - yd) * zd + c[5] * xd * (1 - yd) * zd + c[6] * (1 - xd) * yd * zd + c[7] * xd * yd * zd)


def _ctm_stability(utan, dist, z0, z0h, tair, tsurf, prt):
    b1, b2, dm, dh = 9.4, 4.7, 7.4, 5.3
    dT = tair - tsurf
    ri0 = GRAV * dist * dT / (tsurf * utan ** 2)
    logdz, logzh, sqdz, fkar2 = np.log(dist / z0), np.log(z0 / z0h), np.sqrt(dist / z0), FKAR ** 2
    if ri0 > 0.:
        fm = 1. / (1. + b2 * ri0) ** 2
        fh = fm
    else:
        cm = (dm * fkar2) / (logdz ** 2) * b1 * sqdz
        ch = (dh * fkar2) / (logdz ** 2) * b1 * sqdz
        fm = 1. - (b1 * ri0) / (1. + cm * np.sqrt(abs(ri0)))
        fh = 1. - (b1 * ri0) / (1. + ch * np.sqrt(abs(ri0)))
    M = prt * logdz * np.sqrt(fm) / fh
    ri1 = ri0 - ri0 * prt * logzh / (prt * logzh + M)
    if ri1 > 0.:
        fm = 1. / (1. + b2 * ri1) ** 2
    else:
        cm = (dm * fkar2) / (logdz ** 2) * b1 * sqdz
        fm = 1. - (b1 * ri1) / (1. + cm * np.sqrt(abs(ri1)))
    return fkar2 / (logdz ** 2) * fm


def wallfunmom(grid, g, S, facets, iwallmom, u0, v0, w0, rhs, thl0=None, mask_c=None, prt=0.71, lnorec=False):
    """rhs (the tendency of the grid's velocity component, m-array) is updated in place; returns the sections that acted."""
    nx, ny, nz, dx, dy = g.nx, g.ny, g.nz, g.dx, g.dy
    xh, xf = np.arange(nx + 1) * dx, (np.arange(nx + 1) + 0.5) * dx
    yh, yf = np.arange(ny + 1) * dy, (np.arange(ny + 1) + 0.5) * dy
    zf, zh = g.zf[1:nz + 2], g.zh[1:nz + 2]
    d = {"u": np.array([1., 0., 0.]), "v": np.array([0., 1., 0.]), "w": np.array([0., 0., 1.])}[grid]
    acted = 0
    for s in range(S["n"]):
        i, j, k = (int(c) for c in S["cell"][s])
        fac = int(S["fac"][s]) - 1
        norm, z0, z0h = facets["norm"][fac], facets["z0"][fac], facets["z0h"][fac]
        tair = 0.
        if S["comprec"][s] or lnorec:
            if grid == "u":
                uvec = np.array([u0[k, j, i], 0.25 * (v0[k, j, i] + v0[k, j + 1, i] + v0[k, j, i - 1] + v0[k, j + 1, i - 1]),
                                 0.25 * (w0[k, j, i] + w0[k + 1, j, i] + w0[k, j, i - 1] + w0[k + 1, j, i - 1])])
                nb = (k, j, i - 1)
            else:      # interp_velocity_v and _w are the same expressions (:1752-1777)
                uvec = np.array([0.25 * (u0[k, j, i] + u0[k, j, i + 1] + u0[k, j - 1, i] + u0[k, j - 1, i + 1]), v0[k, j, i],
                                 0.25 * (w0[k, j, i] + w0[k + 1, j, i] + w0[k, j - 1, i] + w0[k + 1, j - 1, i])])
                nb = (k, j - 1, i) if grid == "v" else (k - 1, j, i)
            if iwallmom == 2:
                c0, c1 = (k, j, i), nb
                tair = 0.5 * (thl0[c0] * mask_c[c0] * (2. - mask_c[c1]) + thl0[c1] * mask_c[c1] * (2. - mask_c[c0]))
            dist = S["dist"][s]
        else:
            p, r = S["recpt"][s], S["recids"][s]
            uvec = np.array([_trilinear(u0, r[0], xh, yf, zf, p), _trilinear(v0, r[1], xf, yh, zf, p), _trilinear(w0, r[2], xf, yf, zh, p)])
            if iwallmom == 2:
                tair = _trilinear(thl0, r[3], xf, yf, zf, p)
            dist = S["dist"][s] + np.linalg.norm(np.array([p[0] - xf[i - 1], p[1] - yf[j - 1], p[2] - zf[k - 1]]))
        if np.log(dist / z0) <= 1.:
            continue
        if np.all(np.abs(uvec) < EPS1):
            continue
        span = np.cross(norm, uvec)
        if np.all(np.abs(span) < EPS1):
            continue
        span = span / np.linalg.norm(span)
        strm = np.cross(span, norm)
        utan = float(np.dot(uvec, strm))
        ctm = _ctm_stability(utan, dist, z0, z0h, tair, facets["tsurf"][fac], prt) if iwallmom == 2 else (FKAR / np.log(dist / z0)) ** 2
        stress = ctm * utan ** 2
        if S["comprec"][s]:
            stress_dir = float(np.dot(d, strm)) * stress
        else:
            a_is = float(np.dot(d, strm))
            stress_dir = float(np.linalg.norm(np.array([a_is * norm[0] * stress, a_is * norm[1] * stress, a_is * norm[2] * stress])))
        stress_dir = np.copysign(abs(stress_dir), float(np.dot(uvec, d)))
        rhs[k, j, i] = rhs[k, j, i] - stress_dir * S["area"][s] / (dx * dy * g.dzf[k])
        acted += 1
    return acted


def _heat_flux(utan, dist, z0, z0h, tair, tsurf, prt, htc_out=None):
    """heat_transfer_coef_flux, src/modibm.f90:1920-1986 -> flux."""
    b1, b2, dm, dh = 9.4, 4.7, 7.4, 5.3
    dT = tair - tsurf
    ri0 = GRAV * dist * dT / (tsurf * utan ** 2)
    logdz, logzh, sqdz, fkar2 = np.log(dist / z0), np.log(z0 / z0h), np.sqrt(dist / z0), FKAR ** 2

    def F(ri):
        if ri > 0.:
            fm = 1. / (1. + b2 * ri) ** 2
            return fm, fm
        cm = (dm * fkar2) / (logdz ** 2) * b1 * sqdz
        ch = (dh * fkar2) / (logdz ** 2) * b1 * sqdz
        return 1. - (b1 * ri) / (1. + cm * np.sqrt(abs(ri))), 1. - (b1 * ri) / (1. + ch * np.sqrt(abs(ri)))
    fm, fh = F(ri0)
    M = prt * logdz * np.sqrt(fm) / fh
    ri1 = ri0 - ri0 * prt * logzh / (prt * logzh + M)
    fm, fh = F(ri1)
    M = prt * logdz * np.sqrt(fm) / fh
    dTrough = dT * 1. / (prt * logzh / M + 1.)
    cth = fkar2 / (logdz * logdz) * fh / prt
    flux = abs(utan) * cth * dTrough
    if htc_out is not None:      # :1975-1979
        htc_out[0] = flux / (abs(utan) * dT) if abs(abs(utan) * dT) > 0. else 0.
    return flux


def wallfunheat(g, S, facets, u0, v0, w0, thl0, thlp, prt=0.71, lnorec=False, prescribed=None, moist=None):
    """wallfunheat (src/modibm.f90:1436-1607): thlp (m-array) updated in place.  Sensible part: iwalltemp = 2 unless
    `prescribed` holds iwalltemp = 1's fluxes {alignment of the facet normal: flux} (:1508-1524).  Latent part (:1556-1600) when
    `moist` = dict(qt0=, qtp= (updated in place), iwallmoist=, prescribed= {alignment: flux} for iwallmoist = 1): on the vegetated
    facets (facets["lgr"]), moist_flux (:1989) of the air's humidity against facets["qsat"] / ["hurel"] through the aerodynamic
    resistance 1 / (htc |utan|) and facets["resc"] / ["ress"]."""
    from udcore.facets import alignment
    nx, ny, nz, dx, dy = g.nx, g.ny, g.nz, g.dx, g.dy
    xh, xf = np.arange(nx + 1) * dx, (np.arange(nx + 1) + 0.5) * dx
    yh, yf = np.arange(ny + 1) * dy, (np.arange(ny + 1) + 0.5) * dy
    zf, zh = g.zf[1:nz + 2], g.zh[1:nz + 2]
    acted = 0
    for s in range(S["n"]):
        i, j, k = (int(c) for c in S["cell"][s])
        fac = int(S["fac"][s]) - 1
        norm, z0, z0h = facets["norm"][fac], facets["z0"][fac], facets["z0h"][fac]
        if S["comprec"][s] or lnorec:
            uvec = np.array([0.5 * (u0[k, j, i] + u0[k, j, i + 1]), 0.5 * (v0[k, j, i] + v0[k, j + 1, i]), 0.5 * (w0[k, j, i] + w0[k + 1, j, i])])
            tair = thl0[k, j, i]
            qtair = moist["qt0"][k, j, i] if moist else 0.
            dist = S["dist"][s]
        else:
            p, r = S["recpt"][s], S["recids"][s]
            uvec = np.array([_trilinear(u0, r[0], xh, yf, zf, p), _trilinear(v0, r[1], xf, yh, zf, p), _trilinear(w0, r[2], xf, yf, zh, p)])
            tair = _trilinear(thl0, r[3], xf, yf, zf, p)
            qtair = _trilinear(moist["qt0"], r[3], xf, yf, zf, p) if moist else 0.
            dist = S["dist"][s] + np.linalg.norm(np.array([p[0] - xf[i - 1], p[1] - yf[j - 1], p[2] - zf[k - 1]]))
        if np.log(dist / z0) <= 1.:
            continue
        if np.all(np.abs(uvec) < EPS1):
            continue
        span = np.cross(norm, uvec)
        if np.all(np.abs(span) < EPS1):
            continue
        span = span / np.linalg.norm(span)
        strm = np.cross(span, norm)
        utan = float(np.dot(uvec, strm))
        htc = [0.]
        flux = prescribed[alignment(norm)] if prescribed is not None else _heat_flux(utan, dist, z0, z0h, tair, facets["tsurf"][fac], prt, htc)
        thlp[k, j, i] = thlp[k, j, i] - flux * S["area"][s] / (dx * dy * g.dzh[k])
        if moist and facets["lgr"][fac]:
            if moist["iwallmoist"] == 1:
                flux = moist["prescribed"][alignment(norm)]
            elif abs(htc[0] * abs(utan)) > 0.:      # (else `flux` keeps the sensible one, as in the reference)
                resa, cveg, qw = 1. / (htc[0] * abs(utan)), 0.8, facets["qsat"][fac]
                flux = min(0., cveg * (qtair - qw) / (resa + facets["resc"][fac]) + (1 - cveg) * (qtair - qw * facets["hurel"][fac]) / (resa + facets["ress"][fac]))
            moist["qtp"][k, j, i] = moist["qtp"][k, j, i] - flux * S["area"][s] / (dx * dy * g.dzh[k])
        acted += 1
    return acted
