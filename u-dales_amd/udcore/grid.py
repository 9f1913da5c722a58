"""Grid metrics and cold-start fields, as the reference's host code builds them.

* `Grid.from_deck`: src/modglobal.f90:708-762 (dx, dy, zh, dzf, dzh from the zf column of prof.inp).
* `cold_start`:     src/modstartup.f90:1088-1290 (profiles -> um/vm/wm, LCG noise, u0 = um) and
                    randomize_field :2367-2396.
These run on the host once; they only produce the inputs the device library is given.
Arrays are numpy float64 shaped [k, j, i] with the reference's halo extents
(0:nx+1, 0:ny+1, 0:nz+1) for momentum fields and (-1:nx+2, ...) for kappa scalars.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .namoptions import Deck, smagorinsky_constant


@dataclass
class Grid:
    nx: int
    ny: int
    nz: int
    dx: float
    dy: float
    zf: np.ndarray      # [nz+2], index = reference k (entry 0 unused)
    zh: np.ndarray      # [nz+2]
    dzf: np.ndarray     # [nz+2] = dzf(0:nz+1)
    dzh: np.ndarray     # [nz+2], dzh[k] = dzh(k), k = 1..nz+1

    @staticmethod
    def from_levels(nx, ny, nz, xlen, ylen, zf_levels) -> "Grid":
        zf = np.zeros(nz + 2)
        zh = np.zeros(nz + 2)
        zf[1:nz + 1] = np.asarray(zf_levels, dtype=np.float64)
        zh[1] = 0.0
        for k in range(1, nz + 1):
            zh[k + 1] = zh[k] + 2.0 * (zf[k] - zh[k])
        zf[nz + 1] = zf[nz] + 2.0 * (zh[nz + 1] - zf[nz])
        dzf = np.zeros(nz + 2)
        for k in range(1, nz + 1):
            dzf[k] = zh[k + 1] - zh[k]
        dzf[nz + 1] = dzf[nz]
        dzf[0] = dzf[1]
        dzh = np.zeros(nz + 2)
        dzh[1] = 2 * zf[1]
        for k in range(2, nz + 2):
            dzh[k] = zf[k] - zf[k - 1]
        return Grid(nx, ny, nz, xlen / float(nx), ylen / float(ny), zf, zh, dzf, dzh)

    @staticmethod
    def from_deck(d: Deck) -> "Grid":
        return Grid.from_levels(d.get("DOMAIN", "itot"), d.get("DOMAIN", "jtot"), d.get("DOMAIN", "ktot"),
                                float(d.get("DOMAIN", "xlen")), float(d.get("DOMAIN", "ylen")), d.zf)

    @staticmethod
    def uniform(nx, ny, nz, dx=0.5, dy=0.5, dz=0.5) -> "Grid":
        zf = (np.arange(nz) + 0.5) * dz
        return Grid.from_levels(nx, ny, nz, nx * dx, ny * dy, zf)

    def mshape(self):
        return (self.nz + 2, self.ny + 2, self.nx + 2)

    def cshape(self):
        return (self.nz + 4, self.ny + 4, self.nx + 4)


def lcg_noise(nx, ny, j0, nyl, k, irandom=43):
    """randomize_field's perturbation for level k (1-based), rows j0+1..j0+nyl (global, 1-based)."""
    imm, ia, ic = 134456, 8121, 28411
    i = np.arange(1, nx + 1, dtype=np.int64)[None, :]
    j = np.arange(j0 + 1, j0 + nyl + 1, dtype=np.int64)[:, None]
    lin = i + nx * (j - 1) + np.int64(nx) * np.int64(ny) * np.int64(k - 1)
    state = (irandom + lin) % imm
    state = (state * ia + ic) % imm
    return state.astype(np.float64) / float(imm)


def sgs_from_deck(d: Deck):
    """(sgs selector, csz, c_vreman, prandtli): src/modsubgrid.f90:65-77,117-121, modsubgriddata.f90:39-61."""
    lsmag = d.get("NAMSUBGRID", "lsmagorinsky")
    lvre = d.get("NAMSUBGRID", "lvreman")
    lles = d.get("RUN", "lles")
    if lsmag:
        sgs = 1
    elif lvre:
        sgs = 2
    elif d.get("NAMSUBGRID", "loneeqn"):
        sgs = 3                 # one-equation TKE closure, src/modsubgrid.f90:363 (needs DynCore.set_tke)
    else:
        sgs = 0
    if (lsmag or lvre or sgs == 3):
        lles = True
    if not lles:
        sgs = 0
    csz = smagorinsky_constant(float(d.get("NAMSUBGRID", "cs")), float(d.get("NAMSUBGRID", "cf")))
    return sgs, csz, float(d.get("NAMSUBGRID", "c_vreman")), 1. / float(d.get("NAMSUBGRID", "Prandtl"))


def scalar_profiles(g: Grid, d: Deck, nsv, scal_a=None, scal_b=None):
    """svprof(k, n) as [nsv][nz+2] (index = reference k): the columns of scalar.inp.xxx when the case has one
    (src/modstartup.f90:1539-1548), else the linear stand-in profile of the oracle driver (&ORACLE scal_a, scal_b)."""
    nz = g.nz
    out = []
    sp = getattr(d, "svprof", None)
    a = float(d.get("ORACLE", "scal_a")) if scal_a is None else scal_a
    b = float(d.get("ORACLE", "scal_b")) if scal_b is None else scal_b
    zsize = g.zh[nz + 1]
    for n in range(nsv):
        p = np.zeros(nz + 2)
        for k in range(1, nz + 1):
            p[k] = sp[n][k - 1] if sp else b + a * float(n + 1) * g.zf[k] / zsize
        out.append(p)
    return out


def cold_start(g: Grid, d: Deck, j0=0, nyl=None, nsv=0, scal_a=None, scal_b=None, pre_boundary=False):
    """Initial um, vm, wm (= u0, v0, w0), and scalars for rows j0+1..j0+nyl of the global grid.

    Returns dict of arrays with halos; x ghosts periodic, y ghosts periodic when the slab is
    the whole domain (otherwise left for the halo exchange), k ghost rows as `boundary` sets them --
    or, with pre_boundary, as readinitfiles leaves them (src/modstartup.f90:1150-1210: nothing above level ke but a copy of
    thl0's last level): the state the reference's start-up `thermodynamics` sees (:1601) BEFORE program.f90:118 calls
    `boundary`; DynCore.start_up() then does those two calls in that order.
    """
    nx, ny, nz = g.nx, g.ny, g.nz
    nyl = ny if nyl is None else nyl
    shape = (nz + 2, nyl + 2, nx + 2)
    um = np.zeros(shape)
    vm = np.zeros(shape)
    wm = np.zeros(shape)
    for k in range(1, nz + 1):
        um[k, :, :] = d.u[k - 1]
        vm[k, :, :] = d.v[k - 1]
    if d.get("RUN", "lrandomize"):
        randu = float(d.get("RUN", "randu"))
        irandom = int(d.get("RUN", "irandom"))
        krand = min(int(d.get("RUN", "krand")), nz)
        for k in range(1, krand + 1):
            ran = lcg_noise(nx, ny, j0, nyl, k, irandom)
            pert = (ran - 0.5) * 2.0 * randu
            um[k, 1:nyl + 1, 1:nx + 1] += pert
            vm[k, 1:nyl + 1, 1:nx + 1] += pert
            wm[k, 1:nyl + 1, 1:nx + 1] += pert
    out = {"um": um, "vm": vm, "wm": wm}
    if pre_boundary:      # ekm = ekh = numol on the levels, ekh(ke+1) = ekh(ke) (src/modstartup.f90:1164-1165, 1190): what the flux tops of
        # the start-up `boundary` divide by
        ek = np.zeros(shape)
        ek[1:nz + 1] = 1.5e-5
        out["ekm"] = ek
        out["ekh"] = ek.copy()
        out["ekh"][nz + 1] = 1.5e-5
    open_x = int(d.get("BC", "BCxm")) in (2, 3)      # the x ghost columns keep the profile readinitfiles put there (src/modstartup.f90:1155-1177)
    for a in (um, vm, wm):
        if not open_x:
            a[:, :, 0] = a[:, :, nx]
            a[:, :, nx + 1] = a[:, :, 1]
        if nyl == ny:
            a[:, 0, :] = a[:, nyl, :]
            a[:, nyl + 1, :] = a[:, 1, :]
    # boundary(): w(kb) = 0, free-slip top rows (src/modboundary.f90:165-178)
    wm[1, :, :] = 0.
    bctopm = int(d.get("BC", "BCtopm"))
    if not pre_boundary:
        for a in (um, vm):
            a[nz + 1] = a[nz] if bctopm != 2 else -a[nz]
    wm[nz + 1] = 0.
    out["u0"], out["v0"], out["w0"] = um.copy(), vm.copy(), wm.copy()
    svprof = scalar_profiles(g, d, nsv, scal_a, scal_b)
    for n in range(nsv):
        c = np.zeros((nz + 4, nyl + 4, nx + 4))
        for k in range(1, nz + 1):
            c[k + 1] = svprof[n][k]
        if not getattr(d, "svprof", None):      # (the reference leaves the sub-floor planes of a scalar.inp start at zero)
            c[1] = c[2]
            c[0] = c[2]
        else:      # ... and fills ib-1 : ie+1, jb-1 : je+1 only (src/modstartup.f90:1561-1570): the outermost ghost columns stay zero,
            # which is what a convective outlet (BCxs = 2) keeps in its second ghost cell for ever
            c[:, :, 0] = 0.; c[:, :, -1] = 0.; c[:, 0, :] = 0.; c[:, -1, :] = 0.
        if pre_boundary:
            pass
        elif int(d.get("BC", "BCtops")) == 2:     # valuetopscal with sv_top = svprof(ke), as `boundary` leaves it
            c[nz + 2] = 2 * svprof[n][nz] - c[nz + 1]
        else:                                   # fluxtopscal with the start value ekh = numol (src/modboundary.f90:1532)
            w = d.get("BC", "wsvtopdum")
            w = list(w) if isinstance(w, (list, tuple)) else [w]
            flux = float(w[n]) if n < len(w) else 0.
            c[nz + 2] = c[nz + 1] + g.dzh[nz + 1] * flux / ((1. / g.dzh[nz + 1]) * (0.5 * (g.dzf[nz] * 1.5e-5 + g.dzf[nz + 1] * 1.5e-5)))
        c[nz + 3] = c[nz + 2]
        if int(d.get("BC", "BCxs")) == 2 and not pre_boundary:       # the start-up call of `boundary`: xsi_profile (src/modboundary.f90:844-861) also runs on
            # level ke+1, where svprof is zero -- the inlet ghosts of the top ghost level mirror about 0 (no stencil reads them)
            c[nz + 2, 2:-2, 1] = -c[nz + 2, 2:-2, 2]
            c[nz + 2, 2:-2, 0] = -c[nz + 2, 2:-2, 1]
        out[f"sv0_{n}"] = c
        out[f"svm_{n}"] = c.copy()
    if d.get("NAMSUBGRID", "loneeqn") and not (d.get("NAMSUBGRID", "lsmagorinsky") or d.get("NAMSUBGRID", "lvreman")):
        # e120 = e12m = max(e12prof(k), e12min), src/modstartup.f90:1140-1169; ghosts: zero below, e12min above (boundary)
        e = np.zeros(shape)
        for k in range(1, nz + 1):
            e[k] = max(d.tke[k - 1], 5.e-5)
        if not pre_boundary:
            e[nz + 1] = 5.e-5
        out["e120"], out["e12m"] = e, e.copy()
    if d.get("PHYSICS", "ltempeq"):
        # thl0 = thlm = thlprof(k), ghosts as src/modstartup.f90:1156-1208, then boundary's top condition
        t = np.zeros(shape)
        for k in range(1, nz + 1):
            t[k] = d.thl[k - 1]
        t[0] = t[1]
        if int(d.get("BC", "BCtopT")) == 2 and not pre_boundary:
            t[nz + 1] = 2 * float(d.get("BC", "thl_top")) - t[nz]
        else:
            t[nz + 1] = t[nz]          # non-zero wttop needs ekh: re-imposed on the device after the first closure
        out["thl0"], out["thlm"] = t, t.copy()
    if d.get("PHYSICS", "lmoist"):
        # qt0 = qtm = qtprof(k) (src/modstartup.f90:1160-1161); the floor ghost stays zero (nothing ever writes it),
        # the top ghost comes from boundary (src/modboundary.f90:222-231)
        q = np.zeros(shape)
        for k in range(1, nz + 1):
            q[k] = d.qt[k - 1]
        if pre_boundary:
            pass
        elif int(d.get("BC", "BCtopq")) == 2:
            q[nz + 1] = 2 * float(d.get("BC", "qt_top")) - q[nz]
        else:
            q[nz + 1] = q[nz]
        out["qt0"], out["qtm"] = q, q.copy()
    return out
