"""Per-level forcings on top of the device core: large-scale subsidence (lstend), nudging and the gravity-wave
sponge (grwdamp).

The reference computes them on the host from slab averages; they all have the form
``tendency(i,j,k) += A(k) + B(k) * field(i,j,k)``.  The device supplies the averages (udc_slab_average) and applies
registered (A, B) tables inside the fused substep (udc_set_level_forcing); the per-level arithmetic below follows

  lstend   src/modforces.f90:719-822   (subsidence of thl and the scalars with whls; the large-scale gradients
                                        d*dxls/d*dyls have no input in this snapshot and are zero; lmomsubs is
                                        not a namelist variable, so momentum is untouched)
  nudge    src/modforces.f90:824-860   (u, v towards uprof, vprof when lnudgevel; thl towards thlprof)
  grwdamp  src/modboundary.f90:1447-1488 (igrw_damp 1, 2, 3; tsc and ksp of initboundary :45-59)
  whls     src/modstartup.f90:2125-2129

Usage: ``ls = LevelForcings(core, deck)``; call ``ls.update()`` before every substep (after the previous one's
boundary, where the reference's diagfld takes its averages).
"""
from __future__ import annotations

import math

import numpy as np

from . import lib as L


class LevelForcings:
    def __init__(self, core, deck):
        self.core, g = core, core.g
        nz = g.nz
        self.nz = nz
        ph = lambda n: deck.get("PHYSICS", n)   # noqa: E731
        self.igrw = int(ph("igrw_damp"))
        self.lnudge = bool(ph("lnudge"))
        self.lnudgevel = bool(ph("lnudgevel"))
        self.tnudge = float(ph("tnudge"))
        self.nnudge = int(ph("nnudge"))
        self.lcoriol = bool(ph("lcoriol"))
        from .namoptions import GEODAMPTIME
        self.geodamptime = GEODAMPTIME
        self.ltempeq = bool(ph("ltempeq"))
        self.lmoist = bool(ph("lmoist"))
        # profiles indexed by the reference's k (entry 0 unused)
        f = lambda a: np.concatenate(([0.], np.asarray(a, dtype=float)[:nz]))   # noqa: E731
        self.uprof, self.vprof, self.thlprof = f(deck.u), f(deck.v), f(deck.thl)
        self.ug, self.vg = f(deck.ug), f(deck.vg)
        self.qtprof = f(deck.qt)
        from .grid import scalar_profiles
        self.svprof = scalar_profiles(g, deck, getattr(core, "nsv", 0))
        # large-scale moisture gradients / tendency (lscale.inp columns 7-9, src/modstartup.f90:2060-2097)
        self.dqtdxls, self.dqtdyls, self.dqtdtls = (f(getattr(deck, n, np.zeros(nz))) for n in ("dqtdxls", "dqtdyls", "dqtdtls"))
        self.qtls = self.lmoist and bool(np.any(self.dqtdxls) or np.any(self.dqtdyls) or np.any(self.dqtdtls))
        wfls = f(getattr(deck, "wfls", np.zeros(nz)))
        dzf, dzh = g.dzf, g.dzh
        whls = np.zeros(nz + 2)                      # src/modstartup.f90:2125-2129
        for k in range(2, nz + 1):
            whls[k] = (wfls[k] * dzf[k - 1] + wfls[k - 1] * dzf[k]) / (2 * dzh[k])
        whls[nz + 1] = (wfls[nz] + dzf[nz] * (wfls[nz] - wfls[nz - 1]) / dzh[nz])
        self.whls = whls
        self.subsidence = bool(np.any(wfls != 0.))
        # sponge: initboundary, src/modboundary.f90:45-59 (rnu0 = 2.75e-3)
        kmax = nz
        ksp = int(deck.get("DOMAIN", "ksp"))                # &DOMAIN ksp (src/modstartup.f90:118); -1: the default below
        self.ksp = ksp if ksp != -1 else max(min(3 * kmax // 4, kmax - 15), 1)
        if not 1 <= self.ksp < nz:
            raise ValueError(f"&DOMAIN ksp = {self.ksp} outside 1..{nz - 1}")
        tsc = np.zeros(nz + 2)
        zspb, zspt = g.zf[self.ksp], g.zf[nz]
        pi = 3.141592653589793116
        for k in range(self.ksp, nz + 1):
            tsc[k] = 2.75e-3 * math.sin(0.5 * pi * (g.zf[k] - zspb) / (zspt - zspb)) ** 2
        self.tsc = tsc
        # fixuinf1 / fixuinf2 (src/modforces.f90:174-288; &PHYSICS ifixuinf, lvinf, tscale; &INLET Uinf, Vinf, inletav)
        self.ifixuinf = int(ph("ifixuinf"))
        self.lvinf = bool(ph("lvinf"))
        self.tscale = float(ph("tscale"))
        self.Uinf, self.Vinf, self.inletav = (float(deck.get("INLET", n)) for n in ("Uinf", "Vinf", "inletav"))
        self.freestreamav, self.dgdt, self._pending = 0., 0., 0.      # src/modglobal.f90:261, src/modfields.f90:396
        # shiftedPBCs (src/modforces.f90:953-980; &BC ds)
        self.ds = float(deck.get("BC", "ds"))
        xlen = g.nx * g.dx
        self.shift_a = 0.5 * pi * self.ds / (0.5 * xlen)
        half = g.nx // 2
        xh = [(i - 1) * g.dx for i in range(0, g.nx + 2)]            # xh(i), src/modglobal.f90:770-776
        self.shift_sinx = np.array([math.sin(pi * (xh[i] - xh[half]) / (0.5 * xlen)) if i > half else 0. for i in range(1, g.nx + 1)])
        self.active = self.subsidence or self.lnudge or self.igrw != 0 or self.qtls or self.ifixuinf in (1, 2) or self.ds > 0

    def tables(self, av, rk3step=None, dt=None):
        """av: dict of slab averages indexed by the reference's k (entries 1..nz+1).  Returns {(tend, when): [src, A, B]}
        with A, B indexed 1..nz; when = 0 for lstend and nudge (before masscorr), 1 for fixuinf1 and the sponge (after
        it).  rk3step, dt: the substep about to run (fixuinf1 acts on RK stage 3 only)."""
        nz, dzh, whls = self.nz, self.core.g.dzh, self.whls
        out = {}

        def acc(tend, src=None, when=0):
            key = (tend, when)
            if key not in out:
                out[key] = [src, np.zeros(nz + 2), np.zeros(nz + 2)]
            if src is not None:
                out[key][0] = src
            return out[key]

        scal = [("thl0", "thlp")] if self.ltempeq else []
        scal += [("qt0", "qtp")] if self.lmoist else []
        scal += [(f"sv0_{n}", f"svp_{n}") for n in range(self.core.nsv)]
        # (numpy slices over k: the same elementwise arithmetic as a loop over the levels, without the Python overhead)
        K = slice(2, nz + 1)                                     # k = kb+1 .. ke
        Kp, Km = slice(3, nz + 2), slice(1, nz)                  # k+1, k-1
        if self.subsidence:                                      # lstend
            down = whls[Kp] < 0
            for name, tend in scal:
                a = np.asarray(av[name])
                A = acc(tend)[1]
                if whls[2] < 0:                                  # k = kb, src/modforces.f90:768-781
                    A[1] -= whls[2] * (a[2] - a[1]) / dzh[2]
                A[K] -= np.where(down, whls[Kp] * (a[Kp] - a[K]) / dzh[Kp], whls[K] * (a[K] - a[Km]) / dzh[K])   # :790-821
        Ka = slice(1, nz + 1)
        if self.qtls:                                            # lstend, src/modforces.f90:783,818
            A = acc("qtp")[1]
            A[Ka] += -np.asarray(av["u0"])[Ka] * self.dqtdxls[Ka] - np.asarray(av["v0"])[Ka] * self.dqtdyls[Ka] + self.dqtdtls[Ka]
        if self.lnudge:                                          # nudge
            Kn = slice(1 + self.nnudge, nz + 1)
            pairs = []
            if self.lnudgevel:
                pairs += [("u0", "up", self.uprof), ("v0", "vp", self.vprof)]
            if self.ltempeq:
                pairs.append(("thl0", "thlp", self.thlprof))
            if self.lmoist:
                pairs.append(("qt0", "qtp", self.qtprof))
            pairs += [(f"sv0_{n}", f"svp_{n}", self.svprof[n]) for n in range(self.core.nsv)]      # src/modforces.f90:840-844
            for name, tend, prof in pairs:
                A = acc(tend)[1]
                A[Kn] -= (np.asarray(av[name])[Kn] - np.asarray(prof)[Kn]) / self.tnudge
        if self.ifixuinf == 1:                                   # fixuinf1, src/modforces.f90:220-288
            if rk3step is None or dt is None:
                raise ValueError("ifixuinf = 1: LevelForcings.update needs the substep's rk3step and dt")
            on = 1. if rk3step == 3 else 0.
            A = acc("up", None, 1)[1]
            A[Ka] -= on * (1. / dt) * (av["u0"][nz] - self.Uinf)
            if self.lvinf:
                A = acc("vp", None, 1)[1]
                A[Ka] -= on * (1. / dt) * (av["v0"][nz] - self.Vinf)
        if self.igrw in (1, 2, 3):                               # grwdamp
            tsc = self.tsc
            Ks = slice(self.ksp, nz + 1)
            for name, tend, geo in (("u0", "up", self.ug), ("v0", "vp", self.vg)):
                _, A, B = acc(tend, name, 1)
                ref = np.asarray(geo)[Ks] if self.igrw == 2 else np.asarray(av[name])[Ks]
                A[Ks] += ref * tsc[Ks]
                B[Ks] -= tsc[Ks]
                if self.igrw == 1 and self.lcoriol:
                    c = (1. / (self.geodamptime * 2.75e-3)) * tsc[Ks]
                    A[Ks] += np.asarray(geo)[Ks] * c
                    B[Ks] -= c
            _, A, B = acc("wp", "w0", 1)
            B[Ks] -= tsc[Ks]
            for on, name, tend in ((self.ltempeq, "thl0", "thlp"), (self.lmoist, "qt0", "qtp")):
                if on:
                    _, A, B = acc(tend, name, 1)
                    A[Ks] += np.asarray(av[name])[Ks] * tsc[Ks]
                    B[Ks] -= tsc[Ks]
        return out

    def averages(self):
        names = ["u0", "v0"] + (["thl0"] if self.ltempeq else []) + (["qt0"] if self.lmoist else [])
        if self.subsidence or self.lnudge:      # the only two users of the scalars' slab means
            names += [f"sv0_{n}" for n in range(self.core.nsv)]
        if hasattr(self.core, "slab_averages"):
            return self.core.slab_averages(names)
        return {n: self.core.slab_average(n) for n in names}

    def capture_startup(self):
        """diagfld's slab averages as the start-up `thermodynamics` takes them (src/modstartup.f90:1601): of the state BEFORE
        program.f90:118's `boundary` has set the planes above level ke, which lstend's upward differences read.  The first
        update() uses them (DynCore.start_up(before_boundary=forcings.capture_startup))."""
        if self.active:
            self._startup_av = self.averages()

    def update(self, rk3step=None, dt=None):
        """Take the slab averages of the current state and register the tables for the next substep (rk3step, dt: that
        substep's; needed by ifixuinf only)."""
        if not self.active:
            return {}
        av = getattr(self, "_startup_av", None) or self.averages()
        self._startup_av = None
        if self.ifixuinf == 2:                                   # fixuinf2 (:174-218) + src/modtstep.f90:194-195
            if rk3step is None or dt is None:
                raise ValueError("ifixuinf = 2: LevelForcings.update needs the substep's rk3step and dt")
            core = self.core
            if self._pending != 0.:                              # dpdxl += dgdt rk3coef of the substep that just ran
                core.set_forcing(core.dpdxl + self._pending, core.dpdyl)
            if rk3step == 3:
                freestream = av["v0" if self.lvinf else "u0"][self.nz]
                self.freestreamav = freestream * dt / self.inletav + (1. - dt / self.inletav) * self.freestreamav
                self.dgdt = (1. / self.tscale) * (self.freestreamav - self.Uinf)
            self._pending = self.dgdt * (dt / (4. - rk3step))
        if self.ds > 0:
            self.core.set_shifted_pbc(self.shift_a, self.shift_sinx, av["u0"][1:self.nz + 1])
        tabs = self.tables(av, rk3step, dt)
        for (tend, when), (src, A, B) in tabs.items():
            self.core.set_level_forcing(tend, src, A[1:self.nz + 1], B[1:self.nz + 1], when)
        return tabs


def field_id(name: str) -> int:
    if name in L.FIELD_IDS:
        return L.FIELD_IDS[name]
    kind, n = name.rsplit("_", 1)
    return L.scalar_field({"sv0": L.SV0, "svm": L.SVM, "svp": L.SVP}[kind], int(n))
