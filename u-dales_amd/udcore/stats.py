"""Statistics on top of the device core: the time-averaged 3-D output of tdump (src/modstatsdump.f90, &OUTPUT ltdump).

The device accumulates (udc_stats_sample); this class keeps the reference's two clocks -- tsamplep since the last sample,
tstatsdumpp since the last dump (:802-811, 1393-1399, 1723-1729) -- and turns the accumulators into the reference's
output variables (:1557-1645: means, and <ab> - <a><b> for fluxes and variances).  `step()` is called after every
substep, like `statsdump` at src/program.f90:205.  Files: ``tdump.<expnr>.npz`` (one array per variable and dump, float64)
and, when scipy is importable, NetCDF-3 ``tdump.<expnr>.nc`` with the reference's variable names (the reference writes
NetCDF-4 through nf90; this image has no NetCDF library).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import lib as L

# include/udcore.h
MOM = ["umt", "vmt", "wmt", "pt", "utc", "vtc", "wtc", "uutc", "vvtc", "wwtc", "uwtik", "vwtjk", "uvtij",
       "utik", "wtik", "vtjk", "wtjk", "utij", "vtij"]
ST_THL, ST_QT, ST_SV, ST_SV_STRIDE = len(MOM), len(MOM) + 4, len(MOM) + 8, 5


class TDump:
    def __init__(self, core, tsample, tstatsdump, tstatstart=0., wdir=None, expnr=0):
        self.core, self.tsample, self.tstatsdump, self.tstatstart = core, float(tsample), float(tstatsdump), float(tstatstart)
        self.tsamplep, self.tstatsdumpp = 0., 0.
        self.wdir, self.expnr = wdir, expnr
        self.nsamples, self.dumps = 0, []
        L._check(core.lib.udc_stats_enable(core.h, 1), "udc_stats_enable")

    # ---- device accumulators
    def get(self, sid):
        g = self.core.g
        a = np.zeros(g.mshape())
        lb, ub = self.core._bounds(a)
        L._check(self.core.lib.udc_stats_get(self.core.h, int(sid), a.ctypes.data_as(L.DP), lb, ub), "udc_stats_get")
        return a[1:, 1:-1, 1:-1]                      # levels kb..ke+kh, interior i, j

    def accumulators(self):
        c = self.core
        a = {n: self.get(q) for q, n in enumerate(MOM)}
        if getattr(c, "ltempeq", False):
            for q, n in enumerate(("thlt", "thltk", "wthltk", "thlthlt")):
                a[n] = self.get(ST_THL + q)
        if getattr(c, "lmoist", False):
            for q, n in enumerate(("qtt", "qttk", "wqttk", "qtqtt")):
                a[n] = self.get(ST_QT + q)
        for n in range(min(c.nsv, 4)):
            for q, nm in enumerate((f"sv{n + 1}t", f"sv{n + 1}tk", f"wsv{n + 1}tk", f"sv{n + 1}sv{n + 1}t", f"sv{n + 1}sgst")):
                a[nm] = self.get(ST_SV + ST_SV_STRIDE * n + q)
        return a

    def output(self):
        """The tdump variables (names of initstatsdump, src/modstatsdump.f90:286-320) on levels kb..ke."""
        a, c = self.accumulators(), self.core
        o = {"ut": a["umt"], "vt": a["vmt"], "wt": a["wmt"], "pt": a["pt"],
             "upwpt": a["uwtik"] - a["utik"] * a["wtik"], "vpwpt": a["vwtjk"] - a["vtjk"] * a["wtjk"],
             "upvpt": a["uvtij"] - a["utij"] * a["vtij"],
             "upuptc": a["uutc"] - a["utc"] * a["utc"], "vpvptc": a["vvtc"] - a["vtc"] * a["vtc"], "wpwptc": a["wwtc"] - a["wtc"] * a["wtc"]}
        o["tketc"] = 0.5 * (o["upuptc"] + o["vpvptc"] + o["wpwptc"])
        if "thlt" in a:
            o["thlt"] = a["thlt"]
            o["wpthlpt"] = a["wthltk"] - a["wmt"] * a["thltk"]
            o["thlpthlpt"] = a["thlthlt"] - a["thlt"] * a["thlt"]
        if "qtt" in a:
            o["qtt"] = a["qtt"]
        for n in range(1, min(c.nsv, 4) + 1):
            o[f"sca{n}t"] = a[f"sv{n}t"]
            o[f"wpsca{n}pt"] = a[f"wsv{n}tk"] - a["wmt"] * a[f"sv{n}tk"]
            o[f"sca{n}psca{n}pt"] = a[f"sv{n}sv{n}t"] - a[f"sv{n}t"] * a[f"sv{n}t"]
            o[f"sv{n}sgs"] = a[f"sv{n}sgst"]
        return {k: v[:-1] for k, v in o.items()}

    # ---- the reference's clocks
    def step(self, rk3step, dt, timee):
        """After a substep (src/program.f90:205).  Returns 'sample', 'dump' or None."""
        if timee < self.tstatstart or rk3step != 3:
            return None
        what = None
        if self.tsamplep == 0. and self.tsample <= dt:
            self.tsamplep = dt
        if self.tstatsdumpp == 0. and self.tsample <= dt:
            self.tstatsdumpp = dt
        if self.tsamplep >= self.tsample:
            L._check(self.core.lib.udc_stats_sample(self.core.h, C.c_double(self.tsamplep), C.c_double(self.tstatsdumpp)), "udc_stats_sample")
            self.nsamples += 1
            self.tsamplep = dt
            what = "sample"
        else:
            self.tsamplep += dt
        if self.tstatsdumpp >= self.tstatsdump:
            self.dumps.append((timee, self.output()))
            if self.wdir:
                self.write()
            self.tstatsdumpp = dt
            what = "dump"
        else:
            self.tstatsdumpp += dt
        return what

    def write(self):
        base = os.path.join(self.wdir, f"tdump.{self.expnr:03d}")
        flat = {"time": np.array([t for t, _ in self.dumps])}
        for q, (_, o) in enumerate(self.dumps):
            for k, v in o.items():
                flat[f"{k}.{q}"] = v
        np.savez(base + ".npz", **flat)
        try:
            from scipy.io import netcdf_file
        except Exception:      # noqa: BLE001
            return
        g = self.core.g
        with netcdf_file(base + ".nc", "w") as f:
            f.createDimension("time", len(self.dumps)); f.createDimension("zt", g.nz)
            f.createDimension("yt", g.ny); f.createDimension("xt", g.nx)
            f.createVariable("time", "d", ("time",))[:] = flat["time"]
            for k in self.dumps[0][1]:
                var = f.createVariable(k, "f", ("time", "zt", "yt", "xt"))
                for q, (_, o) in enumerate(self.dumps):
                    var[q] = o[k].astype(np.float32)
