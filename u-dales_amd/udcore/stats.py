"""Statistics on top of the device core: the time-averaged 3-D output of tdump and the x-, y- and time-averaged profiles
of xytdump (src/modstatsdump.f90, &OUTPUT ltdump / lxytdump).

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


# rows of udc_stats_xyt, the reference's output variables (initstatsdump, src/modstatsdump.f90:249-271)
XYT = ["uxyt", "vxyt", "wxyt", "thlxyt", "qtxyt", "pxyt", "upwpxyt", "wpthlpxyt", "vpwpxyt", "upvpxyt", "uwxyt", "wthlxyt", "uvxyt",
       "vwxyt", "wwxyt", "usgsxyt", "thlsgsxyt", "vsgsxyt", "thlpthlptxy", "upuptxyc", "vpvptxyc", "wpwptxyc", "tketxyc"]


# rows of udc_stats_yt (initstatsdump, src/modstatsdump.f90:165-199)
YT = ["uyt", "vyt", "wyt", "thlyt", "qtyt", "sca1yt", "sca2yt", "sca3yt", "upwpyt", "wpthlpyt", "wpqtpyt", "wpsca1tpyt", "wpsca2tpyt",
      "wpsca3tpyt", "uwyt", "wthlyt", "wqtyt", "wsca1yt", "wsca2yt", "wsca3yt", "upupyt", "wpwpyt", "thlpthlpyt", "qtpqtpyt",
      "sca1tpsca1pyt", "sca2tpsca2pyt", "sca3tpsca3pyt", "usgsyt", "wsgsyt", "thlsgsyt", "qtsgsyt", "sca1sgsyt", "sca2sgsyt", "sca3sgsyt"]


# rows of udc_stats_xy / udc_stats_y: the instantaneous tables of xydump / ydump (initstatsdump, src/modstatsdump.f90:215-231, 131-148)
XY = ["uxy", "vxy", "wxy", "thlxy", "qtxy", "pxy", "upwpxy", "wpthlpxy", "vpwpxy", "usgsxy", "thlsgsxy", "vsgsxy", "uwxyik", "wthlxy", "vwxy"]
Y = ["uy", "vy", "wy", "thly", "qty", "sca1y", "sca2y", "sca3y", "upwpy", "wpthlpy", "usgsy", "thlsgsy", "uwyik", "wthlyk"]


def xyt_masks(nx, ny, nz, lists, wrapx=False, wrapy=False, j0=0, nyl=None):
    """createmasks' fluid masks (src/modibm.f90:2141-2190) for udc_stats_set_masks: (bits[nz, nyl, nx] uint8, counts[7, nz], forced[7])
    from the solid point lists {grid: (solid[n, 3], ...)} (global 1-based i, j, k).  Bit order IIu, IIv, IIw, IIc, IIuw,
    IIvw, IIuv.  The point masks' ghost cells are fluid unless the direction wraps (one the reference run splits over
    ranks, as for udc_set_ibm_mask_wrap).  avexy_ibm's rule for a first level without fluid points (src/modmpi.f90:646-649,
    lnan = .false.: the unmasked sum over the count of level ke) is applied here: such a level gets all its bits set and the
    count of the last level.  j0, nyl: this rank's rows of the bits (the counts are global)."""
    pt = {}
    for q in "uvwc":
        a = np.ones((nz + 2, ny + 2, nx + 2), dtype=np.uint8)
        if q in lists and len(lists[q][0]):
            i, j, k = np.asarray(lists[q][0]).T
            a[k, j, i] = 0
        if wrapx:
            a[:, :, 0], a[:, :, -1] = a[:, :, -2], a[:, :, 1]
        if wrapy:
            a[:, 0], a[:, -1] = a[:, -2], a[:, 1]
        pt[q] = a
    pt["w"][1] = 0                                             # IIw(:, :, kb) = 0

    def at(a, di=0, dj=0, dk=0):
        return a[1 + dk:nz + 1 + dk, 1 + dj:ny + 1 + dj, 1 + di:nx + 1 + di]
    II = [at(pt["u"]).copy(), at(pt["v"]).copy(), at(pt["w"]).copy(), at(pt["c"]).copy()]
    uw = at(pt["u"]) * at(pt["u"], dk=-1) * at(pt["w"]) * at(pt["w"], di=-1)
    vw = at(pt["v"]) * at(pt["v"], dk=-1) * at(pt["w"]) * at(pt["w"], dj=-1)
    uw[0], vw[0] = 0, 0                                        # IIuw(:, :, kb) = IIvw(:, :, kb) = 0
    II += [uw, vw, at(pt["u"]) * at(pt["u"], dj=-1) * at(pt["v"]) * at(pt["v"], di=-1)]
    counts = np.array([m.sum(axis=(1, 2)) for m in II], dtype=np.int32)
    forced = np.zeros(7, dtype=np.int32)
    for q, m in enumerate(II):
        if counts[q, 0] == 0:
            m[0] = 1
            counts[q, 0] = counts[q, nz - 1]
            forced[q] = 1
    bits = np.zeros((nz, ny, nx), dtype=np.uint8)
    for q, m in enumerate(II):
        bits |= (m.astype(np.uint8) << q)
    nyl = ny if nyl is None else nyl
    return np.ascontiguousarray(bits[:, j0:j0 + nyl]), np.ascontiguousarray(counts), forced


class TDump:
    def __init__(self, core, tsample, tstatsdump, tstatstart=0., wdir=None, expnr=0, xyt=False, ibm_lists=None, wrap=(False, False),
                 jtot=None, j0=0, nyl=None, yt=False, xy=False, y=False):
        """xyt: also xytdump's profiles; ibm_lists (udcore.ibm.read_ibm) when the deck has obstacles, wrap = the directions
        the reference run of the deck splits over ranks, jtot / j0 / nyl = global rows / this rank's first row / its row count for y-slab runs."""
        self.core, self.tsample, self.tstatsdump, self.tstatstart = core, float(tsample), float(tstatsdump), float(tstatstart)
        self.tsamplep, self.tstatsdumpp = 0., 0.
        self.wdir, self.expnr = wdir, expnr
        self.nsamples, self.dumps, self.xyt_on, self.xyt_dumps = 0, [], bool(xyt), []
        self.yt_on, self.yt_dumps = bool(yt), []
        self.mint = False             # mintdump (src/modstatsdump.f90:1670-1684): ut, vt, wt, thlt, qtt, pt of the same accumulators
        L._check(core.lib.udc_stats_enable(core.h, 1 + (2 if xyt else 0) + (4 if yt else 0) + (8 if xy else 0) + (16 if y else 0)), "udc_stats_enable")
        if (xyt or yt or xy or y) and ibm_lists is not None:
            g = core.g
            bits, counts, forced = xyt_masks(g.nx, jtot or g.ny, g.nz, ibm_lists, wrap[0], wrap[1], j0=j0, nyl=nyl or g.ny)
            L._check(core.lib.udc_stats_set_masks(core.h, bits.ctypes.data_as(C.POINTER(C.c_ubyte)), counts.ctypes.data_as(C.POINTER(C.c_int))),
                     "udc_stats_set_masks")
            L._check(core.lib.udc_stats_set_forced(core.h, forced.ctypes.data_as(C.POINTER(C.c_int))), "udc_stats_set_forced")

    def yt(self):
        """ytdump's table {name: field[ktot, itot]} (src/modstatsdump.f90:1513-1551); -999 in columns without fluid points."""
        g = self.core.g
        t = np.zeros((len(YT), g.nz, g.nx))
        L._check(self.core.lib.udc_stats_yt(self.core.h, t.ctypes.data_as(L.DP)), "udc_stats_yt")
        c = self.core
        have = lambda n: (("thl" not in n) or getattr(c, "ltempeq", False)) and (("qt" not in n) or getattr(c, "lmoist", False)) and \
            not any(f"sca{q}" in n and c.nsv < q for q in (1, 2, 3))      # noqa: E731
        return {n: t[q] for q, n in enumerate(YT) if have(n)}

    def xy(self):
        """xydump's table of the LAST sample {name: profile[ktot]} (src/modstatsdump.f90:1330-1344)."""
        t = np.zeros((len(XY), self.core.g.nz))
        L._check(self.core.lib.udc_stats_xy(self.core.h, t.ctypes.data_as(L.DP)), "udc_stats_xy")
        return {n: t[q] for q, n in enumerate(XY)}

    def y(self):
        """ydump's table of the LAST sample {name: field[ktot, itot]} (src/modstatsdump.f90:1303-1316)."""
        g = self.core.g
        t = np.zeros((len(Y), g.nz, g.nx))
        L._check(self.core.lib.udc_stats_y(self.core.h, t.ctypes.data_as(L.DP)), "udc_stats_y")
        return {n: t[q] for q, n in enumerate(Y)}

    def xyt(self):
        """xytdump's table {name: profile[ktot]} (src/modstatsdump.f90:1437-1460); rows the deck does not have (thl, qt) left out."""
        t = np.zeros((len(XYT), self.core.g.nz))
        L._check(self.core.lib.udc_stats_xyt(self.core.h, t.ctypes.data_as(L.DP)), "udc_stats_xyt")
        skip = set()
        if not getattr(self.core, "ltempeq", False):
            skip |= {"thlxyt", "wpthlpxyt", "wthlxyt", "thlsgsxyt", "thlpthlptxy"}
        if not getattr(self.core, "lmoist", False):
            skip.add("qtxyt")
        return {n: t[q] for q, n in enumerate(XYT) if n not in skip}

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
            if self.xyt_on:
                self.xyt_dumps.append((timee, self.xyt()))
            if self.yt_on:
                self.yt_dumps.append((timee, self.yt()))
            if self.wdir:
                self.write()
            self.tstatsdumpp = dt
            what = "dump"
        else:
            self.tstatsdumpp += dt
        return what

    def write(self):
        """The reference's naming (src/modstatsdump.f90:320-365): the 3-D sets are per rank -- tdump.<myidx>.<myidy>.<expnr>, each rank
        its own slab --, the tables of global reductions (xytdump, ytdump) are written once, by rank 0.  A single-rank run keeps the
        short names tdump.<expnr> / mintdump.<expnr>."""
        rank, nranks = getattr(self.core, "rank", 0), getattr(self.core, "nranks", 1)
        tag = f"000.{rank:03d}." if nranks > 1 else ""
        if self.yt_on and rank == 0:
            np.savez(os.path.join(self.wdir, f"ytdump.{self.expnr:03d}.npz"), time=np.array([t for t, _ in self.yt_dumps]),
                     **{k: np.array([o[k] for _, o in self.yt_dumps]) for k in self.yt_dumps[0][1]})
        if self.mint:
            keep = [k for k in ("ut", "vt", "wt", "thlt", "qtt", "pt") if k in self.dumps[0][1]]
            np.savez(os.path.join(self.wdir, f"mintdump.{tag}{self.expnr:03d}.npz"), time=np.array([t for t, _ in self.dumps]),
                     **{k: np.array([o[k] for _, o in self.dumps]).astype(np.float32) for k in keep})
        if self.xyt_on and rank == 0:
            np.savez(os.path.join(self.wdir, f"xytdump.{self.expnr:03d}.npz"), time=np.array([t for t, _ in self.xyt_dumps]),
                     **{k: np.array([o[k] for _, o in self.xyt_dumps]) for k in self.xyt_dumps[0][1]})
        base = os.path.join(self.wdir, f"tdump.{tag}{self.expnr:03d}")
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
