"""TEST INFRASTRUCTURE -- numpy restatement of the tdump / xytdump statistics of the reference, NOT the product.

Pinned on the reference's own compiled lines: the sampling half of statsdump (src/modstatsdump.f90:514-1235) and xytdump's
final slab averages (:1404-1431) are assembled from the reference file at build time (oracle/extract_statsdump.sh; the
NetCDF output half cannot be compiled here) and run inside oracle/_ref/udales_ref; the running averages they leave in
modfields are the `st.*` / `xyt.*` records of the fixtures run_stats_16x8x12s and run_stats_ibm_16x12x10, which
tests/test_oracle_vs_reference.py::test_statistics_match_reference holds this restatement against.  Lines followed:
sampling :812-857 (interpolations, SGS fluxes), :858-927 (scalars); slab averages :1037-1056 with avexy_ibm
(src/modmpi.f90:623-664) over createmasks' masks (src/modibm.f90:2141-2190); running averages :1086-1101, :1137-1213;
output variables :1404-1460 (xytdump), :1557-1645 (tdump).

Arrays are m-arrays [nz+2, ny+2, nx+2] (index == the reference's i, j, k with ghosts) for velocities / pres0 / ekh / thl /
qt, c-arrays [nz+4, ny+4, nx+4] for the passive scalars.  Results are [nz+1, ny, nx]: levels kb..ke+kh, interior i, j.
"""
import numpy as np


def _m(a, di=0, dj=0, dk=0, nzp=None):
    """a(i+di, j+dj, k+dk) for i, j interior and k = 1..nz+1 (m-array)."""
    nz = a.shape[0] - 2
    ny, nx = a.shape[1] - 2, a.shape[2] - 2
    ks = np.clip(np.arange(1, nz + 2) + dk, 0, nz + 1)
    return a[ks][:, 1 + dj:1 + dj + ny, 1 + di:1 + di + nx]


MASKS = ("u", "v", "w", "c", "uw", "vw", "uv")


def createmasks(nx, ny, nz, lists=None, wrapx=False, wrapy=False):
    """IIu, IIv, IIw, IIc, IIuw, IIvw, IIuv on levels kb..ke+kh [nz+1, ny, nx] and their level counts II*s
    (src/modibm.f90:2120-2206).  `lists`: {grid: (solid[n,3], ...)} global 1-based i, j, k, or None without libm.
    Ghost cells of the point masks are fluid unless the direction wraps (a direction the reference run splits over ranks)."""
    one = np.ones((nz + 1, ny, nx), dtype=np.int64)
    if lists is None:
        return {m: one.copy() for m in MASKS}, {m: np.full(nz + 1, nx * ny, dtype=np.int64) for m in MASKS}
    pt = {}
    for q in "uvwc":
        a = np.ones((nz + 2, ny + 2, nx + 2), dtype=np.int64)          # index == i, j, k of the reference (ghost ring 0 / n+1)
        if q in lists:
            for i, j, k in lists[q][0]:
                a[k, j, i] = 0
        if wrapx:
            a[:, :, 0], a[:, :, -1] = a[:, :, -2], a[:, :, 1]
        if wrapy:
            a[:, 0], a[:, -1] = a[:, -2], a[:, 1]
        pt[q] = a
    pt["w"][1] = 0                                                      # IIw(:, :, kb) = 0
    I = lambda a, di=0, dj=0, dk=0: a[1 + dk:nz + 2 + dk, 1 + dj:ny + 1 + dj, 1 + di:nx + 1 + di]     # noqa: E731
    II = {q: I(pt[q]).copy() for q in "uvwc"}
    II["uv"], II["uw"], II["vw"] = one.copy(), one.copy(), one.copy()
    II["uv"][:nz] = (I(pt["u"]) * I(pt["u"], dj=-1) * I(pt["v"]) * I(pt["v"], di=-1))[:nz]
    II["uw"][1:nz] = (I(pt["u"]) * I(pt["u"], dk=-1) * I(pt["w"]) * I(pt["w"], di=-1))[1:nz]
    II["vw"][1:nz] = (I(pt["v"]) * I(pt["v"], dk=-1) * I(pt["w"]) * I(pt["w"], dj=-1))[1:nz]
    II["uw"][0] = 0
    II["vw"][0] = 0
    return II, {m: II[m].sum(axis=(1, 2)) for m in MASKS}


def avexy_ibm(var, II, IIs, lnan=False):
    """src/modmpi.f90:623-664: masked slab average; a level without fluid points gives -999, except the first one when
    lnan is false, which takes the unmasked sum over the count of the last level."""
    s = (var * II).sum(axis=(1, 2))
    cnt = IIs.copy()
    if not lnan and cnt[0] == 0:
        s[0] = var[0].sum()
        cnt[0] = cnt[-2]                                                # IId(kb) = IId(ke)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(cnt == 0, -999., s / np.where(cnt == 0, 1, cnt))


class TDumpOracle:
    def __init__(self, g, nsv=0, ltempeq=False, lmoist=False, lists=None, wrapx=False, wrapy=False):
        self.g, self.nsv, self.ltempeq, self.lmoist = g, nsv, ltempeq, lmoist
        self.acc = {}
        self.II, self.IIs = createmasks(g.nx, g.ny, g.nz, lists, wrapx, wrapy)
        self.prof = {}
        self.yprof = {}

    def _updy(self, name, var, mask, ts, T):
        """avey_ibm (src/modmpi.f90: masked sum over y / column count, -999 where a column has no fluid point) + running average"""
        nz = self.g.nz
        II = self.II[mask][:nz]
        cnt = II.sum(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            ya = np.where(cnt == 0, -999., (var[:nz] * II).sum(axis=1) / np.where(cnt == 0, 1, cnt))
        old = self.yprof.get(name, np.zeros_like(ya))
        self.yprof[name] = (old * (T - ts) + ya * ts) * (1. / T)

    def _avey(self, var, mask):
        nz = self.g.nz
        II = self.II[mask][:nz]
        cnt = II.sum(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(cnt == 0, -999., (var[:nz] * II).sum(axis=1) / np.where(cnt == 0, 1, cnt))

    def _updp(self, name, var, mask, ts, T):
        xy = avexy_ibm(var, self.II[mask], self.IIs[mask])
        old = self.prof.get(name, np.zeros_like(xy))
        self.prof[name] = (old * (T - ts) + xy * ts) * (1. / T)

    def _upd(self, name, sample, ts, T):
        old = self.acc.get(name, np.zeros_like(sample))
        self.acc[name] = (old * (T - ts) + sample * ts) * (1. / T)

    def sample(self, st, ts, T):
        """st: um, vm, wm, pres0, ekh (m-arrays), thlm / qtm (m-arrays), svm_n (c-arrays)."""
        g = self.g
        nz = g.nz
        dzf, dzh = g.dzf, g.dzh
        k = np.arange(1, nz + 2)
        dzhi = (1. / dzh[k])[:, None, None]
        dzf_k, dzf_km = dzf[np.minimum(k, nz + 1)][:, None, None], dzf[k - 1][:, None, None]
        um, vm, wm = st["um"], st["vm"], st["wm"]
        u, v, w = _m(um), _m(vm), _m(wm)
        dx, dxi = g.dx, 1. / g.dx
        uik = 0.5 * dzhi * (u * dzf_km + _m(um, dk=-1) * dzf_k)
        wik = 0.5 * dxi * (w * dx + _m(wm, di=-1) * dx)
        vjk = 0.5 * dzhi * (v * dzf_km + _m(vm, dk=-1) * dzf_k)
        wjk = 0.5 * (w + _m(wm, dj=-1))
        uij = 0.5 * (u + _m(um, dj=-1))
        vij = 0.5 * dxi * (v * dx + _m(vm, di=-1) * dx)
        uc = 0.5 * (_m(um, di=1) + u)
        vc = 0.5 * (_m(vm, dj=1) + v)
        wc = 0.5 * (_m(wm, dk=1) + w)
        wc[-1] = wc[-2]                                      # :827-828
        for name, s in (("uwtik", wik * uik), ("vwtjk", wjk * vjk), ("uvtij", uij * vij), ("uutc", uc * uc), ("vvtc", vc * vc),
                        ("wwtc", wc * wc), ("utik", uik), ("wtik", wik), ("vtjk", vjk), ("wtjk", wjk), ("utij", uij), ("vtij", vij),
                        ("umt", u), ("vmt", v), ("wmt", w), ("utc", uc), ("vtc", vc), ("wtc", wc), ("pt", _m(st["pres0"]))):
            self._upd(name, s, ts, T)
        ekh = st["ekh"]
        dzh2i = (1. / dzh[k] ** 2)[:, None, None]
        # SGS momentum fluxes at the uw / vw edges (:834-846) and the slab averages of xytdump (:1041-1054, 1088-1099)
        if "ekm" in st:
            ekm, dyi, dzhiq = st["ekm"], 1. / g.dy, 0.25 * dzhi
            emom = (dzf_km * (_m(ekm) * dx + _m(ekm, di=-1) * dx) + dzf_k * (_m(ekm, dk=-1) * dx + _m(ekm, di=-1, dk=-1) * dx)) * dxi * dzhiq
            usgs = emom * ((u - _m(um, dk=-1)) * dzhi + (w - _m(wm, di=-1)) * dxi)
            emom = (dzf_km * (_m(ekm) + _m(ekm, dj=-1)) + dzf_k * (_m(ekm, dk=-1) + _m(ekm, dj=-1, dk=-1))) * dzhiq
            vsgs = emom * ((v - _m(vm, dk=-1)) * dzhi + (w - _m(wm, dj=-1)) * dyi)
            # ytdump's sample (:964-999): y-averages of the same quantities (+ wsgs, :851-855), levels kb..ke
            dzfi = (1. / dzf[np.minimum(k, nz + 1)])[:, None, None]
            dzfi_m = (1. / dzf[k - 1])[:, None, None]
            wsgs = (_m(ekm) * (_m(wm, dk=1) - w) * dzfi - _m(ekm, dk=-1) * (w - _m(wm, dk=-1)) * dzfi_m) * 2. * dzhi
            self._updy("uyt", u, "u", ts, T); self._updy("vyt", v, "v", ts, T); self._updy("wyt", w, "w", ts, T)
            self._updy("usgsyt", usgs, "uw", ts, T); self._updy("wsgsyt", wsgs, "w", ts, T)
            if self.ltempeq:
                th_, thm_ = _m(st["thlm"]), _m(st["thlm"], dk=-1)
                self._updy("thlyt", th_, "c", ts, T)
                self._updy("thlsgsyt", 0.5 * (dzf_km * _m(ekh) + dzf_k * _m(ekh, dk=-1)) * (th_ - thm_) * dzh2i, "w", ts, T)
            for n in range(min(self.nsv, 3)):
                a = st[f"svm_{n}"][1:-1, 1:-1, 1:-1]
                p0, pm = _m(a), _m(a, dk=-1)
                self._updy(f"sca{n + 1}yt", p0, "c", ts, T)
                self._updy(f"sv{n + 1}sgsyt", 0.5 * (dzf_km * _m(ekh) + dzf_k * _m(ekh, dk=-1)) * (p0 - pm) * dzh2i, "w", ts, T)
            self._updp("uxyt", u, "u", ts, T); self._updp("vxyt", v, "v", ts, T); self._updp("wxyt", w, "w", ts, T)
            self._updp("pxyt", _m(st["pres0"]), "c", ts, T)
            self._updp("usgsxyt", usgs, "uw", ts, T); self._updp("vsgsxyt", vsgs, "vw", ts, T)
            if self.ltempeq:
                th, thm = _m(st["thlm"]), _m(st["thlm"], dk=-1)
                thlsgs = 0.5 * (dzf_km * _m(ekh) + dzf_k * _m(ekh, dk=-1)) * (th - thm) * dzh2i
                thlsgs[-1] = 0.                              # ke+kh: never set in the reference
                self._updp("thlxyt", th, "c", ts, T); self._updp("thlsgsxyt", thlsgs, "w", ts, T)
            if self.lmoist:
                self._updp("qtxyt", _m(st["qtm"]), "c", ts, T)

        def scalar(tag, a, with_sgs):
            p0, pm = _m(a), _m(a, dk=-1)
            pk = 0.5 * dzhi * (p0 * dzf_km + pm * dzf_k)
            self._upd(tag + "t", p0, ts, T)
            self._upd(tag + "tk", pk, ts, T)
            self._upd("w" + tag + "tk", w * pk, ts, T)
            self._upd(tag + tag + "t", p0 * p0, ts, T)
            if with_sgs:
                sg = 0.5 * (dzf_km * _m(ekh) + dzf_k * _m(ekh, dk=-1)) * (p0 - pm) * dzh2i
                sg[-1] = 0.                                  # ke+kh: never set in the reference
                self._upd(tag + "sgst", sg, ts, T)
        if self.ltempeq:
            scalar("thl", st["thlm"], False)
        if self.lmoist:
            scalar("qt", st["qtm"], False)
        for n in range(min(self.nsv, 4)):
            scalar(f"sv{n + 1}", st[f"svm_{n}"][1:-1, 1:-1, 1:-1], True)

    def output(self):
        """The tdump variables (names of initstatsdump :286-320), levels kb..ke."""
        a = self.acc
        o = {"ut": a["umt"], "vt": a["vmt"], "wt": a["wmt"], "pt": a["pt"],
             "upwpt": a["uwtik"] - a["utik"] * a["wtik"], "vpwpt": a["vwtjk"] - a["vtjk"] * a["wtjk"],
             "upvpt": a["uvtij"] - a["utij"] * a["vtij"],
             "upuptc": a["uutc"] - a["utc"] * a["utc"], "vpvptc": a["vvtc"] - a["vtc"] * a["vtc"], "wpwptc": a["wwtc"] - a["wtc"] * a["wtc"]}
        o["tketc"] = 0.5 * (o["upuptc"] + o["vpvptc"] + o["wpwptc"])
        if self.ltempeq:
            o["thlt"] = a["thlt"]
            o["wpthlpt"] = a["wthltk"] - a["wmt"] * a["thltk"]
            o["thlpthlpt"] = a["thlthlt"] - a["thlt"] * a["thlt"]
        if self.lmoist:
            o["qtt"] = a["qtt"]
        for n in range(1, min(self.nsv, 4) + 1):
            o[f"sca{n}t"] = a[f"sv{n}t"]
            o[f"wpsca{n}pt"] = a[f"wsv{n}tk"] - a["wmt"] * a[f"sv{n}tk"]
            o[f"sca{n}psca{n}pt"] = a[f"sv{n}sv{n}t"] - a[f"sv{n}t"] * a[f"sv{n}t"]
            o[f"sv{n}sgs"] = a[f"sv{n}sgst"]
        return {k: v[:-1] for k, v in o.items()}

    def xyt(self):
        """xytdump's table (:1404-1460) on levels kb..ke."""
        a, p = self.acc, self.prof
        av = lambda v, m: avexy_ibm(v, self.II[m], self.IIs[m])      # noqa: E731
        up2, vp2, wp2 = a["uutc"] - a["utc"] * a["utc"], a["vvtc"] - a["vtc"] * a["vtc"], a["wwtc"] - a["wtc"] * a["wtc"]
        o = {"uxyt": p["uxyt"], "vxyt": p["vxyt"], "wxyt": p["wxyt"], "pxyt": p["pxyt"], "usgsxyt": p["usgsxyt"], "vsgsxyt": p["vsgsxyt"],
             "uwtxyik": av(a["utik"] * a["wtik"], "uw"), "vwtxyjk": av(a["vtjk"] * a["wtjk"], "vw"), "wwtxyk": av(a["wmt"] * a["wmt"], "w"),
             "uvtxyij": av(a["utij"] * a["vtij"], "uv"), "upwptxyik": av(a["uwtik"] - a["utik"] * a["wtik"], "uw"),
             "vpwptxyjk": av(a["vwtjk"] - a["vtjk"] * a["wtjk"], "vw"), "upvptxyij": av(a["uvtij"] - a["utij"] * a["vtij"], "uv"),
             "upuptxyc": av(up2, "c"), "vpvptxyc": av(vp2, "c"), "wpwptxyc": av(wp2, "c"), "tketxyc": av(0.5 * (wp2 + vp2 + up2), "c")}
        if self.ltempeq:
            o.update({"thlxyt": p["thlxyt"], "thlsgsxyt": p["thlsgsxyt"], "wthltxyk": av(a["wmt"] * a["thltk"], "w"),
                      "wpthlptxyk": av(a["wthltk"] - a["wmt"] * a["thltk"], "w"), "thlpthlptxy": av(a["thlthlt"] - a["thlt"] * a["thlt"], "c")})
        if self.lmoist:
            o["qtxyt"] = p["qtxyt"]
        return {k: v[:-1] for k, v in o.items()}

    def yt(self):
        """ytdump's table (:1104-1132 running y-averages, :1471-1507 y-averages of the time-averaged fields), [nz, nx]."""
        a = self.acc
        o = dict(self.yprof)
        o.update({"upwptyik": self._avey(a["uwtik"] - a["utik"] * a["wtik"], "uw"), "uwtyik": self._avey(a["utik"] * a["wtik"], "uw"),
                  "upuptyc": self._avey(a["uutc"] - a["utc"] * a["utc"], "c"), "wpwptyc": self._avey(a["wwtc"] - a["wtc"] * a["wtc"], "c")})
        if self.ltempeq:
            o.update({"wpthlptyk": self._avey(a["wthltk"] - a["wmt"] * a["thltk"], "w"), "wthltyk": self._avey(a["wmt"] * a["thltk"], "w"),
                      "thlpthlpty": self._avey(a["thlthlt"] - a["thlt"] * a["thlt"], "c")})
        for n in range(1, min(self.nsv, 3) + 1):
            o.update({f"wpsv{n}ptyk": self._avey(a[f"wsv{n}tk"] - a["wmt"] * a[f"sv{n}tk"], "w"), f"wsv{n}tyk": self._avey(a["wmt"] * a[f"sv{n}tk"], "w"),
                      f"sv{n}psv{n}pty": self._avey(a[f"sv{n}sv{n}t"] - a[f"sv{n}t"] * a[f"sv{n}t"], "c")})
        return o
