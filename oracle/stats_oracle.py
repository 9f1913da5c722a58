"""TEST INFRASTRUCTURE -- numpy restatement of the tdump statistics of the reference, NOT the product.

PARITY UNPINNED: src/modstatsdump.f90 cannot be compiled here (it writes NetCDF and its arithmetic is interleaved with
those calls), and the reference holds no golden vectors for it, so this restatement is checked only against the lines it
follows and against properties (tests/test_gpu_stats.py).  Lines followed: sampling :812-832, :858-927; running
averages :1137-1213; output variables :1557-1645.

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


class TDumpOracle:
    def __init__(self, g, nsv=0, ltempeq=False, lmoist=False):
        self.g, self.nsv, self.ltempeq, self.lmoist = g, nsv, ltempeq, lmoist
        self.acc = {}

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
