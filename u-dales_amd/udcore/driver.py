"""Inflow from the planes of a precursor run (&BC BCxm = 3, &DRIVER idriver = 2) for the Python runner: the host side the Fortran drop-ins
leave to the reference's own moddriver.  readdriverfile (src/moddriver.f90:752-932): `tdriver_000.NNN` holds `driverstore` time stamps,
`u / v / w / h / q / sdriver_RRR.NNN` one plane (jb-jh : je+jh, kb-kh : ke+kh), j fastest, per stamp (direct access, no record marks; the
scalars' planes carry jhc / khc ghost cells and all nsv fields per record).  drivergen (:213-376, the branch without lchunkread): the record
nearest to `timee`; within 1e-4 of it the record itself, else the linear interpolation with the record on the other side; the m planes are
the 0 planes (both are refreshed on the same calls: the start-up and RK stage 3)."""
import os

import numpy as np


class DriverInlet:
    def __init__(self, wdir, jobnr, ny, nz, nstore, rank=0, thl=False, qt=False, nsv=0):
        self.ny, self.nz, self.n = ny, nz, int(nstore)
        nm = (nz + 2) * (ny + 2)

        def rd(q, per, who=rank):
            fn = os.path.join(wdir, f"{q}driver_{who:03d}.{jobnr:03d}")
            a = np.fromfile(fn, dtype="<f8")
            if a.size < self.n * per:
                raise ValueError(f"{fn}: {a.size // per} records, &DRIVER driverstore asks for {self.n}")
            return a[:self.n * per]
        self.t = rd("t", 1, 0)
        self.planes0 = {q: rd(q, nm).reshape(self.n, nz + 2, ny + 2) for q in "uvw"}
        if thl:
            self.planes0["h"] = rd("h", nm).reshape(self.n, nz + 2, ny + 2)
        if qt:
            self.planes0["q"] = rd("q", nm).reshape(self.n, nz + 2, ny + 2)
        if nsv:      # (jb-jhc : je+jhc, kb-khc : ke+khc, 1 : nsv) per record, jhc = khc = 2
            self.planes0["s"] = rd("s", (nz + 4) * (ny + 4) * nsv).reshape(self.n, nsv, nz + 4, ny + 4)
        self.last = None

    def at(self, timee):
        """-> {'u': plane, ...} as drivergen leaves u0driver ... at this time"""
        if timee > self.t.max():
            raise RuntimeError("Time in simulation has exceeded the inlet information - no more inlet data available!")      # :233-239
        x = int(np.argmin(np.abs(self.t - timee)))
        el = self.t[x] - timee
        out = {}
        for q, st in self.planes0.items():
            if abs(el) < 1e-4 or (el > 0. and x == 0):      # (ahead of the first record: that record, :278-297)
                out[q] = st[x].copy()
            elif el < 0.:
                w = (timee - self.t[x]) / (self.t[x + 1] - self.t[x])
                out[q] = st[x] + (st[x + 1] - st[x]) * w
            else:
                w = (timee - self.t[x - 1]) / (self.t[x] - self.t[x - 1])
                out[q] = st[x - 1] + (st[x] - st[x - 1]) * w
        self.last = out
        return out

    def hand_over(self, core, timee):
        """What the reference's `boundary` does with drivergen's planes on the start-up and on RK stage 3 (src/modboundary.f90:262-266):
        they go to the device, which applies them with its next `boundary` (udc_set_open_x_inlet[_scalar])."""
        from . import lib as L
        p = self.at(timee)
        core.set_open_x_inlet(p["u"], p["v"], p["w"])
        if "h" in p:
            core.set_open_x_inlet_scalar(L.THL0, p["h"])
        if "q" in p:
            core.set_open_x_inlet_scalar(L.QT0, p["q"])
        if "s" in p:
            for n in range(p["s"].shape[0]):
                core.set_open_x_inlet_scalar(L.scalar_field(L.SV0, n), p["s"][n], halo=2)
