"""Host-side mirror of the reference's dynamical-core call surface, in Python.

`DynCore` exposes the procedures program.f90 calls (src/program.f90:134-207) with the same
names and order -- tstep_update, advection, subgrid, forces, poisson, tstep_integrate, halos,
boundary -- each a thin call into the C ABI (include/udcore.h).  The Fortran drop-in modules
in u-dales_amd/fortran/ do the same thing from the reference's own driver.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import lib as L
from .grid import Grid


def tke_constants(cf=2.5, cn=0.76, Rigc=0.25, Prandtl=0.333, ch1=1., alpha_kolm=1.5):
    """cm, ch2, ce1, ce2 of initsubgrid (src/modsubgrid.f90:63-71)."""
    pi = 3.141592653589793116
    cm = cf / (2. * pi) * (1.5 * alpha_kolm) ** (-1.5)
    ch2 = Prandtl - ch1
    ceps = 2. * pi / cf * (1.5 * alpha_kolm) ** (-1.5)
    ce1 = (cn ** 2) * (cm / Rigc - ch1 * cm)
    return dict(cm=cm, cn=cn, ch1=ch1, ch2=ch2, ce1=ce1, ce2=ceps - ce1)


class DynCore:
    def __init__(self, g: Grid, sgs=L.SGS_VREMAN, bctopm=1, nsv=0, numol=1.5e-5, prandtlmol=0.71,
                 prandtli=1. / 0.333, c_vreman=0.07, csz=0.21658244510412, uinf=0., vinf=0.,
                 device=0, rank=0, nranks=1, lbottom=False, z0=-1., open_x=None):
        # open_x = (uprof, vprof), each [ktot+2] by the reference's k: inflow / outflow in x (&BC BCxm = 2, udc_create_open_x)
        self.g = g
        self.nsv = nsv
        self.lib = L.load()
        self._dzf = np.ascontiguousarray(g.dzf, dtype=np.float64)
        self._dzh = np.ascontiguousarray(g.dzh, dtype=np.float64)
        cfg = L.UdcConfig(g.nx, g.ny, g.nz, nranks, rank, device, g.dx, g.dy,
                          self._dzf.ctypes.data_as(L.DP), self._dzh.ctypes.data_as(L.DP),
                          numol, 1. / prandtlmol, prandtli, c_vreman, csz, sgs, bctopm, uinf, vinf, nsv,
                          int(bool(lbottom)), z0)
        self.h = C.c_void_p()
        self.open_x = open_x is not None
        if self.open_x:
            up = np.ascontiguousarray(open_x[0], dtype=np.float64)
            vp = np.ascontiguousarray(open_x[1], dtype=np.float64)
            assert up.size == g.nz + 2 and vp.size == g.nz + 2
            L._check(self.lib.udc_create_open_x(C.byref(cfg), up.ctypes.data_as(L.DP), vp.ctypes.data_as(L.DP), C.byref(self.h)),
                     "udc_create_open_x")
        else:
            L._check(self.lib.udc_create(C.byref(cfg), C.byref(self.h)), "udc_create")
        self.nyl = g.ny // nranks
        self.rank, self.nranks = rank, nranks
        self.rk3step = 0
        self.dt = 0.
        self.ltempeq = False
        self.lmoist = False
        self.moist_thermo, self._thermo_started = False, False
        self.loneeqn = False
        self.timee = 0.

    # ---- lifetime
    def close(self):
        if self.h:
            self.lib.udc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_init(self, unique_id: bytes):
        """RCCL communicator over the y-slab ranks (id from udc_comm_unique_id on rank 0)."""
        shm = os.environ.get("UDC_TEST_SHM", "")
        if shm and getattr(self.lib, "udc_comm_init_shm", None) is not None:
            # the test library's inter-process transport (libudcore_test.so via UDC_LIBPATH): ranks sharing one device in tests
            L._check(self.lib.udc_comm_init_shm(self.h, shm.encode()), "udc_comm_init_shm")
            return
        buf = (C.c_ubyte * 128)(*unique_id)
        L._check(self.lib.udc_comm_init(self.h, buf), "udc_comm_init")

    def comm_info(self) -> dict:
        """What the communicator says about itself (udc_comm_info): transport, RCCL's own rank count / rank / device / version."""
        info = (C.c_int * 8)()
        L._check(self.lib.udc_comm_info(self.h, info), "udc_comm_info")
        v = info[4]
        return {"transport": ("none", "rccl", "test: in-process group", "test: shared memory")[info[0]],
                "nranks": info[1], "rank": info[2], "device": info[3],
                "version": f"{v // 10000}.{v // 100 % 100}.{v % 100}", "transpose_k_chunks": info[5],
                "handle_nranks": info[6], "handle_rank": info[7]}

    def comm_stats(self, mode: int = 0):
        """Exchange bookkeeping (udc_comm_stats): mode 1 resets and starts timing, 0 reads, 2 reads and stops."""
        out = (C.c_double * 16)()
        L._check(self.lib.udc_comm_stats(self.h, mode, out), "udc_comm_stats")
        if mode == 1:
            return None
        keys = ("alltoall_ops", "alltoall_bytes_per_peer", "alltoall_bytes_sent", "alltoall_ms", "ghost_row_exchanges",
                "ghost_row_bytes_to_prev", "ghost_row_bytes_to_next", "ghost_row_ms", "allreduce_ops", "allreduce_doubles")
        return {k: out[i] for i, k in enumerate(keys)}

    def comm_dry_run(self, on: bool):
        """Exchanges skipped (timing only; the state is wrong from there on)."""
        L._check(self.lib.udc_comm_dry_run(self.h, 1 if on else 0), "udc_comm_dry_run")

    def last_plan(self) -> dict:
        """The order the last fused substep ran in (udc_last_plan)."""
        o = (C.c_int * 16)()
        L._check(self.lib.udc_last_plan(self.h, o), "udc_last_plan")
        row = ("folded", "beside a sweep", "in line", "with the pipelined sweep", "inside the backward transpose")
        return {"ghost_rows_folded": bool(o[0]), "closure": ("folded", "edge rows first, ekm rows beside the interior", "plain")[o[1]],
                "ekh_written": bool(o[2]), "momentum_sweep_pipelined_with_solve": bool(o[3]), "divergence_in_x_transform": bool(o[4]),
                "vp_ghost_row": row[o[5]], "p_ghost_row": row[o[6]],
                "integration": ("one launch", "edge rows first, velocity rows beside the interior")[o[7]],
                "slab_layout": bool(o[11]), "fused_line_transforms": bool(o[12]), "transpose_k_chunks": o[13],
                "own_forward_half": bool(o[14]), "pressure_total_form": bool(o[15] & 1), "scalars_updated_in_their_sweep": int(o[15] >> 1)}

    def comm_init_local(self, group: int):
        """Attach to an in-process group of virtual ranks (udc_local_group_create); test transport."""
        L._check(self.lib.udc_comm_init_local(self.h, group), "udc_comm_init_local")

    # ---- residency
    def _bounds(self, arr):
        nz, ny, nx = arr.shape
        hx = (nx - self.g.nx) // 2
        hy = (ny - self.nyl) // 2
        hz = (nz - self.g.nz) // 2
        lb = (C.c_int * 3)(1 - hx, 1 - hy, 1 - hz)
        ub = (C.c_int * 3)(self.g.nx + hx, self.nyl + hy, self.g.nz + hz)
        return lb, ub

    def upload(self, field, arr):
        """arr: [k, j, i] float64 with symmetric halos (any width) around the local interior."""
        fid = L.FIELD_IDS[field] if isinstance(field, str) else field
        a = np.ascontiguousarray(arr, dtype=np.float64)
        lb, ub = self._bounds(a)
        L._check(self.lib.udc_field_upload(self.h, fid, a.ctypes.data_as(L.DP), lb, ub), f"upload {field}")

    def download(self, field, halo=1):
        fid = L.FIELD_IDS[field] if isinstance(field, str) else field
        a = np.zeros((self.g.nz + 2 * halo, self.nyl + 2 * halo, self.g.nx + 2 * halo))
        lb, ub = self._bounds(a)
        L._check(self.lib.udc_field_download(self.h, fid, a.ctypes.data_as(L.DP), lb, ub), f"download {field}")
        return a

    def set_forcing(self, dpdxl, dpdyl):
        a = np.ascontiguousarray(dpdxl, dtype=np.float64)
        b = np.ascontiguousarray(dpdyl, dtype=np.float64)
        self.dpdxl, self.dpdyl = a.copy(), b.copy()          # host copies (the dp/dx ODE of ifixuinf = 2 advances them)
        L._check(self.lib.udc_set_forcing(self.h, a.ctypes.data_as(L.DP), b.ctypes.data_as(L.DP), len(a)),
                 "udc_set_forcing")

    def load_state(self, st: dict):
        """st: output of grid.cold_start() or any dict of named arrays."""
        for name, arr in st.items():
            if name in L.FIELD_IDS:
                self.upload(name, arr)
            elif name.startswith("sv0_"):
                self.upload(L.scalar_field(L.SV0, int(name[4:])), arr)
            elif name.startswith("svm_"):
                self.upload(L.scalar_field(L.SVM, int(name[4:])), arr)
            elif name.startswith("svp_"):
                self.upload(L.scalar_field(L.SVP, int(name[4:])), arr)

    # ---- the reference's call surface
    def tstep_update(self, dtmax, ladaptive=False, courant=1.5, diffnr=0.25):
        """src/modtstep.f90:49-154 (normal time loop branch)."""
        self.rk3step = self.rk3step % 3 + 1
        if self.rk3step == 1:
            if ladaptive:
                c, d = C.c_double(), C.c_double()
                L._check(self.lib.udc_tstep_maxima(self.h, C.c_double(self.dt), C.byref(c), C.byref(d)),
                         "udc_tstep_maxima")
                self.dt = min(dtmax, self.dt * courant / c.value, self.dt * diffnr / d.value)
            else:
                self.dt = dtmax
            self.timee += self.dt
        return self.rk3step, self.dt

    def advection(self):
        L._check(self.lib.udc_advection(self.h), "udc_advection")

    def subgrid(self):
        L._check(self.lib.udc_subgrid(self.h), "udc_subgrid")

    def bottom(self):
        """`bottom` (src/modibm.f90:1998): floor wall function, between subgrid and forces."""
        L._check(self.lib.udc_bottom(self.h), "udc_bottom")

    def forces(self):
        self._ensure_thermo()
        L._check(self.lib.udc_forces(self.h), "udc_forces")

    def set_masscorr_outflow(self, luoutflowr=True, uflowrate=1.):
        """&PHYSICS luoutflowr: include/udcore.h udc_set_masscorr_outflow."""
        L._check(self.lib.udc_set_masscorr_outflow(self.h, int(bool(luoutflowr)), C.c_double(uflowrate)), "udc_set_masscorr_outflow")

    def set_masscorr(self, luvolflowr=False, uflowrate=1., lvvolflowr=False, vflowrate=1.):
        """&PHYSICS luvolflowr/uflowrate, lvvolflowr/vflowrate (src/modglobal.f90:231,331)."""
        L._check(self.lib.udc_set_masscorr(self.h, int(bool(luvolflowr)), C.c_double(uflowrate),
                                           int(bool(lvvolflowr)), C.c_double(vflowrate)), "udc_set_masscorr")

    def set_tempeq(self, iadv_thl=2, bctopt=1, wttop=0., thl_top=-1., bcbott=1, wtsurf=0., thlpcar=None):
        """&PHYSICS ltempeq: thl becomes a transported (passive) field, see include/udcore.h udc_set_tempeq."""
        L._check(self.lib.udc_set_tempeq(self.h, int(iadv_thl), int(bctopt), C.c_double(wttop), C.c_double(thl_top),
                                         int(bcbott), C.c_double(wtsurf)), "udc_set_tempeq")
        self.ltempeq = True
        if thlpcar is not None:
            a = np.ascontiguousarray(thlpcar, dtype=np.float64)
            L._check(self.lib.udc_set_thl_source(self.h, a.ctypes.data_as(L.DP), len(a)), "udc_set_thl_source")

    def set_chem(self, lchem=True, k1=0., jno2=0.):
        """&CHEMISTRY lchem, k1, JNO2: NO - NO2 - O3 chemistry on scalars 1-3 (include/udcore.h udc_set_chem)."""
        L._check(self.lib.udc_set_chem(self.h, int(bool(lchem)), C.c_double(k1), C.c_double(jno2)), "udc_set_chem")

    def set_shifted_pbc(self, a, sinx, u0av):
        """shiftedPBCs (&BC ds): vs = a u0av(k) sinx(i), see include/udcore.h udc_set_shifted_pbc."""
        sx = np.ascontiguousarray(sinx, dtype=np.float64)
        ua = np.ascontiguousarray(u0av, dtype=np.float64)
        L._check(self.lib.udc_set_shifted_pbc(self.h, C.c_double(a), sx.ctypes.data_as(L.DP), len(sx), ua.ctypes.data_as(L.DP), len(ua)),
                 "udc_set_shifted_pbc")

    def shifted_pbcs(self):
        L._check(self.lib.udc_shifted_pbcs(self.h), "udc_shifted_pbcs")

    def set_scalar_top(self, n, bctops=1, value=0.):
        """&BC BCtops for scalar n: 1 = flux wsvtop(n), 2 = value sv_top(n) (include/udcore.h udc_set_scalar_top)."""
        L._check(self.lib.udc_set_scalar_top(self.h, int(n), int(bctops), C.c_double(value)), "udc_set_scalar_top")

    def set_scalar_source(self, n, src):
        """Constant source of scalar n (scalsource): src is [nz, nyl, nx] over the interior of this slab, or None to remove
        it.  Only the smallest box holding its non-zeros is kept on the device."""
        if src is None:
            L._check(self.lib.udc_set_scalar_source(self.h, int(n), None, (C.c_int * 3)(1, 1, 1), (C.c_int * 3)(0, 0, 0)),
                     "udc_set_scalar_source")
            return
        a = np.asarray(src, dtype=np.float64)
        assert a.shape == (self.g.nz, self.nyl, self.g.nx), a.shape
        nzr = np.nonzero(a)
        if len(nzr[0]) == 0:
            return self.set_scalar_source(n, None)
        lo = [int(x.min()) for x in nzr]
        hi = [int(x.max()) for x in nzr]
        box = np.ascontiguousarray(a[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1])
        lb = (C.c_int * 3)(lo[2] + 1, lo[1] + 1, lo[0] + 1)          # (i, j, k), the reference's local indices
        ub = (C.c_int * 3)(hi[2] + 1, hi[1] + 1, hi[0] + 1)
        L._check(self.lib.udc_set_scalar_source(self.h, int(n), box.ctypes.data_as(L.DP), lb, ub), "udc_set_scalar_source")

    def scalsource(self):
        L._check(self.lib.udc_scalsource(self.h), "udc_scalsource")

    def set_scalar_bcx(self, bcxs, svprof, uouttot):
        """&BC BCxs (include/udcore.h): svprof[nsv, ktot+2] indexed by the reference's k."""
        p = np.ascontiguousarray(svprof, dtype=np.float64)
        L._check(self.lib.udc_set_scalar_bcx(self.h, int(bcxs), p.ctypes.data_as(L.DP), C.c_double(uouttot)), "udc_set_scalar_bcx")

    def set_open_x_outflow(self, wlev=None, uouttot=0., hold_first=False):
        """The outlet's speed under BCxm = 2: a constant, or (wlev [ktot]) the weighted mean of u's slab averages."""
        w = None if wlev is None else np.ascontiguousarray(wlev, dtype=np.float64)
        self._open_x_avg, self._uouttot = w is not None, float(uouttot)
        L._check(self.lib.udc_set_open_x_outflow(self.h, None if w is None else w.ctypes.data_as(L.DP), C.c_double(uouttot),
                                                     1 if hold_first else 0),
                 "udc_set_open_x_outflow")

    def set_open_x_sample_gap(self, on=True):
        """One-shot: the next fused stage-3 substep ends ahead of `boundary` (the state the reference's dumps see, src/program.f90:199-214);
        boundary() (+ thermodynamics()) must follow the download."""
        L._check(self.lib.udc_set_open_x_sample_gap(self.h, 1 if on else 0), "udc_set_open_x_sample_gap")

    def set_scalar_bcx_outflow(self, wlev):
        w = np.ascontiguousarray(wlev, dtype=np.float64)
        L._check(self.lib.udc_set_scalar_bcx_outflow(self.h, w.ctypes.data_as(L.DP)), "udc_set_scalar_bcx_outflow")

    def set_ibm_wallfun(self, iwallmom, prandtlturb, zf, zh):
        """Facet wall functions for momentum (include/udcore.h): zf, zh = levels 1..ktot+1."""
        zf, zh = np.ascontiguousarray(zf, dtype=np.float64), np.ascontiguousarray(zh, dtype=np.float64)
        L._check(self.lib.udc_set_ibm_wallfun(self.h, int(iwallmom), C.c_double(prandtlturb), zf.ctypes.data_as(L.DP), zh.ctypes.data_as(L.DP)),
                 "udc_set_ibm_wallfun")

    def set_ibm_wallheat(self, iwalltemp):
        L._check(self.lib.udc_set_ibm_wallheat(self.h, int(iwalltemp)), "udc_set_ibm_wallheat")

    def set_ibm_sections(self, grid, S, facets, tmask):
        """S: udcore.facets.wall_sections table; facets: read_facets; tmask[n, 2]."""
        ip = C.POINTER(C.c_int)
        f = S["fac"] - 1
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)      # noqa: E731
        cell, comp, rid = a(S["cell"], np.int32), a(S["comprec"], np.int32), a(S["recids"], np.int32)
        area, dist, recpt, tm = a(S["area"], np.float64), a(S["dist"], np.float64), a(S["recpt"], np.float64), a(tmask, np.float64)
        norm, z0, z0h, ts = a(facets["norm"][f], np.float64), a(facets["z0"][f], np.float64), a(facets["z0h"][f], np.float64), a(facets["tsurf"][f], np.float64)
        L._check(self.lib.udc_set_ibm_sections(self.h, int(grid), int(S["n"]), cell.ctypes.data_as(ip), area.ctypes.data_as(L.DP),
                                               dist.ctypes.data_as(L.DP), norm.ctypes.data_as(L.DP), z0.ctypes.data_as(L.DP),
                                               z0h.ctypes.data_as(L.DP), ts.ctypes.data_as(L.DP), comp.ctypes.data_as(ip),
                                               recpt.ctypes.data_as(L.DP), rid.ctypes.data_as(ip), tm.ctypes.data_as(L.DP)), "udc_set_ibm_sections")

    def set_ibm_wallmoist(self, iwallmoist, lgr, qwall, hurel=None, resc=None, ress=None):
        """Latent part of wallfunheat per c-grid section (include/udcore.h udc_set_ibm_wallmoist)."""
        lg = np.ascontiguousarray(lgr, dtype=np.int32)
        arr = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (qwall, hurel, resc, ress)]
        ptr = [None if x is None else x.ctypes.data_as(L.DP) for x in arr]
        L._check(self.lib.udc_set_ibm_wallmoist(self.h, int(iwallmoist), int(lg.size), lg.ctypes.data_as(C.POINTER(C.c_int)), *ptr), "udc_set_ibm_wallmoist")

    def set_floor_air_temperature(self, thl_kb):
        """ltempeq off + wfuno floor: the frozen temperature of the first level (include/udcore.h)."""
        L._check(self.lib.udc_set_floor_air_temperature(self.h, C.c_double(thl_kb)), "udc_set_floor_air_temperature")

    def set_poisson_bczp(self, bczp):
        """&BC BCzp: 1 the tridiagonal solve in z, 2 the cosine transform's solution (include/udcore.h udc_set_poisson_bczp)."""
        L._check(self.lib.udc_set_poisson_bczp(self.h, int(bczp)), "udc_set_poisson_bczp")

    def set_fkar(self, fkar):
        """&WALLS fkar: the von Karman constant of the floor and facet wall functions (include/udcore.h udc_set_fkar)."""
        L._check(self.lib.udc_set_fkar(self.h, C.c_double(fkar)), "udc_set_fkar")

    def set_floor_wf(self, bcbotm=3, bcbott=1, thls=-1., z0h=-1., prandtlturb=0.71):
        """Floor wall function choice of `bottom` (BCbotm 2 / BCbotT 2 = wfuno), see include/udcore.h udc_set_floor_wf."""
        L._check(self.lib.udc_set_floor_wf(self.h, int(bcbotm), int(bcbott), C.c_double(thls), C.c_double(z0h),
                                           C.c_double(prandtlturb)), "udc_set_floor_wf")

    def set_moisture(self, iadv_qt=2, bctopq=1, wqtop=0., qt_top=-1., bcbotq=1, wqsurf=0.):
        """&PHYSICS lmoist: qt becomes a transported field, see include/udcore.h udc_set_moisture."""
        L._check(self.lib.udc_set_moisture(self.h, int(iadv_qt), int(bctopq), C.c_double(wqtop), C.c_double(qt_top),
                                           int(bcbotq), C.c_double(wqsurf)), "udc_set_moisture")
        self.lmoist = True

    TH_TABLES = ("presf", "presh", "exnf", "exnh", "thvh", "thl0av", "qt0av", "ql0av", "th0av")

    def set_moist_thermo(self, thls, qts, ps=101325., lqlnr=False):
        """Moist thermodynamics (needed by lmoist with lbuoyancy), see include/udcore.h udc_set_moist_thermo."""
        zf = np.ascontiguousarray(self.g.zf[1:self.g.nz + 2], dtype=np.float64)
        zh = np.ascontiguousarray(self.g.zh[1:self.g.nz + 2], dtype=np.float64)
        L._check(self.lib.udc_set_moist_thermo(self.h, C.c_double(thls), C.c_double(qts), C.c_double(ps),
                                               zf.ctypes.data_as(L.DP), zh.ctypes.data_as(L.DP), len(zf), int(bool(lqlnr))),
                 "udc_set_moist_thermo")
        self.moist_thermo, self._thermo_started = True, False

    def calthv(self):
        """Record the top ghost planes of thl0 / qt0 as calthv would see them now (include/udcore.h udc_calthv)."""
        L._check(self.lib.udc_calthv(self.h), "udc_calthv")

    def set_open_x_thl(self, thlprof):
        """BCxT = 2 on an open-x core: the temperature's inflow profile, [ktot+2] by the reference's k."""
        a = np.ascontiguousarray(thlprof, dtype=np.float64)
        assert a.size == self.g.nz + 2
        L._check(self.lib.udc_set_open_x_thl(self.h, a.ctypes.data_as(L.DP)), "udc_set_open_x_thl")

    def set_open_x_qt(self, qtprof):
        """BCxq = 2 on an open-x core: the total water's inflow profile, [ktot+2] by the reference's k."""
        a = np.ascontiguousarray(qtprof, dtype=np.float64)
        assert a.size == self.g.nz + 2
        L._check(self.lib.udc_set_open_x_qt(self.h, a.ctypes.data_as(L.DP)), "udc_set_open_x_qt")

    def set_open_x_inlet(self, u, v, w):
        """BCxm = 3: the inlet's planes [nz+2][ny+2] (k = kb-1 .. ke+1, j = jb-1 .. je+1) as the reference's drivergen leaves u0driver,
        v0driver, w0driver (the m planes are the same arrays: both are refreshed on the same calls); applied by the next `boundary`."""
        a = [np.ascontiguousarray(q, dtype=np.float64) for q in (u, u, v, v, w, w)]
        assert all(q.shape == (self.g.nz + 2, self.nyl + 2) for q in a)
        lb, ub = (C.c_int * 2)(0, 0), (C.c_int * 2)(self.nyl + 1, self.g.nz + 1)
        L._check(self.lib.udc_set_open_x_inlet(self.h, *[q.ctypes.data_as(L.DP) for q in a], lb, ub), "udc_set_open_x_inlet")

    def set_open_x_inlet_scalar(self, field, plane, halo=1):
        """BCxT / BCxq / BCxs = 3: a scalar's inlet plane [nz+2 halo][ny+2 halo] (thl0driver, qt0driver, sv0driver(:, :, n))."""
        a = np.ascontiguousarray(plane, dtype=np.float64)
        assert a.shape == (self.g.nz + 2 * halo, self.nyl + 2 * halo)
        lb, ub = (C.c_int * 2)(1 - halo, 1 - halo), (C.c_int * 2)(self.nyl + halo, self.g.nz + halo)
        L._check(self.lib.udc_set_open_x_inlet_scalar(self.h, int(field), a.ctypes.data_as(L.DP), a.ctypes.data_as(L.DP), lb, ub),
                 "udc_set_open_x_inlet_scalar")

    def set_open_x_scalars(self, svprof):
        """BCxs = 2 on an open-x core: the scalars' inflow profiles [nsv][ktot+2] by the reference's k."""
        a = np.ascontiguousarray(svprof, dtype=np.float64)
        assert a.shape == (self.nsv, self.g.nz + 2)
        L._check(self.lib.udc_set_open_x_scalars(self.h, a.ctypes.data_as(L.DP)), "udc_set_open_x_scalars")

    def set_boundary_rk3coef(self, rk3coef):
        L._check(self.lib.udc_set_boundary_rk3coef(self.h, C.c_double(rk3coef)), "udc_set_boundary_rk3coef")

    def start_up(self, before_boundary=None, dtmax=None):
        """What the reference does between filling the fields and entering the loop (src/modstartup.f90:1601, src/program.f90:118):
        `thermodynamics` on the state as readinitfiles left it -- grid.cold_start(..., pre_boundary=True) --, then `boundary`.
        `before_boundary`: a callable run in between (the host's level forcings take diagfld's slab averages there)."""
        if getattr(self, "moist_thermo", False):
            self.thermodynamics()
        else:
            self.calthv()
        if before_boundary is not None:
            before_boundary()
        if self.open_x and dtmax is not None:
            self.open_x_startup(dtmax / 100.)      # (a cold start's dt, src/modstartup.f90:1099)
        self.boundary()

    def open_x_startup(self, dt):
        """BCxm = 2, ahead of the start-up's `boundary` (src/program.f90:118): diagfld has just formed u0av from the fields as
        readinitfiles left them (cold start) or as the restart files hold them; that `boundary` takes its uouttot from it -- and so does
        the first substep's -- and convects the outlet once with rk3step = 0 and the start-up's dt (src/modboundary.f90:141-160, 914)."""
        g = self.g
        if getattr(self, "_open_x_avg", True):      # (a prescribed volume flow convects the outlet with ubulk instead: from_deck)
            wl = g.dzf[1:g.nz + 1] / (g.zh[g.nz + 1] - g.zh[2])
            u0av = self.slab_average("u0")
            self.set_open_x_outflow(wl, float(np.sum(np.asarray(u0av)[1:g.nz + 1] * wl)), hold_first=True)
        self.set_boundary_rk3coef(dt / 4.)

    def thermodynamics(self):
        """The reference's `thermodynamics` (src/program.f90:120 before the loop, :214 at the end of every substep)."""
        L._check(self.lib.udc_thermodynamics(self.h), "udc_thermodynamics")
        self._thermo_started = True

    def thermo_state(self, tables=None):
        """Read (tables None) or write the per-level state of the moist thermodynamics: dict name -> [nz+2] arrays
        indexed by the reference's k (entry 0 unused), names in TH_TABLES."""
        n = self.g.nz + 1
        a = np.zeros((len(self.TH_TABLES), n))
        if tables is not None:
            for q, name in enumerate(self.TH_TABLES):
                a[q] = np.asarray(tables[name], dtype=np.float64)[1:n + 1]
        L._check(self.lib.udc_thermo_state(self.h, a.ctypes.data_as(L.DP), n, 0 if tables is None else 1), "udc_thermo_state")
        if tables is not None:
            self._thermo_started = True
            return None
        return {name: np.concatenate(([0.], a[q])) for q, name in enumerate(self.TH_TABLES)}

    def _ensure_thermo(self):
        # program.f90:120: one thermodynamics call on the start state before the first forces
        if getattr(self, "moist_thermo", False) and not self._thermo_started:
            self.thermodynamics()

    def slab_average(self, field):
        """Horizontal mean of `field` per level, indexed by the reference's k: entries 1..nz+1 (entry 0 unused)."""
        from .forcings import field_id
        fid = field_id(field) if isinstance(field, str) else field
        a = np.zeros(self.g.nz + 2)
        L._check(self.lib.udc_slab_average(self.h, fid, a[1:].ctypes.data_as(L.DP), self.g.nz + 1), "udc_slab_average")
        return a

    def slab_averages(self, fields):
        """Horizontal means of several fields per level in one device round trip: {name: [nz+2] by the reference's k}."""
        from .forcings import field_id
        n = self.g.nz + 1
        a = np.zeros((len(fields), n))
        batch = 16                              # one reduction takes at most 16 fields
        for q0 in range(0, len(fields), batch):
            part = fields[q0:q0 + batch]
            ids = (C.c_int * len(part))(*[field_id(f) if isinstance(f, str) else f for f in part])
            L._check(self.lib.udc_slab_averages(self.h, ids, len(part), a[q0:q0 + len(part)].ctypes.data_as(L.DP), n), "udc_slab_averages")
        return {f: np.concatenate(([0.], a[q])) for q, f in enumerate(fields)}

    def set_level_forcing(self, tend, src, A, B, when=0):
        """tend(i,j,k) += A(k) + B(k) src(i,j,k) inside every following substep; A=None removes it."""
        from .forcings import field_id
        t = field_id(tend) if isinstance(tend, str) else tend
        s_ = -1 if src is None else (field_id(src) if isinstance(src, str) else src)
        if A is None:
            L._check(self.lib.udc_set_level_forcing(self.h, t, s_, None, None, 0, int(when)), "udc_set_level_forcing")
            return
        a = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(B if B is not None else np.zeros_like(a), dtype=np.float64)
        L._check(self.lib.udc_set_level_forcing(self.h, t, s_, a.ctypes.data_as(L.DP), b.ctypes.data_as(L.DP), len(a),
                                                int(when)), "udc_set_level_forcing")

    def level_forcings(self, when=0):
        L._check(self.lib.udc_level_forcings(self.h, int(when)), "udc_level_forcings")

    def set_coriolis(self, mode, om22, om23, ug=None):
        """mode 1 = &PHYSICS lcoriol, 2 = lprofforc (relaxation towards ug(k)); src/modforces.f90:600-717."""
        n = 0 if ug is None else len(ug)
        a = None if ug is None else np.ascontiguousarray(ug, dtype=np.float64)
        L._check(self.lib.udc_set_coriolis(self.h, int(mode), C.c_double(om22), C.c_double(om23),
                                           None if a is None else a.ctypes.data_as(L.DP), n), "udc_set_coriolis")

    def coriolis(self):
        L._check(self.lib.udc_coriolis(self.h), "udc_coriolis")

    def set_tke(self, cf=2.5, cn=0.76, Rigc=0.25, Prandtl=0.333, ch1=1., thvs=288., ldelta=False, e12min=5.e-5,
                grav=9.81):
        """&NAMSUBGRID loneeqn: constants as initsubgrid derives them (src/modsubgrid.f90:63-71)."""
        k = tke_constants(cf, cn, Rigc, Prandtl, ch1)
        L._check(self.lib.udc_set_tke(self.h, C.c_double(k["cm"]), C.c_double(cn), C.c_double(ch1), C.c_double(k["ch2"]),
                                      C.c_double(k["ce1"]), C.c_double(k["ce2"]), C.c_double(e12min), C.c_double(grav),
                                      C.c_double(thvs), int(bool(ldelta))), "udc_set_tke")
        self.loneeqn = True

    def set_buoyancy(self, on=True, grav=9.81):
        """&PHYSICS lbuoyancy (dry air): forces adds grav (thv0h - thvh)/thvh to wp."""
        L._check(self.lib.udc_set_buoyancy(self.h, int(bool(on)), C.c_double(grav)), "udc_set_buoyancy")

    def set_buoycorr(self, on=True, rigc=0.25):
        """&NAMSUBGRID lbuoycorr (Vreman closure with lbuoyancy): stable-stratification correction of ekm."""
        L._check(self.lib.udc_set_buoycorr(self.h, int(bool(on)), C.c_double(rigc)), "udc_set_buoycorr")

    def masscorr(self):
        """masscorr (src/modforces.f90:328), after forces."""
        L._check(self.lib.udc_masscorr(self.h, self.rk3step, C.c_double(self.dt)), "udc_masscorr")

    def poisson(self):
        L._check(self.lib.udc_poisson(self.h, self.rk3step, C.c_double(self.dt)), "udc_poisson")

    def tstep_integrate(self):
        L._check(self.lib.udc_tstep_integrate(self.h, self.rk3step, C.c_double(self.dt)), "udc_tstep_integrate")

    def halos(self):
        L._check(self.lib.udc_halos(self.h), "udc_halos")

    def boundary(self):
        L._check(self.lib.udc_boundary(self.h), "udc_boundary")

    # ---- fused fast path
    def substep(self, rk3step, dt, with_forces=True):
        self._ensure_thermo()
        L._check(self.lib.udc_substep(self.h, rk3step, C.c_double(dt), 1 if with_forces else 0), "udc_substep")

    def run(self, nsub, dt, rk3step0=1, with_forces=True):
        self._ensure_thermo()
        L._check(self.lib.udc_run(self.h, nsub, rk3step0, C.c_double(dt), 1 if with_forces else 0), "udc_run")

    # ---- immersed boundary (sparse corrections, udcore/ibm.py reads the reference's input files)
    def set_ibm_points(self, grid, solid, bound):
        s = np.ascontiguousarray(solid, dtype=np.int32).reshape(-1, 3)
        b = np.ascontiguousarray(bound, dtype=np.int32).reshape(-1, 3)
        ip = C.POINTER(C.c_int)
        L._check(self.lib.udc_set_ibm_points(self.h, int(grid), s.ctypes.data_as(ip), len(s), b.ctypes.data_as(ip), len(b)),
                 "udc_set_ibm_points")

    def bottom_diagnostics(self, on=True):
        """Keep tau_x, tau_y, thl_flux of `bottom` (what it adds to up, vp, thlp on the k = kb plane)."""
        L._check(self.lib.udc_bottom_diagnostics(self.h, 1 if on else 0), "udc_bottom_diagnostics")

    def bottom_diag(self, which):
        """which: "tau_x" | "tau_y" | "thl_flux" -> [ny_l, nx]."""
        out = np.empty((self.nyl, self.g.nx))
        L._check(self.lib.udc_bottom_diag_get(self.h, {"tau_x": 0, "tau_y": 1, "thl_flux": 2}[which],
                                              out.ctypes.data_as(C.POINTER(C.c_double))), "udc_bottom_diag_get")
        return out

    def set_ibm_mask_wrap(self, wrapx, wrapy):
        """Mask look-ups across the domain's lateral boundaries: periodic image (True) or "fluid" (False) -- the
        reference's masks hold the latter in a direction its run keeps on one rank (udcore.h)."""
        L._check(self.lib.udc_set_ibm_mask_wrap(self.h, int(bool(wrapx)), int(bool(wrapy))), "udc_set_ibm_mask_wrap")

    def set_ibm_conservative(self, on):
        """lconservativeibm: advecc2nd_corr_conservative instead of _liberal for thl / qt next to obstacles."""
        L._check(self.lib.udc_set_ibm_conservative(self.h, int(bool(on))), "udc_set_ibm_conservative")

    def ibm_commit(self):
        L._check(self.lib.udc_ibm_commit(self.h), "udc_ibm_commit")

    def ibmwallfun(self):
        """ibmwallfun (src/modibm.f90:1167) without facet wall functions, after nudge."""
        L._check(self.lib.udc_ibmwallfun(self.h), "udc_ibmwallfun")

    def ibmnorm(self):
        """ibmnorm (src/modibm.f90:697), after masscorr."""
        L._check(self.lib.udc_ibmnorm(self.h), "udc_ibmnorm")

    # ---- deferred execution: the routine-by-routine surface runs as the fused substep (include/udcore.h)
    def set_deferred(self, on=True):
        L._check(self.lib.udc_set_deferred(self.h, 1 if on else 0), "udc_set_deferred")

    def flush(self):
        L._check(self.lib.udc_flush(self.h), "udc_flush")

    def deferred_stats(self):
        """(substeps that ran fused, substeps that ran routine by routine) since the handle was created."""
        a, b = C.c_long(), C.c_long()
        L._check(self.lib.udc_deferred_stats(self.h, C.byref(a), C.byref(b)), "udc_deferred_stats")
        return a.value, b.value

    def divergence(self):
        a, b = C.c_double(), C.c_double()
        L._check(self.lib.udc_divergence(self.h, C.byref(a), C.byref(b)), "udc_divergence")
        return a.value, b.value

    def sync(self):
        L._check(self.lib.udc_sync(self.h), "udc_sync")

    # ---- measurement
    def profile(self, on=True, focus=None, every=1):
        """HIP-event timing of the launches: every one (on=True), or only those whose name starts with `focus`
        (two events per such launch and none elsewhere: affordable inside a timed region), in every `every`-th substep."""
        if on and focus:
            L._check(self.lib.udc_profile_focus(self.h, focus.encode()), "udc_profile_focus")
            L._check(self.lib.udc_profile_every(self.h, int(every)), "udc_profile_every")
            L._check(self.lib.udc_profile_enable(self.h, 2), "udc_profile_enable")
        else:
            L._check(self.lib.udc_profile_enable(self.h, 1 if on else 0), "udc_profile_enable")

    def profile_reset(self):
        self.lib.udc_profile_reset(self.h)

    def profile_get(self):
        cap = 64
        names = ((C.c_char * 64) * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_int * cap)()
        n = self.lib.udc_profile_get(self.h, cap, names, ms, cnt)
        return {names[i].value.decode(): (ms[i], cnt[i]) for i in range(min(n, cap))}
