"""ctypes binding of the CPU oracle (oracle/_ref/liboracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are numpy float64, C-contiguous, shaped [k, j, i] (== Fortran (i, j, k) storage):
m-arrays (nz+2, ny+2, nx+2), c-arrays (nz+4, ny+4, nx+4).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "liboracle.so")

DP = C.POINTER(C.c_double)


class OrcGrid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("dx", C.c_double), ("dy", C.c_double),
                ("dzf", DP), ("dzh", DP),
                ("numol", C.c_double), ("prandtlmoli", C.c_double), ("prandtli", C.c_double),
                ("c_vreman", C.c_double), ("csz", C.c_double),
                ("sgs", C.c_int), ("bctopm", C.c_int),
                ("uinf", C.c_double), ("vinf", C.c_double), ("nsv", C.c_int),
                ("lbottom", C.c_int), ("z0", C.c_double),
                ("luvolflowr", C.c_int), ("lvvolflowr", C.c_int), ("uflowrate", C.c_double), ("vflowrate", C.c_double),
                ("ltempeq", C.c_int), ("bctopt", C.c_int), ("wttop", C.c_double), ("thl_top", C.c_double),
                ("wtsurf", C.c_double), ("lbuoyancy", C.c_int),
                ("coriolis_mode", C.c_int), ("om22", C.c_double), ("om23", C.c_double),
                ("cm", C.c_double), ("cn", C.c_double), ("ch1", C.c_double), ("ch2", C.c_double),
                ("ce1", C.c_double), ("ce2", C.c_double), ("thvs", C.c_double), ("ldelta", C.c_int),
                ("lmoist", C.c_int), ("bctopq", C.c_int), ("wqtop", C.c_double), ("qt_top", C.c_double),
                ("wqsurf", C.c_double), ("thls", C.c_double), ("qts", C.c_double), ("ps", C.c_double),
                ("zf", DP), ("zh", DP), ("bcbotm", C.c_int), ("bcbott", C.c_int), ("z0h", C.c_double),
                ("prandtlturb", C.c_double), ("bctops", C.c_int), ("wsvtop", C.c_double * 4), ("sv_top", C.c_double * 4),
                ("lchem", C.c_int), ("k1", C.c_double), ("JNO2", C.c_double), ("lqlnr", C.c_int), ("iadv_thl", C.c_int),
                ("lbuoycorr", C.c_int), ("Rigc", C.c_double)]


class OrcIbm(C.Structure):
    _fields_ = [("sol", C.POINTER(C.c_int) * 4), ("nsol", C.c_int * 4), ("bnd", C.POINTER(C.c_int) * 4), ("nbnd", C.c_int * 4),
                ("mask", DP * 4), ("conservative", C.c_int)]


class OrcState(C.Structure):
    _fields_ = [(n, DP) for n in ("u0", "v0", "w0", "um", "vm", "wm", "up", "vp", "wp", "pres0",
                                  "ekm", "ekh", "p", "pup", "pvp", "pwp", "sv0", "svm", "svp",
                                  "dpdxl", "dpdyl", "thl0", "thlm", "thlp", "thlpcar", "ug", "e120", "e12m", "e12p",
                                  "qt0", "qtm", "qtp", "thermo", "ql0", "thl0c", "svsrc")]


TH_TABLES = ("presf", "presh", "exnf", "exnh", "thvh", "thl0av", "qt0av", "ql0av", "th0av")      # ORC_TH_* order


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
    return _lib


def ptr(a):
    if a is None:
        return C.cast(None, DP)
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(DP)


class Oracle:
    """Grid + convenience wrappers around the orc_* functions."""

    def __init__(self, nx, ny, nz, dx, dy, dzf, dzh, sgs=2, bctopm=1, nsv=0, numol=1.5e-5,
                 prandtlmoli=1. / 0.71, prandtli=1. / 0.333, c_vreman=0.07, csz=None,
                 uinf=0., vinf=0., lbottom=False, z0=0.05, luvolflowr=False, uflowrate=0.,
                 lvvolflowr=False, vflowrate=0., ltempeq=False, bctopt=1, wttop=0., thl_top=-1., wtsurf=-1.,
                 lbuoyancy=False, coriolis_mode=0, om22=0., om23=0., tke=None,
                 lmoist=False, bctopq=1, wqtop=0., qt_top=-1., wqsurf=-1., thls=-1., qts=-1., ps=101325., zf=None, zh=None,
                 bcbotm=3, bcbott=1, z0h=-1., prandtlturb=0.71, iadv_thl=2, lqlnr=False, bctops=1, wsvtop=(), sv_top=(),
                 lchem=False, k1=0., JNO2=0., lbuoycorr=False, Rigc=0.25):
        self.nx, self.ny, self.nz, self.nsv = nx, ny, nz, nsv
        self.dzf = np.ascontiguousarray(dzf, dtype=np.float64)
        self.dzh = np.ascontiguousarray(dzh, dtype=np.float64)
        assert self.dzf.shape == (nz + 2,) and self.dzh.shape == (nz + 2,)
        if csz is None:
            cf, alpha = 2.5, 1.5
            cm = cf / (2. * np.pi) * (1.5 * alpha) ** (-1.5)
            ceps = 2. * np.pi / cf * (1.5 * alpha) ** (-1.5)
            csz = (cm ** 3 / ceps) ** 0.25
        self.zf = None if zf is None else np.ascontiguousarray(zf, dtype=np.float64)
        self.zh = None if zh is None else np.ascontiguousarray(zh, dtype=np.float64)
        self.g = OrcGrid(nx, ny, nz, dx, dy, ptr(self.dzf), ptr(self.dzh), numol, prandtlmoli,
                         prandtli, c_vreman, csz, sgs, bctopm, uinf, vinf, nsv, int(bool(lbottom)), z0,
                         int(bool(luvolflowr)), int(bool(lvvolflowr)), uflowrate, vflowrate,
                         int(bool(ltempeq)), bctopt, wttop, thl_top, wtsurf, int(bool(lbuoyancy)),
                         coriolis_mode, om22, om23,
                         *((tke["cm"], tke["cn"], tke["ch1"], tke["ch2"], tke["ce1"], tke["ce2"], tke["thvs"],
                            int(tke.get("ldelta", 0))) if tke else (0., 0., 0., 0., 0., 0., 1., 0)),
                         int(bool(lmoist)), bctopq, wqtop, qt_top, wqsurf, thls, qts, ps, ptr(self.zf), ptr(self.zh),
                         bcbotm, bcbott, z0h, prandtlturb, bctops,
                         (C.c_double * 4)(*(list(wsvtop) + [0.] * 4)[:4]), (C.c_double * 4)(*(list(sv_top) + [0.] * 4)[:4]),
                         int(bool(lchem)), k1, JNO2, int(bool(lqlnr)), iadv_thl, int(bool(lbuoycorr)), Rigc)
        self.L = lib()
        self.L.orc_set_poisson_bczp(1)      # (a static switch of the library: every new oracle starts from the tridiagonal solve)

    def mshape(self):
        return (self.nz + 2, self.ny + 2, self.nx + 2)

    def cshape(self):
        return (self.nz + 4, self.ny + 4, self.nx + 4)

    def call(self, name, *args):
        f = getattr(self.L, name)
        f.restype = None
        cargs = [C.byref(self.g)]
        for a in args:
            if isinstance(a, np.ndarray) or a is None:
                cargs.append(ptr(a))
            elif isinstance(a, float):
                cargs.append(C.c_double(a))
            elif isinstance(a, int):
                cargs.append(C.c_int(a))
            else:
                cargs.append(a)
        f(*cargs)

    def thermo_tables(self):
        """Zeroed state block of the moist thermodynamics: ORC_TH_N tables of [nz+2] and the 'diagfld has run' flag."""
        return np.zeros(len(TH_TABLES) * (self.nz + 2) + 1)

    def state(self, st: dict):
        return OrcState(*[ptr(st.get(n)) for n, _ in OrcState._fields_])

    def thermodynamics(self, st: dict):
        f = self.L.orc_thermodynamics
        f.restype = None
        s = self.state(st)
        f(C.byref(self.g), C.byref(s))

    # ---- immersed boundary: lists = {grid letter: (solid[n,3], boundary[n,3])} as udcore.ibm.read_ibm returns them
    def set_ibm(self, lists, wrapx=False, wrapy=False, conservative=False):
        """Build the masks (orc_ibm_mask) and make orc_substep run ibmwallfun / ibmnorm; None switches it off.
        wrapx / wrapy: the masks' lateral ghost cells are periodic images (a reference run that splits the direction
        over ranks) or stay "fluid" (one rank: the fixtures' build)."""
        if lists is None:
            self.L.orc_set_ibm(None)
            self._ibm = None
            return None
        ip = C.POINTER(C.c_int)
        keep, b = [], OrcIbm()
        for q, gname in enumerate("uvwc"):
            sol, bnd = lists.get(gname, (np.zeros((0, 3), np.int32), np.zeros((0, 3), np.int32)))
            sol, bnd = np.ascontiguousarray(sol, dtype=np.int32), np.ascontiguousarray(bnd, dtype=np.int32)
            mask = np.zeros(self.mshape())
            f = self.L.orc_ibm_mask
            f.restype = None
            f(C.byref(self.g), C.c_int(1 if gname == "w" else 0), sol.ctypes.data_as(ip), C.c_int(len(sol)), ptr(mask),
              C.c_int(int(wrapx)), C.c_int(int(wrapy)))
            keep += [sol, bnd, mask]
            b.sol[q], b.nsol[q] = sol.ctypes.data_as(ip), len(sol)
            b.bnd[q], b.nbnd[q] = bnd.ctypes.data_as(ip), len(bnd)
            b.mask[q] = ptr(mask)
        b.conservative = int(bool(conservative))      # lconservativeibm: advecc2nd_corr_conservative instead of _liberal
        self._ibm = (b, keep)
        self.L.orc_set_ibm(C.byref(b))
        return {g: keep[3 * q + 2] for q, g in enumerate("uvwc")}

    def substep(self, st: dict, rk3step: int, dt: float):
        s = self.state(st)
        f = self.L.orc_substep
        f.restype = None
        f(C.byref(self.g), C.byref(s), C.c_int(rk3step), C.c_double(dt))


def fft_r2c(x):
    L = lib()
    n = len(x)
    L.fft_ref_plan_create.restype = C.c_void_p
    p = C.c_void_p(L.fft_ref_plan_create(C.c_int(n)))
    xin = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(2 * (n // 2 + 1))
    L.fft_ref_r2c(p, ptr(xin), ptr(out))
    L.fft_ref_plan_destroy(p)
    return out[0::2] + 1j * out[1::2]


def fft_c2r(X, n):
    L = lib()
    L.fft_ref_plan_create.restype = C.c_void_p
    p = C.c_void_p(L.fft_ref_plan_create(C.c_int(n)))
    xin = np.zeros(2 * (n // 2 + 1))
    xin[0::2] = X.real
    xin[1::2] = X.imag
    out = np.zeros(n)
    L.fft_ref_c2r(p, ptr(xin), ptr(out))
    L.fft_ref_plan_destroy(p)
    return out
