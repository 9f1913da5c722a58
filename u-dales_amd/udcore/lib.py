"""ctypes binding of libudcore.so (the C ABI declared in include/udcore.h).

There is no CPU fallback: importing works anywhere (so the host logic can be tested on CPU),
but creating a `DynCore` requires the built library and a visible MI355X; otherwise it raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("UDC_LIBPATH", os.path.join(os.path.dirname(HERE), "lib", "libudcore.so"))   # override: A/B builds

DP = C.POINTER(C.c_double)

# field ids (include/udcore.h)
U0, V0, W0, UM, VM, WM, UP, VP, WP, PRES0, P, EKM, EKH, SV0, SVM, SVP = range(16)
THL0, THLM, THLP = SV0 + 45, SVM + 45, SVP + 45      # temperature equation: scalar slot 15 (include/udcore.h)
E120, E12M, E12P = SV0 + 42, SVM + 42, SVP + 42      # one-equation closure: scalar slot 14
QT0, QTM, QTP = SV0 + 39, SVM + 39, SVP + 39         # moisture: scalar slot 13
FIELD_IDS = dict(u0=U0, v0=V0, w0=W0, um=UM, vm=VM, wm=WM, up=UP, vp=VP, wp=WP, pres0=PRES0, p=P,
                 ekm=EKM, ekh=EKH, thl0=THL0, thlm=THLM, thlp=THLP, e120=E120, e12m=E12M, e12p=E12P,
                 qt0=QT0, qtm=QTM, qtp=QTP, ql0=SV0 + 48)
SGS_DNS, SGS_SMAGORINSKY, SGS_VREMAN, SGS_ONEEQN = 0, 1, 2, 3

EXPORTS = ["udc_create", "udc_destroy", "udc_last_error", "udc_version", "udc_comm_unique_id",
           "udc_comm_init", "udc_comm_info", "udc_comm_stats", "udc_comm_dry_run", "udc_last_plan", "udc_field_upload", "udc_field_download", "udc_set_forcing",
           "udc_advection", "udc_subgrid", "udc_bottom", "udc_forces", "udc_slab_average", "udc_slab_averages", "udc_set_level_forcing", "udc_level_forcings", "udc_set_coriolis", "udc_coriolis", "udc_set_masscorr", "udc_set_masscorr_outflow", "udc_masscorr", "udc_set_tempeq", "udc_set_thl_source", "udc_set_floor_wf", "udc_set_fkar", "udc_set_chem", "udc_set_shifted_pbc", "udc_shifted_pbcs", "udc_set_scalar_top", "udc_set_scalar_source", "udc_scalsource", "udc_set_moisture", "udc_set_moist_thermo", "udc_thermodynamics", "udc_calthv", "udc_thermo_state", "udc_set_buoyancy", "udc_set_buoycorr", "udc_set_tke", "udc_poisson", "udc_tstep_integrate",
           "udc_halos", "udc_boundary", "udc_tstep_maxima", "udc_substep", "udc_run",
           "udc_set_deferred", "udc_flush", "udc_deferred_stats",
           "udc_stats_enable", "udc_stats_sample", "udc_stats_get", "udc_stats_set_masks", "udc_stats_xyt", "udc_stats_set_forced", "udc_stats_yt", "udc_stats_xy", "udc_stats_y", "udc_set_floor_air_temperature", "udc_set_ibm_wallfun", "udc_set_ibm_sections", "udc_set_ibm_wallheat", "udc_set_ibm_wallmoist", "udc_set_poisson_bczp", "udc_set_ibm_facet_output", "udc_ibm_facet_sample", "udc_ibm_facet_get", "udc_set_scalar_bcx", "udc_set_scalar_bcx_outflow", "udc_create_open_x", "udc_set_open_x_outflow", "udc_set_open_x_sample_gap", "udc_set_boundary_rk3coef", "udc_set_open_x_profile", "udc_set_open_x_inlet", "udc_set_open_x_thl", "udc_set_open_x_qt", "udc_set_open_x_inlet_scalar", "udc_set_open_x_scalars", "udc_set_ibm_points", "udc_ibm_commit", "udc_ibmwallfun", "udc_ibmnorm",
           "udc_divergence", "udc_checksim", "udc_checksim_begin", "udc_checksim_end", "udc_sync", "udc_profile_enable", "udc_profile_reset",
           "udc_profile_get", "udc_profile_focus", "udc_profile_every", "udc_set_ibm_mask_wrap", "udc_bottom_diagnostics", "udc_bottom_diag_get", "udc_set_ibm_conservative"]


class UdcConfig(C.Structure):
    _fields_ = [("itot", C.c_int), ("jtot", C.c_int), ("ktot", C.c_int),
                ("nranks", C.c_int), ("rank", C.c_int), ("device", C.c_int),
                ("dx", C.c_double), ("dy", C.c_double),
                ("dzf", DP), ("dzh", DP),
                ("numol", C.c_double), ("prandtlmoli", C.c_double), ("prandtli", C.c_double),
                ("c_vreman", C.c_double), ("csz", C.c_double),
                ("sgs", C.c_int), ("bctopm", C.c_int),
                ("uinf", C.c_double), ("vinf", C.c_double), ("nsv", C.c_int),
                ("lbottom", C.c_int), ("z0", C.c_double)]


class UdcError(RuntimeError):
    pass


_lib = None


def load():
    """Load libudcore.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise UdcError(f"{LIBPATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        # torch (bench.py, the tests, `python -m torch.distributed.run`) ships its own copies of the ROCm runtime libraries under
        # the same sonames as the system's.  Whichever is loaded first serves both; torch-first works, libudcore-first makes the
        # process abort at exit (glibc: "double free or corruption" in the runtimes' tear-down).  So where torch is installed it
        # goes first (UDC_TORCH_FIRST=0: not).  The Fortran route never meets torch.
        # (only where torch is installed at all -- found without importing it -- and said out loud under UDC_VERBOSE=1: an implicit
        # import of this size should not be a secret)
        if "torch" not in sys.modules and os.environ.get("UDC_TORCH_FIRST", "1") != "0":
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                try:
                    import torch      # noqa: F401
                    if os.environ.get("UDC_VERBOSE", "0") not in ("", "0"):
                        print("udcore: imported torch ahead of libudcore.so (two ROCm runtimes under one soname; UDC_TORCH_FIRST=0 skips this)",
                              file=sys.stderr)
                except Exception as e:      # noqa: BLE001
                    print(f"udcore: torch is installed but did not import ({e!r}); loading libudcore.so without it", file=sys.stderr)
        _lib = C.CDLL(LIBPATH, mode=C.RTLD_GLOBAL)
        _lib.udc_last_error.restype = C.c_char_p
    return _lib


TEST_EXPORTS = ["udc_local_group_create", "udc_comm_init_local", "udc_comm_init_shm"]      # libudcore_test.so only (the virtual-rank tests' transport)
TESTLIBPATH = os.path.join(os.path.dirname(LIBPATH), "libudcore_test.so")


class test_transport:
    """Context manager: inside, load() returns libudcore_test.so -- the library built with -DUDC_TEST_TRANSPORT, which adds the
    in-process `local group` transport (several virtual ranks on one GPU).  Handles made inside must be closed inside."""

    def __enter__(self):
        global _lib
        if not os.path.exists(TESTLIBPATH):
            raise UdcError(f"{TESTLIBPATH} not found (make -C u-dales_amd/csrc)")
        self.saved = _lib
        _lib = C.CDLL(TESTLIBPATH, mode=C.RTLD_LOCAL)
        _lib.udc_last_error.restype = C.c_char_p
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def scalar_field(kind: int, n: int) -> int:
    """Field id of scalar n: kind is SV0, SVM or SVP."""
    return kind + 3 * n


def _check(rc, what):
    if rc != 0:
        raise UdcError(f"{what}: {load().udc_last_error().decode()}")
