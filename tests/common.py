"""Shared helpers for the parity tests: golden fixtures <-> oracle / device arrays."""
from __future__ import annotations

import gzip
import os

import numpy as np

from refdump import read_dump

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

KERNEL_CASES = {
    "k_vreman_12x8x6": 11, "k_smag_8x12x10s": 12, "k_noslip_8x8x6": 13,
    "k_scalar_8x8x8": 14, "k_dns_8x8x6": 15, "k_floor_12x8x6": 16,
    "k_volflow_12x8x6": 17, "k_thl_12x8x6": 18,
    "k_buoy_12x8x6": 19,
    "k_coriol_12x8x6": 20,
    "k_tke_12x8x6": 21, "k_tke_thl_12x8x6": 28, "k_qt_12x8x6": 32, "k_moist_12x8x8": 35, "k_uno_12x8x6": 37, "k_src_12x8x8": 39, "k_thlk_12x8x6": 41, "k_svtop_8x8x8": 47, "k_tke_moist_12x8x8": 50, "k_vreman_buoycorr_12x8x10": 52, "k_floor_uno_nothl_12x8x6": 64, "k_bcxs_16x8x12": 75,
    "k_ptop_12x8x6": 84,      # the open lid (BCtopm = 3)
    "k_bczp2_12x8x8": 88,     # BCzp = 2: cosine transform in z
}
# per-level forcings (lstend, nudge, grwdamp): host-level routines, checked in tests/test_level_forcings.py
LSF_CASES = {"k_lsf_12x8x24": 29, "run_lsf_16x8x24s": 30, "k_lsfq_12x8x20": 34}
RUN_CASES = {"run_16x16x8": 21, "run_smag_scalar_16x8x12s": 22, "run_floor_scalar_16x8x12s": 23,
             "run_volflow_uv_16x16x8": 24, "run_thl_16x8x12s": 25, "run_qt_16x8x12s": 33, "run_moist_16x8x12s": 36, "run_uno_16x8x12s": 38, "run_src_16x8x12s": 40, "run_thlk_16x8x12s": 42, "run_moistnr_16x8x12s": 45, "run_svflux_16x8x12s": 48, "run_chem_16x8x12s": 51,
             "run_buoy_16x8x12s": 26,
             "run_profforc_16x16x8": 27, "run_vreman_buoycorr_16x8x12s": 53, "run_ibm_16x12x10": 55, "run_ibm_volflow_16x12x10": 56, "run_ibm_edge_16x12x10": 57,
             "run_ibm_thl_16x12x10": 59, "run_ibm_thlcons_16x12x10": 60, "run_ibm_qt_16x12x10": 61,
             "run_stats_16x8x12s": 62, "run_stats_ibm_16x12x10": 63,
             "run_floor_uno_nothl_16x8x12s": 65, "run_ibm_wf2_16x12x10": 68, "run_ibm_wh2_16x12x10": 70, "run_ibm_wh1_16x12x10": 72,
             "run_ground_wf3_16x8x12": 73, "run_ground_wh2_16x8x12": 74, "run_bcxs_16x8x12s": 76, "run_bcxs_avg_16x8x12s": 77, "run_ytstats_ibm_16x12x10": 78,
             "run_ibm_moist_16x12x10": 79, "run_ibm_moistwq_16x12x10": 80, "run_uoutflow_16x16x8": 82, "run_ibm_uoutflow_16x12x10": 83,
             "run_ptop_16x8x12s": 85, "run_ptop_ibm_16x12x10": 86, "run_ibmtall_16x12x10": 87, "run_bczp2_16x16x8": 89}      # the open lid (BCtopm = 3), the second with a block that reaches it
# decks with the facet wall functions (iwallmom > 1): on the device path and in the reference build; not in the C oracle's
# whole-substep driver (the numpy restatement covers the routine); the Fortran drop-in modibm builds the section tables itself
# (also: BCxs = 2, the scalars' inflow / outflow -- pinned device against reference fixture, not restated in the C oracle)
WF_RUN_CASES = {"run_bcxs_16x8x12s", "run_bcxs_avg_16x8x12s", "run_ibm_wf2_16x12x10", "run_ibm_wh2_16x12x10", "run_ibm_wh1_16x12x10", "run_ground_wf3_16x8x12", "run_ground_wh2_16x8x12",
                "run_ibm_moist_16x12x10", "run_ibm_moistwq_16x12x10", "run_uoutflow_16x16x8", "run_ibm_uoutflow_16x12x10"}


# A start-up transient of the reference: readinitfiles calls `thermodynamics` BEFORE program.f90:118's `boundary` has set the top
# ghost plane of thl0 (src/modstartup.f90:1601 vs src/program.f90:118), so the dthvdz that the FIRST closure after a cold start reads
# has, at level ke, `thl0(ke+1) = thl0(ke)` where every later substep sees the top condition's ghost value.  Only a deck with a value
# top for thl (BCtopT = 2) and a closure that reads dthvdz (lbuoycorr) notices: the eddy viscosity of the top level in one substep,
# 1e-6 .. 3e-5 of the fields afterwards.  The library reproduces it (udc_calthv, DynCore.start_up); the C oracle's whole-substep driver
# (tests only) starts from the dumped state s000 and does not -- its run-level tolerance for these decks:
STARTUP_TRANSIENT_TOL = {"run_vreman_buoycorr_16x8x12s": 1e-4}


def load_fixture(name):
    with gzip.open(os.path.join(GOLDEN, name + ".bin.gz"), "rb") as f:
        return read_dump(f.read())


def deck_path(name, iexp):
    return os.path.join(GOLDEN, "cases", name, f"namoptions.{iexp:03d}")


def marr(fix, key, nz):
    """Golden record -> full m-array [nz+2, ny+2, nx+2] (tendencies start at k=1 in the reference)."""
    f = fix[key]
    nk, nj, ni = f.data.shape
    a = np.zeros((nz + 2, nj, ni))
    lk = f.lb[2]
    a[lk:lk + nk] = f.data
    return a


def carr(fix, key, nz):
    """Golden record -> full c-array [nz+4, ny+4, nx+4]."""
    f = fix[key]
    nk, nj, ni = f.data.shape
    a = np.zeros((nz + 4, nj, ni))
    lk = f.lb[2] + 1
    a[lk:lk + nk] = f.data
    return a


def interior(a, h=1):
    return a[h:-h, h:-h, h:-h]


def relerr(a, b, scale=None):
    """max |a-b| / max |b| (field-scale relative error).  `scale` overrides the denominator
    where the field is a small difference of larger ones (pres0 = pres0_old + p)."""
    s = np.abs(b).max() if scale is None else scale
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def nocorner(a, h=1):
    """Copy with the xy-corner ghost columns zeroed.  The reference never fills those cells of
    p/pres0 (bcp only wraps interior rows/columns, src/modboundary.f90:1365-1408) and no stencil
    reads them; the device library has no x ghosts at all, so downloads carry periodic images."""
    b = a.copy()
    b[:, :h, :h] = 0; b[:, :h, -h:] = 0; b[:, -h:, :h] = 0; b[:, -h:, -h:] = 0
    return b


# ---- multi-rank launches: real RCCL wherever the box has a GPU per rank, the shared-memory test transport on a one-GPU box --------
REFDIR = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
BINDIR = os.path.join(os.path.dirname(HERE), "u-dales_amd", "bin")      # the drop-in programs (u-dales_amd/fortran/Makefile)
MPIEXEC = "/opt/conda/bin/mpiexec"


_SHM_SERIAL = [0]


def gpu_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:      # noqa: BLE001
        return 0


def mpi_transport(nranks, tag):
    """-> (executable, environment, label) for `mpiexec -n nranks` of the reference's program over the drop-in modules.

    A box with at least `nranks` GPUs runs the PRODUCT: u-dales_amd/bin/udales_full_dropin_mpi over libudcore.so, one rank per GPU,
    ghost rows / transposes / reductions through RCCL (udc_comm_init).  A one-GPU box runs the test build
    (udales_full_dropin_mpi_test over libudcore_test.so), every rank on device 0, the same exchanges through a shared-memory
    segment (UDC_TEST_SHM) -- which validates the harness and everything but RCCL's byte mover."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", UDC_RESIDENCY="2")
    if gpu_count() >= nranks:
        env.pop("UDC_GPUS_PER_NODE", None)
        env.pop("UDC_TEST_SHM", None)
        return os.path.join(BINDIR, "udales_full_dropin_mpi"), env, "rccl"
    # (a name per launch: a run that died before its ranks had all attached leaves its segment behind under its name)
    _SHM_SERIAL[0] += 1
    env.update(UDC_GPUS_PER_NODE="1", UDC_TEST_SHM=f"/udc_{tag}_{os.getpid()}_{nranks}_{_SHM_SERIAL[0]}")
    return os.path.join(BINDIR, "udales_full_dropin_mpi_test"), env, "shm"
