"""The reference's OWN tests of this path, run here (SURVEY.md section 4 lists them; there are no others that touch the solver).

1. src/tests.f90's operator test -- `runmode = 1005` in &RUN (program.f90:238-274 dispatches it): avexy_ibm / avey_ibm / sumx_ibm /
   sumy_ibm on analytic fields over the masks of tests/cases/100, against a brute-force MPI_ALLREDUCE, tolerance 1e-9
   (src/tests.f90:188-431); driven by tests/integration/mpi_operators/run_test.sh with the decks namoptions.1005.{serial,xsplit,
   ysplit,xysplit}.  Here: through oracle/_ref/udales_full (the reference's whole src/ tree, one rank) and udales_full_mpi (2 and 4
   ranks over MPICH and the y-slab decomposition stand-in: `ysplit`; nprocx > 1 is not something the stand-in does) -- which puts
   the MPI stand-ins under the reference's own test -- and, on a GPU box, through udales_full_dropin: the same unmodified
   program.f90 / tests.f90 over the ten drop-in modules, i.e. the masks of the drop-in createmasks, its initibm, its start-up.
   The device's own masked averages get the same treatment: the test's analytic fields and tolerance on udc_stats_xy / udc_stats_y.

2. tests/integration/processor_boundaries/test_processor_boundaries.py -- "the only solver-numerics pin" of the reference: case 100
   (128^3, the Xie / Castro cube array) run for one step (namoptions.100.serial) on 1 x 1, 2 x 1, 1 x 2, 2 x 2 ranks, `ut, vt, wt` of
   tdump compared: <= 1e-9 on the bands next to the rank boundaries, <= 2e-8 everywhere (their files are float32).  Here the
   candidates are the device runs -- the reference's program with the drop-in modules on one rank and on 2 / 4 (/ 8) MPI ranks (y split;
   over RCCL where the box has a GPU per rank, else sharing the one GPU through the test library's transport) -- and the reference is the ALL-REFERENCE executable on the
   box's host: <= 1e-9 EVERYWHERE for all three fields (the recording NetCDF stand-in keeps float64, so no float32 allowance is
   needed), i.e. their decomposition-invariance test and a parity test against the reference in one.

(runmode 1004, the sparse point-list reader, compares against rank-local golden files of a 2 x 2 decomposition,
tests/integration/ibm_sparse_input/*_X_Y.txt: not runnable over a y-slab stand-in; runmode 1003 only prints pencil extents.)"""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

from common import GOLDEN, gpu_count, mpi_transport
from refdump import read_ncrec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
FULL = os.path.join(REFDIR, "udales_full")
FULL_MPI = os.path.join(REFDIR, "udales_full_mpi")
BINDIR = os.path.join(ROOT, "u-dales_amd", "bin")
DROPIN = os.path.join(BINDIR, "udales_full_dropin")
DROPIN_MPI_TEST = os.path.join(BINDIR, "udales_full_dropin_mpi_test")
MPIEXEC = "/opt/conda/bin/mpiexec"
TOL = 1.0e-9            # ABS_TOL of test_processor_boundaries.py:28 and of src/tests.f90:389


def stage(tmp, deck, nprocy=1, steps=None, nprocx=1, case=100):
    """tests/cases/100 (or 526) + one of the test drivers' decks as namoptions.100 (what run_test.sh / _copy_namelist do)."""
    os.makedirs(tmp, exist_ok=True)
    cdir = os.path.join(GOLDEN, "cases", f"case_{case}")
    for fn in os.listdir(cdir):
        if fn.startswith("namoptions"):
            continue
        with gzip.open(os.path.join(cdir, fn), "rb") as f, open(os.path.join(tmp, fn[:-3]), "wb") as o:
            o.write(f.read())
    with gzip.open(os.path.join(cdir, deck + ".gz"), "rt") as f:
        txt = f.read()
    txt = re.sub(r"nprocy\s*=\s*\d+", f"nprocy       = {nprocy}", txt)
    assert re.search(r"nprocx\s*=\s*1\b", txt)
    txt = re.sub(r"nprocx\s*=\s*1\b", f"nprocx       = {nprocx}", txt)
    if steps is not None:      # a longer variant of the one-step deck: `steps` steps of dtmax, one tdump record at the end
        dtmax = float(re.search(r"dtmax\s*=\s*([0-9.eE+-]+)", txt).group(1))
        txt = re.sub(r"runtime\s*=\s*[0-9.eE+-]+", f"runtime      = {dtmax * (steps - 0.5)!r}", txt)
        # (half a step early: the dump clock is a running sum of dt, which may fall a rounding short of the product dtmax * steps)
        txt = re.sub(r"tstatsdump\s*=\s*[0-9.eE+-]+", f"tstatsdump   = {dtmax * (steps - 0.5)!r}", txt)
        txt = re.sub(r"tsample\s*=\s*[0-9.eE+-]+", f"tsample      = {dtmax!r}", txt)
    with open(os.path.join(tmp, f"namoptions.{case}"), "w") as f:
        f.write(txt)


def run(tmp, exe, nranks=1, env=None, timeout=1500, case=100):
    cmd = f"{exe} namoptions.{case}" if nranks == 1 and "mpi" not in os.path.basename(exe) else f"{MPIEXEC} -n {nranks} {exe} namoptions.{case}"
    r = subprocess.run(f"ulimit -s unlimited; exec {cmd}", shell=True, cwd=tmp, env=env, capture_output=True, text=True, timeout=timeout,
                       executable="/bin/bash")
    return r


# ---- 1. runmode 1005 ------------------------------------------------------------------------------------------------------------

def test_operator_test_of_the_reference_on_one_rank(tmp_path):
    if not os.path.exists(FULL):
        pytest.skip("oracle/_ref/udales_full not built")
    stage(tmp_path, "namoptions.1005.serial")
    r = run(tmp_path, FULL)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]          # program.f90:268-272: stop 1 on failure
    assert "ALL TESTS PASSED: tests_mpi_operators" in r.stdout and "FAIL" not in r.stdout


@pytest.mark.parametrize("P", [2, 4])
def test_operator_test_of_the_reference_over_the_mpi_stand_in(P, tmp_path):
    if not (os.path.exists(FULL_MPI) and os.path.exists(MPIEXEC)):
        pytest.skip("oracle/_ref/udales_full_mpi or MPICH not available")
    stage(tmp_path, "namoptions.1005.ysplit", nprocy=P)
    r = run(tmp_path, FULL_MPI, P)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert "ALL TESTS PASSED: tests_mpi_operators" in r.stdout and "FAIL" not in r.stdout


@pytest.mark.parametrize("P", [1, 2])
def test_pencil_extents_of_the_decomposition_stand_in(P, tmp_path):
    """runmode 1003 (src/tests.f90:30-41, tests_2decomp_init_exit): the reference prints 2DECOMP's xstart .. zsize after its own
    start-up.  Over the stand-in: one rank owns 1..128 in every direction; with nprocy = 2 every pencil of rank r holds rows
    64 r + 1 .. 64 r + 64 (the stand-in keeps y-slabs in all three orientations)."""
    exe = FULL if P == 1 else FULL_MPI
    if not os.path.exists(exe) or (P > 1 and not os.path.exists(MPIEXEC)):
        pytest.skip("oracle/_ref/udales_full(_mpi) or MPICH not available")
    stage(tmp_path, "namoptions.1005.serial", nprocy=P)
    deck = tmp_path / "namoptions.100"
    deck.write_text(re.sub(r"runmode\s*=\s*1005", "runmode      = 1003", deck.read_text()))
    r = run(tmp_path, exe, P)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    rows = [tuple(int(v) for v in ln.split()) for ln in r.stdout.splitlines() if re.fullmatch(r"\s*\d+\s+\d+\s+\d+\s*", ln)]
    assert len(rows) == 9 * P, r.stdout[-1500:]
    if P == 1:
        assert rows == [(1, 1, 1)] * 3 + [(128, 128, 128)] * 6
    else:
        starts = sorted(set(rows[q] for q in range(len(rows)) if rows[q][0] == 1 and rows[q][2] == 1 and rows[q][1] in (1, 65)))
        assert starts == [(1, 1, 1), (1, 65, 1)], rows
        assert (128, 64, 128) in rows and (128, 128, 128) in rows      # sizes and the upper rank's ends


@pytest.mark.gpu
def test_operator_test_of_the_reference_over_the_dropin_modules(tmp_path):
    if not os.path.exists(DROPIN):
        pytest.skip("u-dales_amd/bin/udales_full_dropin not built")
    stage(tmp_path, "namoptions.1005.serial")
    r = run(tmp_path, DROPIN, env=dict(os.environ, UDC_RESIDENCY="2"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert "ALL TESTS PASSED: tests_mpi_operators" in r.stdout and "FAIL" not in r.stdout


def analytic(loc_id, nx, ny, nz, c):
    """var_clean of src/tests.f90:287 (xy) / :334 (y); i, j, k the Fortran indices ib.., jb.., kb.. = 1.."""
    k, j, i = np.meshgrid(np.arange(1, nz + 1), np.arange(1, ny + 1), np.arange(1, nx + 1), indexing="ij")
    return c[0] * loc_id + c[1] * i + c[2] * j + c[3] * k


@pytest.mark.gpu
def test_masked_averages_of_the_device_pass_the_operator_test():
    """check_loc_xy / check_loc_y of src/tests.f90 put to the device's masked averages: the same analytic fields, one per grid
    location (C = 1 on thl, U = 2, V = 3, W = 4), over case 100's masks, udc_stats_sample -> udc_stats_xy (avexy_ibm with IIu, IIv,
    IIw, IIc) and udc_stats_y (avey_ibm), against the brute-force sums, 1e-9 absolute; -999 where a level / column has no fluid."""
    import tempfile
    import udcore
    from udcore import read_deck
    from udcore.ibm import read_ibm
    from udcore.stats import TDump, xyt_masks
    with tempfile.TemporaryDirectory() as tmp:
        stage(tmp, "namoptions.100.serial")
        with open(os.path.join(tmp, "namoptions.100")) as f:
            txt = f.read()
        with open(os.path.join(tmp, "namoptions.100"), "w") as f:      # thl carries the C-location field
            f.write(txt + "\n&PHYSICS\nltempeq = .true.\n/\n")
        d = read_deck(os.path.join(tmp, "namoptions.100"))
        lists = read_ibm(d)
        core = udcore.from_deck(d)
    g = core.g
    nx, ny, nz = g.nx, g.ny, g.nz
    assert (nx, ny, nz) == (128, 128, 128)
    cxy, cy = (0.25, 0.13, -0.07, 0.011), (0.5, 0.21, -0.03, 0.017)
    bits, counts, forced = xyt_masks(nx, ny, nz, lists)
    II = {q: ((bits >> b) & 1).astype(float) for b, q in enumerate("uvwc")}
    for b, q in enumerate("uvwc"):
        if forced[b]:
            II[q][0] = 0.      # createmasks' own mask (IIw(:, :, kb) = 0); xyt_masks fills such a level for avexy_ibm's lnan = .false. rule
    for coef, which in ((cxy, "xy"), (cy, "y")):
        fld = {"u": analytic(2, nx, ny, nz, coef), "v": analytic(3, nx, ny, nz, coef), "w": analytic(4, nx, ny, nz, coef),
               "c": analytic(1, nx, ny, nz, coef)}
        for name, q in (("um", "u"), ("vm", "v"), ("wm", "w"), ("thlm", "c"), ("u0", "u"), ("v0", "v"), ("w0", "w"), ("thl0", "c")):
            core.upload(name, np.pad(fld[q], 1, mode="edge"))
        td = TDump(core, 1., 1e9, xyt=False, ibm_lists=lists, xy=(which == "xy"), y=(which == "y"))
        td.step(3, 1., 1.)
        assert td.nsamples == 1
        if which == "xy":
            got = td.xy()
            for row, q, b in (("uxy", "u", 0), ("vxy", "v", 1), ("wxy", "w", 2), ("thlxy", "c", 3)):
                s = (fld[q] * II[q]).sum(axis=(1, 2))
                n = II[q].sum(axis=(1, 2))
                exp = np.where(n > 0, s / np.maximum(n, 1), -999.)
                lev = np.ones(nz, dtype=bool)
                if forced[b]:
                    lev[0] = False      # (avexy_ibm's lnan = .false. rule for a first level without fluid: not what tests.f90 checks)
                assert np.abs(got[row] - exp)[lev].max() <= TOL, (row, np.abs(got[row] - exp)[lev].max())
        else:
            got = td.y()
            for row, q in (("uy", "u"), ("vy", "v"), ("wy", "w"), ("thly", "c")):
                s = (fld[q] * II[q]).sum(axis=1)
                n = II[q].sum(axis=1)
                exp = np.where(n > 0, s / np.maximum(n, 1), -999.)
                assert np.abs(got[row] - exp).max() <= TOL, (row, np.abs(got[row] - exp).max())
    core.close()


# ---- 2. processor boundaries, case 100 ------------------------------------------------------------------------------------------

def tdump_fields(tmp, nranks, case=100, prefix="tdump", want=("ut", "vt", "wt")):
    """ut, vt, wt of tdump.000.RRR.100.nc stitched over the ranks' rows -> {name: [k, j, i]} (the reference's _load_global_fields)."""
    parts = []
    for r in range(nranks):
        rec = read_ncrec(os.path.join(tmp, f"{prefix}.000.{r:03d}.{case}.nc"), want=want)
        parts.append({k: np.squeeze(np.asarray(v[-1][1], dtype=float)) for k, v in rec.items()})
    return {k: np.concatenate([p[k] for p in parts], axis=1) for k in want}


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [None, 10])
def test_processor_boundaries_case_100(steps, tmp_path):
    """steps = None: the reference's deck as it is (one step).  10: the same deck run for 10 steps with a sample every step and the
    one tdump record at the end (not a test the reference holds: the same comparison on a run long enough for the wall functions,
    the immersed boundary and the pressure solver to have acted on each other's output)."""
    if not (os.path.exists(FULL) and os.path.exists(DROPIN)):
        pytest.skip("oracle/_ref/udales_full(_dropin) not built")
    out = {}
    stage(tmp_path / "ref", "namoptions.100.serial", steps=steps)
    r = run(tmp_path / "ref", FULL)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out["reference, serial"] = tdump_fields(tmp_path / "ref", 1)
    stage(tmp_path / "dev1", "namoptions.100.serial", steps=steps)
    r = run(tmp_path / "dev1", DROPIN, env=dict(os.environ, UDC_RESIDENCY="2"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out["device, serial"] = tdump_fields(tmp_path / "dev1", 1)
    if os.path.exists(MPIEXEC):
        # RCCL where the box has a GPU per rank (2, 4, 8 ranks of the product executable), the shared-memory test transport on one GPU
        for P in ((2, 4) if gpu_count() < 2 else tuple(q for q in (2, 4, 8) if q <= gpu_count())):
            exe, env, how = mpi_transport(P, "c100")
            if not os.path.exists(exe):
                continue
            d = tmp_path / f"dev{P}"
            stage(d, "namoptions.100.serial", nprocy=P, steps=steps)
            r = run(d, exe, P, env=env)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
            out[f"device, y split over {P} ({how})"] = tdump_fields(d, P)
        # the reference's own x-split and x-y-split variants (2 x 1 and 2 x 2 pencils, test_processor_boundaries.py:28-34): the decks
        # run UNEDITED in their process grid -- u-dales_amd/fortran/decomp_2d.f90 hands the pencils out as 2 / 4 y-slabs
        for px, py in ((2, 1), (2, 2)):
            P = px * py
            if 1 < gpu_count() < P:
                continue
            exe, env, how = mpi_transport(P, f"c100x{px}{py}")
            if not os.path.exists(exe):
                continue
            d = tmp_path / f"dev{px}x{py}"
            stage(d, "namoptions.100.serial", nprocy=py, steps=steps, nprocx=px)
            r = run(d, exe, P, env=env)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
            assert f"the deck asks for {px} x {py} pencils" in r.stdout, r.stdout[-1500:]
            out[f"device, deck {px} x {py} ({how})"] = tdump_fields(d, P)
    ref = out["reference, serial"]
    assert ref["ut"].shape == (128, 128, 128) and np.abs(ref["ut"]).max() > 1.      # (u0 = 3 m/s + noise of amplitude randu = 1)
    worst = {}
    for label, cand in out.items():
        for against in ("reference, serial", "device, serial"):
            if label == against or (against == "device, serial" and label == "reference, serial"):
                continue
            for k in ("ut", "vt", "wt"):
                worst[(label, against, k)] = float(np.abs(cand[k] - out[against][k]).max())
    if os.environ.get("UDC_TEST_KEEP_LOGS"):
        with open(os.path.join(os.environ["UDC_TEST_KEEP_LOGS"], f"processor_boundaries_case100_{steps or 1}steps.txt"), "w") as f:
            f.write(f"max |candidate - against| of tdump's ut, vt, wt after {steps or 1} step(s) of namoptions.100.serial (tolerance 1e-9)\n")
            for (label, against, k), v in worst.items():
                f.write(f"{label:28s} vs {against:18s} {k}: {v:.3e}\n")
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, bad
    assert len(worst) >= 3


# ---- 2b. processor boundaries, case 526: trees, temperature + moisture, the open lid --------------------------------------------
DROPIN_HS = os.path.join(BINDIR, "udales_full_dropin_hoststats")
DROPIN_HS_MPI_TEST = os.path.join(BINDIR, "udales_full_dropin_hoststats_mpi_test")


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [None, 6])
def test_processor_boundaries_case_526(steps, tmp_path):
    """The other half of test_processor_boundaries.py (:43-50, TREE_CASE_ID): tests/cases/526 -- trees (the reference's vegetation.f90,
    untouched, on the host: the drop-ins fall back to the strict residency for it) over a floor that is an immersed boundary with
    facet wall functions, temperature + moisture + buoyancy, the adaptive time step and BCtopm = 3, the lid open to the pressure
    gradient (bcpup / tderive / tstep_integrate's row w(ke+1): k_lid_*, udc_pois.hip) -- one step of namoptions.526.serial; `tr_u,
    tr_v, tr_w` of treedump (the reference's own modstatsdump linked: the device statistics do not take the tree dump over) and
    `ut, vt, wt` of tdump, serial and split over 2 / 4 ranks in y and as 2 x 1 / 2 x 2 decks, against the ALL-REFERENCE executable:
    <= 1e-9 everywhere (the reference's ABS_TOL on its support masks and processor-boundary bands).
    steps = 6 (not a test the reference holds): the same deck for six steps with a sample every step, so that the trees' drag, the wall
    functions, the moist thermodynamics and the pressure solver have acted on each other's output; serial and one 2 x 2 deck."""
    if not (os.path.exists(FULL) and os.path.exists(DROPIN_HS)):
        pytest.skip("oracle/_ref/udales_full or u-dales_amd/bin/udales_full_dropin_hoststats not built")
    want_tr, want_t = ("tr_u", "tr_v", "tr_w"), ("ut", "vt", "wt")

    def fields(d, P):
        out = tdump_fields(d, P, 526, "treedump", want_tr)
        out.update(tdump_fields(d, P, 526, "tdump", want_t))
        return out
    out = {}
    stage(tmp_path / "ref", "namoptions.526.serial", case=526, steps=steps)
    r = run(tmp_path / "ref", FULL, case=526)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out["reference, serial"] = fields(tmp_path / "ref", 1)
    stage(tmp_path / "dev1", "namoptions.526.serial", case=526, steps=steps)
    r = run(tmp_path / "dev1", DROPIN_HS, env=dict(os.environ, UDC_RESIDENCY="2"), case=526)      # (asks for 2, gets 0: the trees)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "UDC_RESIDENCY=0" in r.stdout
    out["device, serial"] = fields(tmp_path / "dev1", 1)
    if os.path.exists(MPIEXEC) and os.path.exists(DROPIN_HS_MPI_TEST) and gpu_count() < 2:
        for px, py in (((1, 2), (1, 4), (2, 1), (2, 2)) if steps is None else ((2, 2),)):
            P = px * py
            _, env, how = mpi_transport(P, f"c526x{px}{py}")
            d = tmp_path / f"dev{px}x{py}"
            stage(d, "namoptions.526.serial", nprocy=py, nprocx=px, case=526, steps=steps)
            r = run(d, DROPIN_HS_MPI_TEST, P, env=env, case=526)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
            out[f"device, deck {px} x {py} ({how})"] = fields(d, P)
    ref = out["reference, serial"]
    assert ref["tr_u"].shape == (64, 64, 128) and np.abs(ref["tr_u"]).max() > 0.1      # (the drag of the trees: O(0.5) m/s^2)
    worst = {}
    for label, cand in out.items():
        if label == "reference, serial":
            continue
        for k in want_tr + want_t:
            worst[(label, k)] = float(np.abs(cand[k] - ref[k]).max())
    if os.environ.get("UDC_TEST_KEEP_LOGS"):
        with open(os.path.join(os.environ["UDC_TEST_KEEP_LOGS"], f"processor_boundaries_case526_{steps or 1}steps.txt"), "w") as f:
            f.write(f"max |candidate - all-reference serial run| of treedump's tr_u, tr_v, tr_w and tdump's ut, vt, wt after {steps or 1} step(s) of namoptions.526.serial (tolerance 1e-9)\n")
            for (label, k), v in worst.items():
                f.write(f"{label:36s} {k}: {v:.3e}\n")
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, bad


# ---- 2c. case 526 under the boundary conditions of case 525 ------------------------------------------------------------------------
def stage_526_open(tmp, steps):
    """tests/cases/526 with the inflow of tests/cases/525: BCxm = 3 from the planes of a precursor run (tests/golden/make_golden.py,
    make_case_526_open: the reference itself on a periodic channel derived from 526's deck), temperature and moisture left periodic in x
    as both decks have them.  525 itself cannot run through anybody's solver: the reference ships it without facet_sections_u / _v / _c.txt
    (it is the pre-processing case of tests/integration/directshortwave; src/modibm.f90:352 stops on the missing file)."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import c526_open_decks
    stage(tmp, "namoptions.526.serial", case=526, steps=steps)
    deck = os.path.join(tmp, "namoptions.526")
    with open(deck) as f:
        txt = c526_open_decks(f.read())[0]
    if steps is not None:      # one restart file near the end: the state with the outlet's columns
        dtmax = float(re.search(r"dtmax\s*=\s*([0-9.eE+-]+)", txt).group(1))
        txt = re.sub(r"trestart\s*=\s*[0-9.eE+-]+", f"trestart     = {dtmax * (steps - 1.5)!r}", txt)
    with open(deck, "w") as f:
        f.write(txt)
    cdir = os.path.join(GOLDEN, "cases", "case_526_open")
    for fn in os.listdir(cdir):
        with gzip.open(os.path.join(cdir, fn), "rb") as f, open(os.path.join(tmp, fn[:-3]), "wb") as o:
            o.write(f.read())


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [None, 4])
def test_case_526_with_the_inflow_of_case_525(steps, tmp_path):
    """Trees (the reference's vegetation.f90 on the host), a floor that is an immersed boundary with facet wall functions, temperature +
    moisture + buoyancy, the open lid, the adaptive step -- and the flow entering from a precursor's planes and leaving through a convective
    outlet (BCxm = 3) while thl and qt stay periodic (BCxT = BCxq = 1): what tests/cases/525 describes, on the files of 526.  Through
    udales_full_dropin_hoststats against the all-reference executable: treedump's tr_u, tr_v, tr_w, tdump's ut, vt, wt, and the restart
    file's u0, v0, w0, thl0, qt0 with the outlet's columns, <= 1e-9 (the reference's ABS_TOL)."""
    if not (os.path.exists(FULL) and os.path.exists(DROPIN_HS)):
        pytest.skip("oracle/_ref/udales_full or u-dales_amd/bin/udales_full_dropin_hoststats not built")
    sys_path = os.path.join(ROOT, "u-dales_amd")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from udcore import restart
    want_tr, want_t = ("tr_u", "tr_v", "tr_w"), ("ut", "vt", "wt")
    out = {}
    for label, exe, env in (("ref", FULL, None), ("dev", DROPIN_HS, dict(os.environ, UDC_RESIDENCY="2"))):
        d = tmp_path / label
        stage_526_open(d, steps)
        r = run(d, exe, env=env, case=526)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        o = tdump_fields(d, 1, 526, "treedump", want_tr)
        o.update(tdump_fields(d, 1, 526, "tdump", want_t))
        out[label] = o
        if steps is not None:      # (trestart shortened to the run's end: the restart file with the x ghost columns)
            rst = sorted(f for f in os.listdir(d) if f.startswith("initd") and f.endswith(".526"))
            assert rst, sorted(os.listdir(d))
            rs = restart.read_initd(os.path.join(d, rst[-1]), 128, 64, 64)
            o["rst.time"] = np.array([rs["timee"], rs["dt"], float(len(rst))])
            for k in ("u0", "v0", "w0", "thl0", "qt0"):
                o["rst." + k] = rs[k][1:65, 1:65, :]
        if label == "dev":
            assert "UDC_RESIDENCY=0" in r.stdout      # (asks for 2, gets 0: the trees)
    ref, dev = out["ref"], out["dev"]
    assert np.abs(ref["tr_u"]).max() > 0.1
    # the flow is not periodic: what leaves differs from what enters
    assert np.abs(ref["ut"][:, :, 0] - ref["ut"][:, :, -1]).max() > 1e-3
    worst = {k: float(np.abs(dev[k] - ref[k]).max()) for k in ref}
    if os.environ.get("UDC_TEST_KEEP_LOGS"):
        with open(os.path.join(os.environ["UDC_TEST_KEEP_LOGS"], f"case526_with_inflow_of_525_{steps or 1}steps.txt"), "w") as f:
            f.write(f"max |udales_full_dropin_hoststats - all-reference run| after {steps or 1} step(s) of tests/cases/526 with BCxm = 3 (tolerance 1e-9)\n")
            for k, v in worst.items():
                f.write(f"{k:10s} {v:.3e}\n")
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, bad
