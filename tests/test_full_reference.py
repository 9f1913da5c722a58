"""The pin under every run-level fixture: the reference's OWN EXECUTABLE.

oracle/_ref/udales_full is `u-dales` as the reference's CMakeLists would link it -- every file of /root/reference/src, unmodified,
program.f90 and modstartup.f90 included (oracle/Makefile) -- over stand-ins for the third-party layers this image lacks (MPI,
2DECOMP&FFT, NetCDF: u-dales_amd/fortran/standins; FFTW: oracle/shims).  The fixtures tests/golden/run_*.bin.gz come from oracle/_ref/udales_ref, the same
objects under a second main program that can dump between routines (oracle/ref_driver.f90).  Here every run deck goes through the
real program -- `udales_full namoptions.NNN`, nothing else on the command line -- with a run time and a restart interval that make it
stop, and write its restart files (the reference's own writerestartfiles, real(8)), where the fixture's last dump was taken: the two
must agree BIT FOR BIT.  Whatever the second main program restates (the list of calls of src/program.f90:63-222) is thereby checked
against the first on every deck, and with it the decks themselves (every namelist group read by the reference's own reader).

Also: xytdump's table as the reference's statsdump hands it to NetCDF (recorded by the stand-in) against the table the driver
assembles from the same accumulators."""
import glob
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from common import GOLDEN, RUN_CASES, load_fixture
from refdump import read_ncrec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "oracle", "_ref", "udales_full")


def run_full(name, iexp, tmp_path, exe=FULL, env=None, deck_text=None):
    """-> (fixture, tag of its last dump, restart dict of the real program stopped at that time, working directory)"""
    from udcore import restart
    fix = load_fixture(name)
    nx, ny, nz = (int(v) for v in fix["meta"].data[:3])
    tags = sorted(set(k.split(".")[0] for k in fix if re.match(r"s\d\d\d\.", k)))
    last = tags[-1]
    nsub = int(last[1:])
    assert nsub % 3 == 0
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    deck = os.path.join(tmp_path, f"namoptions.{iexp:03d}")
    with open(deck) as f:
        txt = f.read()
    if deck_text is not None:      # (a variant of the case's deck)
        txt = deck_text
    dtmax = float(re.search(r"dtmax\s*=\s*([0-9.eE+-]+)", txt).group(1))
    tend = fix[last + ".time"].data[0] if last + ".time" in fix else dtmax * nsub / 3
    runtime = tend * (1. - 1e-9)       # the loop runs while timeleft > 0 (src/program.f90:132): stops after the step that reaches tend
    txt = re.sub(r"runtime\s*=\s*[0-9.eE+-]+", f"runtime = {runtime!r}\ntrestart = {0.999 * runtime!r}", txt)
    with open(deck, "w") as f:
        f.write(txt)
    r = subprocess.run(f"ulimit -s unlimited; exec {exe} namoptions.{iexp:03d}", shell=True, cwd=tmp_path, capture_output=True,
                       text=True, timeout=600, executable="/bin/bash", env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    files = glob.glob(os.path.join(tmp_path, "initd*"))
    assert len(files) == 1, (files, r.stdout[-800:])
    rs = restart.read_initd(files[0], nx, ny, nz)
    nsv = int(fix["meta"].data[12])
    if nsv:
        rs.update(restart.read_inits(files[0].replace("initd", "inits"), nx, ny, nz, nsv))
    return fix, last, rs, tmp_path


# (+ the inflow / outflow decks of tests/test_gpu_open_x.py: fixtures the oracle does not restate are pinned on the program too)
OPEN_X_CASES = {"run_xopen_16x8x12s": 91, "run_xopen_vr_24x8x10": 92, "run_xopen_ibm_16x12x10": 93, "run_xopen_ibmwf3_16x12x10": 94,
                "run_xopen_volflow_16x8x12s": 99, "run_xopen_thl_16x8x12s": 101, "run_xopen_sv_16x8x12s": 103, "run_xopen_ibm_sv_16x12x10": 104, "run_xopen_ibm_thl_16x12x10": 105, "run_xdriver_16x8x12s": 96, "run_xdriver_ibm_16x12x10": 98,
                "run_xopen_moist_16x8x12s": 107, "run_xopen_ibm_moist_16x12x10": 108, "run_xdriver_moist_16x12x10": 110,
                "run_xopen_qt2_16x8x12s": 111, "run_xdriver_scal_16x8x12s": 113}


@pytest.mark.parametrize("name,iexp", sorted({**RUN_CASES, **OPEN_X_CASES}.items()))
def test_fixture_equals_the_reference_executable(name, iexp, tmp_path):
    if not os.path.exists(FULL):
        pytest.skip("oracle/_ref/udales_full not built (needs the reference sources + flang)")
    fix, last, rs, _ = run_full(name, iexp, tmp_path)
    nz = int(fix["meta"].data[2])
    if last + ".time" in fix:
        assert (rs["timee"], rs["dt"]) == tuple(fix[last + ".time"].data)
    checked = 0
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0", "e120"):
        key = f"{last}.{k}"
        if key not in fix:
            continue
        a, b = fix[key].data, rs[k]
        if name.startswith("run_xdriver"):
            # (inflow from driver files: the program, stopped by `runtime` at the dump's time, skips the last drivergen -- "if (timee >
            #  runtime + btime) return", src/moddriver.f90:216 -- which the fixture's driver, with its long runtime, makes: the inlet's two
            #  columns ib-1, ib of the LAST `boundary` differ, nothing else has seen them yet)
            a, b = a[:, :, 2:], b[:, :, 2:]
        assert np.array_equal(a[1:nz + 2], b[1:nz + 2]), key          # (the file holds kb : ke + kh, ghost columns and rows included)
        checked += 1
    for n in range(int(fix["meta"].data[12])):
        a = fix[f"{last}.sv0_{n + 1:02d}"].data[2:nz + 3, 1:-1, 1:-1]      # the fixture keeps two ghost cells, the file one
        b = rs["sv0"][n][1:nz + 2]
        if name.startswith("run_xdriver"):      # (BCxs = 3: the inlet's column of the last `boundary`, as above)
            a, b = a[:, :, 1:], b[:, :, 1:]
        assert np.array_equal(a, b), n
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("name,iexp", [("run_stats_16x8x12s", 62), ("run_stats_ibm_16x12x10", 63)])
def test_xytdump_table_as_handed_to_netcdf(name, iexp, tmp_path):
    """What the reference's statsdump writes into xytdump.NNN.nc (float64, before NetCDF's conversion to float; recorded by the
    NetCDF stand-in) is the table the fixtures hold under `xyt.*` -- there assembled by the driver from the same accumulators with the
    reference's avexy_ibm, because the table is local to statsdump."""
    if not os.path.exists(FULL):
        pytest.skip("oracle/_ref/udales_full not built")
    fix, last, rs, wd = run_full(name, iexp, tmp_path)
    rec = read_ncrec(os.path.join(wd, f"xytdump.{iexp:03d}.nc"))
    nz = int(fix["meta"].data[2])
    ncname = {"uwtxyik": "uwxyt", "vwtxyjk": "vwxyt", "wwtxyk": "wwxyt", "uvtxyij": "uvxyt", "upwptxyik": "upwpxyt", "vpwptxyjk": "vpwpxyt",
              "upvptxyij": "upvpxyt", "wthltxyk": "wthlxyt", "wpthlptxyk": "wpthlpxyt"}
    if "time" not in rec:
        pytest.skip("the run ends before the first statistics dump is due")
    checked = 0
    for k, f in fix.items():
        if not k.startswith("xyt."):
            continue
        nm = ncname.get(k[4:], k[4:])
        if nm not in rec:
            continue
        got = rec[nm][-1][1]
        assert np.array_equal(got[:nz], f.data[:nz]), k
        checked += 1
    assert checked >= 10
