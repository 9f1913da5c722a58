"""The command-line runner (u-dales_amd/udcore/run.py): deck checks on the CPU, an end-to-end run on the GPU whose
restart files are compared with the reference's dumps."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from common import GOLDEN, RUN_CASES, deck_path, load_fixture, marr, nocorner, relerr
from udcore import read_deck
from udcore import restart as R
from udcore.run import check_supported, courant_default

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unsupported_decks_are_refused(tmp_path):
    src = os.path.join(GOLDEN, "cases", "run_16x16x8")
    for fn in os.listdir(src):
        shutil.copy(os.path.join(src, fn), tmp_path)
    path = os.path.join(tmp_path, "namoptions.021")
    check_supported(read_deck(path))                       # the fixture deck itself is fine
    txt = open(path).read()
    for bad in ("&BC\nBCzp = 3", "&BC\nBCxm = 4", "&BC\nBCym = 2"):      # (BCzp = 2, BCxm = 2 and 3 are served since round 6)
        grp = bad.split("\n")[0]
        with open(path, "w") as f:
            f.write(txt.replace(grp, bad, 1))
        with pytest.raises(SystemExit):
            check_supported(read_deck(path))
    # obstacles with the wall-function switches at their defaults and no temperature equation: checkinitvalues makes that the
    # neutral wall function before anything reads it (src/modstartup.f90:811-816), and so does the deck reader
    with open(path, "w") as f:
        f.write(txt.replace("libm = .false.", "libm = .true.").replace("&ORACLE", "&WALLS\nnfcts = 12\n/\n&ORACLE"))
    d = read_deck(path)
    check_supported(d)
    assert int(d.get("WALLS", "iwallmom")) == 3 and int(d.get("BC", "BCbotm")) == 3


def test_courant_default():
    d = read_deck(deck_path("run_16x16x8", 21))
    assert courant_default(d) == 1.5                       # cd2 only, src/modglobal.f90:565-567
    d = read_deck(deck_path("run_smag_scalar_16x8x12s", 22))
    assert courant_default(d) == 1.1                       # kappa scalars, :570-571


@pytest.mark.gpu
def test_cli_run_writes_the_reference_state(tmp_path):
    name, iexp = "run_16x16x8", RUN_CASES["run_16x16x8"]
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_case.py"), f"namoptions.{iexp:03d}", "--steps", "3"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "3 steps" in r.stdout
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    n = int(d.get("DOMAIN", "itot")), int(d.get("DOMAIN", "jtot")), int(d.get("DOMAIN", "ktot"))
    got = R.read_initd(os.path.join(tmp_path, R.restart_name(3, 0, iexp)), *n)
    assert abs(got["timee"] - 3 * float(d.get("RUN", "dtmax"))) < 1e-12
    for k in ("u0", "v0", "w0", "pres0"):                  # 3 steps = 9 substeps = the fixture's last dump
        ref = marr(fix, f"s009.{k}", n[2])
        assert relerr(nocorner(got[k][1:-1]), nocorner(ref[1:-1])) <= 1e-9, k


@pytest.mark.gpu
def test_adaptive_time_step_matches_reference():
    """tstep_update with ladaptive (src/modtstep.f90:49-154): dt = dtmax/100 at the cold start, then
    min(dtmax, dt courant / C, dt diffnr / D) from the device's maxima -- the dt history and the fields of a reference run."""
    import udcore
    from common import carr, interior, load_fixture, marr, nocorner, relerr
    from udcore import cold_start
    from udcore import lib as L
    name, iexp = "run_adaptive_16x8x12s", 46
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    assert d.get("RUN", "ladaptive")
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, nsv=core.nsv))
    dtmax = float(d.get("RUN", "dtmax"))
    courant, diffnr = courant_default(d), float(d.get("RUN", "diffnr"))
    assert courant == 1.1
    core.dt, core.timee, core.rk3step = dtmax / 100., 0., 0          # src/modstartup.f90:1099
    np.testing.assert_allclose(fix["s000.time"].data, [0., dtmax / 100.])
    nz = core.g.nz
    for isub in range(1, 19):
        rk, dt = core.tstep_update(dtmax, True, courant, diffnr)
        core.substep(rk, dt, with_forces=True)
        if isub in (3, 9, 18):
            tag = f"s{isub:03d}"
            np.testing.assert_allclose([core.timee, core.dt], fix[tag + ".time"].data, rtol=1e-10)
            for k in ("u0", "v0", "w0", "pres0"):
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(marr(fix, f"{tag}.{k}", nz)[1:-1])) <= 1e-9, (tag, k)
            got = core.download(L.scalar_field(L.SV0, 0), halo=2)
            assert relerr(interior(got, 2), interior(carr(fix, f"{tag}.sv0_01", nz), 2)) <= 1e-9
    assert 0.3 < core.dt < dtmax            # the limiter, not dtmax, set the step
    core.close()


@pytest.mark.gpu
def test_cli_run_writes_tdump_statistics(tmp_path):
    """&OUTPUT ltdump through the runner: the device accumulates (udc_stats_sample), the runner keeps the reference's sample /
    dump clocks and writes tdump.<expnr>.npz (and NetCDF-3 when scipy is there) with the reference's variable names."""
    name, iexp = "run_smag_scalar_16x8x12s", RUN_CASES["run_smag_scalar_16x8x12s"]
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    path = os.path.join(tmp_path, f"namoptions.{iexp:03d}")
    d = read_deck(path)
    dt = float(d.get("RUN", "dtmax"))
    with open(path) as f:
        txt = f.read()
    with open(path, "w") as f:
        f.write(txt.replace("&ORACLE", f"&OUTPUT\nltdump = .true.\ntsample = {dt}\ntstatsdump = {3 * dt}\n/\n&ORACLE"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_case.py"), f"namoptions.{iexp:03d}", "--steps", "7"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("tdump written") == 2
    z = np.load(os.path.join(tmp_path, f"tdump.{iexp:03d}.npz"))
    assert len(z["time"]) == 2 and abs(z["time"][0] - 3 * dt) < 1e-12
    nx, ny, nz = (int(d.get("DOMAIN", k)) for k in ("itot", "jtot", "ktot"))
    for k in ("ut", "vt", "wt", "pt", "upwpt", "tketc", "sca1t", "wpsca1pt", "sca1psca1pt", "sv1sgs"):
        assert z[f"{k}.0"].shape == (nz, ny, nx), k
    assert 0.8 < z["ut.1"].mean() < 1.2 and z["tketc.1"].min() > -1e-12 and z["tketc.1"].max() > 0.
    if os.path.exists(os.path.join(tmp_path, f"tdump.{iexp:03d}.nc")):
        from scipy.io import netcdf_file
        with netcdf_file(os.path.join(tmp_path, f"tdump.{iexp:03d}.nc"), "r", mmap=False) as f:
            assert f.variables["ut"].shape == (2, nz, ny, nx)
            assert np.abs(f.variables["ut"][1] - z["ut.1"]).max() < 1e-6
