"""The drop-in boundary, exercised from Fortran: the reference-shaped driver (oracle/ref_driver.f90,
call order of src/program.f90) linked with u-dales_amd/fortran/{modadvection,modsubgrid,modpois,
modtstep,modforces,modboundary,modthermodynamics,modibm}.f90 -- same module and procedure names as the
reference -- instead of the reference's own modules.  Its dumps must match the golden dumps the
all-reference build produced from the same namoptions (tests/golden).  The state modules (modglobal,
modfields, modsurfdata ...), modsave and modscalsource in the binary are the reference's unmodified
code.  Residency 0 / 1: every routine carries the state / the tendencies over PCIe; residency 2: device
resident, the routines record and tstep_integrate launches the fused substep.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from common import GOLDEN, KERNEL_CASES, RUN_CASES, load_fixture, nocorner, relerr
from refdump import read_dump

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "udales_dropin")


def run_dropin(name, iexp, mode, tmp_path, residency):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/udales_dropin not built (needs the reference sources + flang)")
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    # (UDC_EK_ALWAYS: this test driver also dumps ekm / ekh after RK stages 1 and 2, where nothing of the reference's own loop looks
    # at them and a deck without scalars does not write ekh; the real program -- tests/test_gpu_full_dropin.py -- runs without it)
    env = dict(os.environ, UDC_RESIDENCY=str(residency), UDC_EK_ALWAYS="1")
    r = subprocess.run(f"ulimit -s unlimited; exec {BIN} namoptions.{iexp:03d} {mode} out.bin", shell=True,
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300, executable="/bin/bash")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return read_dump(os.path.join(tmp_path, "out.bin"))


@pytest.mark.parametrize("residency", [0, 1, 2])
@pytest.mark.parametrize("name,iexp", sorted(RUN_CASES.items()))
def test_fortran_driver_with_dropin_modules(name, iexp, residency, tmp_path):
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "run", tmp_path, residency)
    checked = 0
    for key, ref in fix.items():
        if "." not in key or key.startswith("s000.") or key.split(".")[1] not in ("u0", "v0", "w0", "pres0", "thl0", "qt0"):
            continue
        a, b = got[key].data[1:-1], ref.data[1:-1]
        sc = 1.0 if key.endswith("thl0") else None          # O(1) K variations on a 288 K mean
        assert relerr(nocorner(a), nocorner(b), sc) <= 1e-9, key
        checked += 1
    # decks with statistics: the reference's own statsdump (its sampling half, oracle/extract_statsdump.sh) runs untouched on
    # the host arrays the drop-in modules keep / refresh -- its running averages against those of the all-reference run
    nz = int(fix["meta"].data[2])
    u2 = np.abs(fix["st.uutc"].data).max() if "st.uutc" in fix else 0.
    for key, ref in fix.items():
        if not key.startswith(("st.", "xyt.", "yt.")):
            continue
        a, b = got[key].data[:nz], ref.data[:nz]
        if key.startswith("yt."):      # -999 in columns without fluid points
            assert np.array_equal(a == -999., b == -999.), key
            a, b = np.where(b == -999., 0., a), np.where(b == -999., 0., b)
        sc = max(np.abs(b).max(), 1e-3 * u2, 1e-6 * np.abs(fix["st.thlthlt"].data).max() if "thlp" in key else 0.)
        assert np.abs(a - b).max() <= 2e-9 * sc, key
        checked += 1
    assert checked >= 8


def test_fortran_kernel_sequence(tmp_path):
    """advection / subgrid / forces / poisson / tstep_integrate called one by one from Fortran."""
    name, iexp = "k_vreman_12x8x6", KERNEL_CASES["k_vreman_12x8x6"]
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "kernels", tmp_path, 0)
    for key in ("adv.up", "adv.vp", "adv.wp", "sub.up", "sub.vp", "sub.wp", "sub.ekm", "pre.up", "poi.up",
                "poi.vp", "poi.wp", "out.u0", "out.v0", "out.w0"):
        a, b = got[key].data, fix[key].data
        if key.startswith("out.") or key == "sub.ekm":
            a, b = a[1:-1], b[1:-1]
            assert relerr(nocorner(a), nocorner(b)) <= 1e-10, key
        else:
            assert relerr(a[:-1, 1:-1, 1:-1], b[:-1, 1:-1, 1:-1]) <= 1e-10, key


@pytest.mark.parametrize("residency", [0, 1, 2])
def test_fortran_driver_fixuinf2(residency, tmp_path):
    """ifixuinf = 2 through the drop-in modtstep: the host's fixuinf2 sets dgdt, tstep_integrate advances dp/dx."""
    name, iexp = "run_fix2_16x8x12s", 44
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "run", tmp_path, residency)
    for tag in ("s003", "s009"):
        np.testing.assert_allclose(got[tag + ".dpdxl"].data, fix[tag + ".dpdxl"].data, rtol=1e-10, atol=0)
        for k in ("u0", "v0", "w0", "pres0"):
            a, b = got[f"{tag}.{k}"].data[1:-1], fix[f"{tag}.{k}"].data[1:-1]
            assert relerr(nocorner(a), nocorner(b)) <= 1e-9, (tag, k)


@pytest.mark.parametrize("residency", [0, 1, 2])
def test_fortran_driver_adaptive_dt(residency, tmp_path):
    """ladaptive through the drop-in modtstep::tstep_update (maxima on the device, src/modtstep.f90:49-154)."""
    name, iexp = "run_adaptive_16x8x12s", 46
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "run", tmp_path, residency)
    for tag in ("s003", "s009", "s018"):
        np.testing.assert_allclose(got[tag + ".time"].data, fix[tag + ".time"].data, rtol=1e-10, atol=0)
        for k in ("u0", "v0", "w0", "pres0"):
            a, b = got[f"{tag}.{k}"].data[1:-1], fix[f"{tag}.{k}"].data[1:-1]
            assert relerr(nocorner(a), nocorner(b)) <= 1e-9, (tag, k)


@pytest.mark.parametrize("residency", [0, 2])
def test_fortran_driver_shifted_pbcs(residency, tmp_path):
    """&BC ds > 0 through the drop-in modforces::shiftedPBCs."""
    name, iexp = "run_shift_16x8x12s", 49
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "run", tmp_path, residency)
    for k in ("u0", "v0", "w0", "pres0"):
        a, b = got[f"s006.{k}"].data[1:-1], fix[f"s006.{k}"].data[1:-1]
        assert relerr(nocorner(a), nocorner(b)) <= 1e-9, k


@pytest.mark.parametrize("residency", [0, 2])
def test_fortran_driver_level_forcings(residency, tmp_path):
    """lstend, nudge, grwdamp through the drop-in modforces / modboundary (tables built in Fortran from the slab
    averages the drop-in thermodynamics fetched)."""
    name, iexp = "run_lsf_16x8x24s", 30
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "run", tmp_path, residency)
    tags = sorted({k.split(".")[0] for k in fix if k.endswith(".u0") and not k.startswith("s000")})
    assert tags
    for tag in tags:
        for k in ("u0", "v0", "w0", "pres0", "thl0"):
            if f"{tag}.{k}" not in fix:
                continue
            a, b = got[f"{tag}.{k}"].data[1:-1], fix[f"{tag}.{k}"].data[1:-1]
            assert relerr(nocorner(a), nocorner(b), 1.0 if k == "thl0" else None) <= 1e-9, (tag, k)


def test_fortran_device_mode_takes_the_fused_path(tmp_path):
    """UDC_RESIDENCY=2: every substep of the untouched loop runs as the fused substep (none routine by routine)."""
    name, iexp = "run_16x16x8", RUN_CASES["run_16x16x8"]
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/udales_dropin not built")
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    env = dict(os.environ, UDC_RESIDENCY="2")
    r = subprocess.run(f"ulimit -s unlimited; exec {BIN} namoptions.{iexp:03d} time none.bin", shell=True,
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300, executable="/bin/bash")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import re
    m = re.search(r"DROPIN residency=2 fused_substeps=(\d+) unfused=(\d+)", r.stdout)
    assert m, r.stdout[-1000:]
    assert int(m.group(1)) > 0 and int(m.group(2)) == 0
