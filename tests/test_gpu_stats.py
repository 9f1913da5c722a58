"""Statistics accumulation on the device (udc_stats.hip, udcore/stats.py: the time-averaged 3-D set of tdump and the
x-, y- and time-averaged profiles of xytdump, src/modstatsdump.f90) against the running averages the reference's own
statsdump lines leave behind (fixtures run_stats_*: oracle/extract_statsdump.sh compiles the sampling half of the
reference routine where it lies), against the numpy restatement oracle/stats_oracle.py on the states the device itself
produced, plus properties that hold whatever the restatement says: with one sample the mean is the sample; the running
average of N equally weighted samples is their arithmetic mean; variances are non-negative up to round-off; the clocks
sample every tsample and dump every tstatsdump seconds."""
import os
import sys

import numpy as np
import pytest

from common import deck_path, interior, load_fixture
from udcore import cold_start, read_deck
from udcore import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu


def _state(core):
    st = {k: core.download(k) for k in ("um", "vm", "wm", "pres0", "ekh")}
    if core.ltempeq:
        st["thlm"] = core.download("thlm")
    if core.lmoist:
        st["qtm"] = core.download("qtm")
    for n in range(core.nsv):
        st[f"svm_{n}"] = core.download(L.scalar_field(L.SVM, n), halo=2)
    return st


@pytest.mark.parametrize("name,iexp", [("run_smag_scalar_16x8x12s", 22), ("run_thl_16x8x12s", 25), ("run_qt_16x8x12s", 33)])
def test_tdump_accumulation_matches_restatement(name, iexp):
    import udcore
    from stats_oracle import TDumpOracle
    from udcore.stats import TDump
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, nsv=core.nsv))
    dt = float(d.get("RUN", "dtmax"))
    td = TDump(core, tsample=2 * dt, tstatsdump=6 * dt)
    orc = TDumpOracle(core.g, nsv=core.nsv, ltempeq=core.ltempeq, lmoist=core.lmoist)
    timee, events = 0., []
    for step in range(8):
        for rk in (1, 2, 3):
            core.substep(rk, dt, True)
        timee += dt
        ts, T = td.tsamplep, td.tstatsdumpp
        if ts == 0. and td.tsample <= dt:
            ts = dt
        if T == 0. and td.tsample <= dt:
            T = dt
        will_sample = ts >= td.tsample
        st = _state(core) if will_sample else None
        what = td.step(3, dt, timee)
        events.append(what)
        if will_sample:
            orc.sample(st, ts, T)
        if what == "dump":
            got, ref = td.dumps[-1][1], orc.output()
            assert set(got) == set(ref)
            for k in ref:
                scale = max(np.abs(ref[k]).max(), 1e-300) if not k.startswith(("up", "vp", "wp", "tke", "thlp", "sca1p")) else max(np.abs(orc.acc["uutc"]).max(), 1.)
                assert np.abs(got[k] - ref[k]).max() <= 1e-12 * scale, (step, k)
            for k in ("upuptc", "vpvptc", "wpwptc", "tketc"):
                assert got[k].min() >= -1e-12
    # clocks: tsample = 2 dt -> a sample every second step; tstatsdump = 6 dt -> dumps at steps 6 (and the clock restarts)
    assert events.count("dump") >= 1 and td.nsamples >= 3
    core.close()


XYT_FIX = {"uwtxyik": "uwxyt", "vwtxyjk": "vwxyt", "wwtxyk": "wwxyt", "uvtxyij": "uvxyt", "upwptxyik": "upwpxyt", "vpwptxyjk": "vpwpxyt",
           "upvptxyij": "upvpxyt", "wthltxyk": "wthlxyt", "wpthlptxyk": "wpthlpxyt"}


@pytest.mark.parametrize("slabs", [1, 2])
@pytest.mark.parametrize("name,iexp", [("run_stats_16x8x12s", 62), ("run_stats_ibm_16x12x10", 63)])
def test_statistics_match_reference(name, iexp, slabs):
    """tdump's 3-D running averages and xytdump's table after the deck's run, against what the reference's own statsdump
    (its sampling half, compiled from src/modstatsdump.f90) left in modfields: with and without obstacles (masked slab
    averages, the floor level of the w-point masks without fluid points), one sample per step and one every second step
    with a dump in between; on one slab and through the forced-slab (multi-rank) path."""
    import udcore
    from udcore.ibm import read_ibm
    from udcore.stats import TDump
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    os.environ["UDC_FORCE_SLAB"] = "1" if slabs == 2 else "0"
    try:
        core = udcore.from_deck(d)
    finally:
        os.environ.pop("UDC_FORCE_SLAB")
    core.load_state(cold_start(core.g, d, nsv=core.nsv))
    dt = float(d.get("RUN", "dtmax"))
    lists = read_ibm(d) if d.get("RUN", "libm") else None
    td = TDump(core, float(d.get("OUTPUT", "tsample")), float(d.get("OUTPUT", "tstatsdump")), xyt=True, ibm_lists=lists)
    nsub = max(int(k[1:4]) for k in fix if k.endswith(".u0"))
    timee = 0.
    for isub in range(1, nsub + 1):
        rk = (isub - 1) % 3 + 1
        core.substep(rk, dt, True)
        if rk == 3:
            timee += dt
        td.step(rk, dt, timee)
    assert td.nsamples >= 3
    nz = core.g.nz
    acc, xyt = td.accumulators(), td.xyt()
    u2 = np.abs(fix["st.uutc"].data).max()
    checked = 0
    for k, rec in fix.items():
        if k.startswith("st."):
            got, ref = acc[k[3:]][:nz], rec.data[:nz]
        elif k.startswith("xyt."):
            got, ref = xyt[XYT_FIX.get(k[4:], k[4:])], rec.data[:nz]
        else:
            continue
        scale = max(np.abs(ref).max(), 1e-30)
        if k.startswith("xyt.") or k[3:] in ("uwtik", "vwtjk", "uvtij", "wtik", "wtjk", "wmt", "wtc", "wwtc", "vvtc"):
            scale = max(scale, 1e-3 * u2)
        if "thlp" in k:
            scale = max(scale, 1e-6 * np.abs(fix["st.thlthlt"].data).max())
        assert np.abs(got - ref).max() <= 2e-9 * scale, (k, np.abs(got - ref).max(), scale)      # RUN_TOL of the state parity tests
        checked += 1
    assert checked >= 50
    core.close()


YT_FIX = {"upwptyik": "upwpyt", "uwtyik": "uwyt", "upuptyc": "upupyt", "wpwptyc": "wpwpyt", "wpthlptyk": "wpthlpyt", "wthltyk": "wthlyt",
          "thlpthlpty": "thlpthlpyt", "wpsv1ptyk": "wpsca1tpyt", "wsv1tyk": "wsca1yt", "sv1psv1pty": "sca1tpsca1pyt", "sv1sgsyt": "sca1sgsyt"}


@pytest.mark.parametrize("slabs", [1, 2])
def test_ytdump_matches_reference(slabs):
    """ytdump (y- and time-averaged x-z fields) after the deck's run against what the reference's own statsdump left in modfields
    and its avey_ibm gives for the table: obstacles (columns without fluid points are -999), temperature, one scalar; on one slab
    and through the forced-slab path."""
    import udcore
    from udcore.ibm import read_ibm
    from udcore.stats import TDump
    name, iexp = "run_ytstats_ibm_16x12x10", 78
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    os.environ["UDC_FORCE_SLAB"] = "1" if slabs == 2 else "0"
    try:
        core = udcore.from_deck(d)
    finally:
        os.environ.pop("UDC_FORCE_SLAB")
    core.load_state(cold_start(core.g, d, nsv=core.nsv))
    dt = float(d.get("RUN", "dtmax"))
    td = TDump(core, float(d.get("OUTPUT", "tsample")), float(d.get("OUTPUT", "tstatsdump")), yt=True, ibm_lists=read_ibm(d))
    nsub = max(int(k[1:4]) for k in fix if k.endswith(".u0"))
    timee = 0.
    for isub in range(1, nsub + 1):
        rk = (isub - 1) % 3 + 1
        core.substep(rk, dt, True)
        if rk == 3:
            timee += dt
        td.step(rk, dt, timee)
    yt = td.yt()
    u2 = np.abs(fix["st.uutc"].data).max()
    checked = 0
    for k, rec in fix.items():
        if not k.startswith("yt."):
            continue
        ref = rec.data[:, 0, :]
        got = yt[YT_FIX.get(k[3:], k[3:])]
        assert np.array_equal(got == -999., ref == -999.), k
        ok = ref != -999.
        scale = max(np.abs(ref[ok]).max(), 1e-3 * u2, 1e-6 * np.abs(fix["st.thlthlt"].data).max() if "thlp" in k else 0.)
        assert np.abs(got[ok] - ref[ok]).max() <= 2e-9 * scale, (k, np.abs(got[ok] - ref[ok]).max(), scale)
        checked += 1
    assert checked >= 18 and (yt["wyt"] == -999.).any()
    core.close()


@pytest.mark.parametrize("fused", [True, False])
def test_stage3_assignment_carries_the_planes_below_the_floor(fused):
    """`thlm = thl0`, `svm = sv0`, `um = u0` at RK stage 3 are whole-array assignments in the reference
    (src/modtstep.f90:331-339): the planes below the floor, which no boundary routine owns, go along.  A driver that starts
    with thlm's still at zero (the reference's own cold start sets thl0(kb-1) only, src/modstartup.f90:1208) must find
    thl0's there after the first step -- statsdump's thlk(kb) reads it (src/modstatsdump.f90:861)."""
    import udcore
    name, iexp = "run_stats_16x8x12s", 62
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    st = cold_start(core.g, d, nsv=core.nsv)
    floor = st["thl0"][0].copy()
    assert floor[1:-1, 1:-1].min() > 280.
    st["thlm"][0] = 0.
    core.load_state(st)
    dt = float(d.get("RUN", "dtmax"))
    for rk in (1, 2, 3):
        if rk == 3:
            assert np.all(core.download("thlm")[0, 1:-1, 1:-1] == 0.)
        if fused:
            core.substep(rk, dt, True)
        else:
            core.tstep_update(dt)
            core.advection(); core.subgrid(); core.bottom(); core.coriolis(); core.forces(); core.masscorr(); core.scalsource()
            core.poisson(); core.tstep_integrate(); core.halos(); core.boundary()
    assert np.array_equal(core.download("thlm")[0, 1:-1, 1:-1], floor[1:-1, 1:-1])
    assert np.array_equal(core.download("thl0")[0, 1:-1, 1:-1], floor[1:-1, 1:-1])
    assert np.all(core.download("um")[0, 1:-1, 1:-1] == 0.)
    core.close()


def test_running_average_is_the_arithmetic_mean():
    import udcore
    from udcore.stats import TDump
    name, iexp = "run_16x16x8", 21
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, nsv=core.nsv))
    dt = float(d.get("RUN", "dtmax"))
    td = TDump(core, tsample=dt, tstatsdump=1e9)
    samples = []
    timee = 0.
    for step in range(5):
        for rk in (1, 2, 3):
            core.substep(rk, dt, True)
        timee += dt
        samples.append(core.download("um")[1:, 1:-1, 1:-1])
        assert td.step(3, dt, timee) == "sample"
        if step == 0:      # one sample: the mean is the sample
            assert np.array_equal(td.get(0), samples[0])
    mean = np.mean(samples, axis=0)
    assert np.abs(td.get(0) - mean).max() <= 1e-14 * np.abs(mean).max()
    core.close()
