"""Statistics accumulation on the device (udc_stats.hip, udcore/stats.py: the time-averaged 3-D set of tdump,
src/modstatsdump.f90) against the numpy restatement oracle/stats_oracle.py on the states the device itself produced, plus
properties that hold whatever the restatement says: with one sample the mean is the sample; the running average of N
equally weighted samples is their arithmetic mean; variances are non-negative up to round-off; the clocks sample every
tsample and dump every tstatsdump seconds.  (The reference's modstatsdump needs NetCDF and cannot be compiled here:
parity for this routine is unpinned, see the oracle's header.)"""
import os
import sys

import numpy as np
import pytest

from common import deck_path, interior
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
