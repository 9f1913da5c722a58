"""Per-level forcings (lstend / nudge / grwdamp, udcore/forcings.py) against the reference's own routines.

CPU: the table arithmetic on the inputs the reference had (fixture k_lsf_12x8x24: tendencies before and after its
lstend + nudge + grwdamp calls, and diagfld's slab averages).  GPU: a six-substep run with all three on, through the
fused substep, against the reference run (run_lsf_16x8x24s)."""
import numpy as np
import pytest

from common import carr, deck_path, interior, load_fixture, marr, nocorner, relerr
from udcore import read_deck
from udcore.forcings import LevelForcings
from udcore.grid import Grid


class _FakeCore:
    def __init__(self, g, nsv=0):
        self.g, self.nsv = g, nsv


def _avg(a, nz):
    out = np.zeros(nz + 2)
    out[1:nz + 2] = a[1:nz + 2, 1:-1, 1:-1].mean(axis=(1, 2))
    return out


LSF_KERNELS = [("k_lsf_12x8x24", 29, 2), ("k_lsfq_12x8x20", 34, 3)]      # (fixture, iexpnr, igrw_damp)


@pytest.mark.parametrize("name,iexp,igrw", LSF_KERNELS)
def test_tables_match_reference_routines(name, iexp, igrw):
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    g = Grid.from_deck(d)
    nz = g.nz
    nsv = int(d.get("SCALARS", "nsv"))
    ls = LevelForcings(_FakeCore(g, nsv), d)
    assert ls.active and ls.subsidence and ls.lnudge and ls.igrw == igrw
    qt = ls.lmoist
    assert qt == ("sub.qt0" in fix) and ls.qtls == qt
    fields = {k: marr(fix, "sub." + k if k == "u0" else "in." + k, nz) for k in ("u0", "v0", "w0")}
    fields["thl0"] = marr(fix, "sub.thl0", nz)      # top ghost row re-imposed by closurebc before the forcings run
    if qt:
        fields["qt0"] = marr(fix, "sub.qt0", nz)
    for n in range(nsv):       # kappa scalars: subsidence and nudging towards the initial profile
        fields[f"sv0_{n}"] = carr(fix, f"in.sv0_{n + 1:02d}", nz)[1:-1, 1:-1, 1:-1]
    av = {k: _avg(fields[k], nz) for k in fields if k != "w0"}
    if qt:
        np.testing.assert_allclose(av["qt0"][1:nz + 1], fix["qt0av"].data[:nz], rtol=0, atol=1e-17)
    # the averages are diagfld's (src/modthermodynamics.f90:262-279)
    np.testing.assert_allclose(av["u0"][1:nz + 2], fix["u0av"].data, rtol=0, atol=2e-15)
    np.testing.assert_allclose(av["thl0"][1:nz + 1], fix["thl0av"].data[:nz], rtol=0, atol=1e-12)
    tabs = ls.tables(av)
    tends = ("up", "vp", "wp", "thlp") + (("qtp",) if qt else ()) + tuple(f"svp_{n}" for n in range(nsv))
    assert {t for t, _ in tabs} == set(tends)
    for tend in tends:
        if tend.startswith("svp_"):
            key = f"svp_{int(tend[4:]) + 1:02d}"
            t = carr(fix, "frc0." + key, nz)[1:-1, 1:-1, 1:-1].copy()
            src, A, B = tabs[(tend, 0)]
            assert src is None and (tend, 1) not in tabs
            for k in range(1, nz + 1):
                t[k] = t[k] + A[k]
            ref = carr(fix, "lsf." + key, nz)[1:-1, 1:-1, 1:-1]
            sc = np.abs(ref - carr(fix, "frc0." + key, nz)[1:-1, 1:-1, 1:-1]).max()
            assert sc > 1e-4 and np.abs(interior(t) - interior(ref)).max() <= 1e-12 * sc, tend
            continue
        t = marr(fix, "frc0." + tend, nz).copy()
        for when in (0, 1):
            if (tend, when) not in tabs:
                continue
            src, A, B = tabs[(tend, when)]
            for k in range(1, nz + 1):
                t[k] = t[k] + A[k] + (B[k] * fields[src][k] if src else 0.)
        sc = np.abs(marr(fix, "lsf." + tend, nz) - marr(fix, "frc0." + tend, nz)).max()
        assert sc > (1e-9 if tend == "qtp" else 1e-6), tend       # the forcings did something
        tol = 1e-12 * (sc if tend == "qtp" else max(sc, 1.))
        assert np.abs(interior(t) - interior(marr(fix, "lsf." + tend, nz))).max() <= tol, tend


@pytest.mark.gpu
@pytest.mark.parametrize("name,iexp,igrw", LSF_KERNELS)
def test_device_applies_tables_like_reference(name, iexp, igrw):
    """udc_slab_average + udc_set_level_forcing + udc_level_forcings on the reference's inputs."""
    import udcore
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    nz = core.g.nz
    for k in ("u0", "v0", "w0", "um", "vm", "wm", "pres0"):
        core.upload(k, marr(fix, ("sub." if k == "u0" else "in.") + k, nz))
    core.upload("thl0", marr(fix, "sub.thl0", nz))
    core.upload("thlm", marr(fix, "in.thlm", nz))
    tends = ("up", "vp", "wp", "thlp") + (("qtp",) if core.lmoist else ())
    from udcore import lib as L
    for n in range(core.nsv):
        core.upload(L.scalar_field(L.SV0, n), carr(fix, f"in.sv0_{n + 1:02d}", nz))
        core.upload(L.scalar_field(L.SVM, n), carr(fix, f"in.sv0_{n + 1:02d}", nz))
        core.upload(L.scalar_field(L.SVP, n), carr(fix, f"frc0.svp_{n + 1:02d}", nz))
    if core.lmoist:
        core.upload("qt0", marr(fix, "sub.qt0", nz))
        core.upload("qtm", marr(fix, "in.qtm", nz))
    for k in tends:
        core.upload(k, marr(fix, "frc0." + k, nz))
    # slab averages = diagfld's
    np.testing.assert_allclose(core.slab_average("u0")[1:nz + 2], fix["u0av"].data, rtol=0, atol=5e-15)
    ls = LevelForcings(core, d)
    ls.update()
    core.level_forcings(0)
    core.level_forcings(1)
    for k in tends:
        sc = np.abs(marr(fix, "lsf." + k, nz) - marr(fix, "frc0." + k, nz)).max()
        tol = 1e-11 * (sc if k == "qtp" else max(sc, 1.))
        assert np.abs(interior(core.download(k)) - interior(marr(fix, "lsf." + k, nz))).max() <= tol, k
    for n in range(core.nsv):
        got = core.download(L.scalar_field(L.SVP, n), halo=2)
        ref = carr(fix, f"lsf.svp_{n + 1:02d}", nz)
        sc = np.abs(ref - carr(fix, f"frc0.svp_{n + 1:02d}", nz)).max()
        assert np.abs(interior(got, 2) - interior(ref, 2)).max() <= 1e-11 * sc
    core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_run_with_level_forcings_matches_reference(fused):
    import udcore
    from udcore import cold_start
    name, iexp = "run_lsf_16x8x24s", 30
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, pre_boundary=True))      # the fields as readinitfiles leaves them
    ls = LevelForcings(core, d)
    core.start_up(before_boundary=ls.capture_startup)             # thermodynamics (diagfld's averages), then boundary
    assert ls.active and ls.igrw == 1 and ls.lcoriol
    dt = float(d.get("RUN", "dtmax"))
    dumps = sorted(int(k[1:4]) for k in fix if k.endswith(".u0") and k != "s000.u0")
    for isub in range(1, max(dumps) + 1):
        ls.update()                                  # diagfld's averages of the state the substep starts from
        if fused:
            core.substep((isub - 1) % 3 + 1, dt, with_forces=True)
        else:
            core.tstep_update(dt)
            core.advection(); core.subgrid(); core.bottom(); core.coriolis(); core.forces()
            core.level_forcings(0); core.masscorr(); core.level_forcings(1)
            core.poisson(); core.tstep_integrate(); core.halos(); core.boundary()
        if isub in dumps:
            for k in ("u0", "v0", "w0", "pres0", "thl0"):
                ref = marr(fix, f"s{isub:03d}.{k}", core.g.nz)
                sc = 1.0 if k == "thl0" else None
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]), sc) <= 1e-9, (isub, k)
    core.close()


def test_fixuinf1_table_matches_reference_routine():
    """ifixuinf = 1 (fixuinf1, src/modforces.f90:220-288): on RK stage 3 the top-level mean is pulled back to Uinf / Vinf."""
    name, iexp = "k_fix1_12x8x6", 43
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    g = Grid.from_deck(d)
    nz = g.nz
    ls = LevelForcings(_FakeCore(g), d)
    assert ls.active and ls.ifixuinf == 1 and ls.lvinf and ls.Uinf == 1.1 and ls.Vinf == 0.02
    rk3step, dt = int(fix["rk3"].data[0]), float(fix["rk3"].data[1])
    assert rk3step == 3
    av = {"u0": _avg(marr(fix, "sub.u0", nz), nz), "v0": _avg(marr(fix, "in.v0", nz), nz)}
    tabs = ls.tables(av, rk3step, dt)
    assert set(tabs) == {("up", 1), ("vp", 1)}
    for tend in ("up", "vp"):
        src, A, B = tabs[(tend, 1)]
        assert src is None and not B.any()
        t = marr(fix, "fix0." + tend, nz).copy()
        for k in range(1, nz + 1):
            t[k] = t[k] + A[k]
        ref = marr(fix, "pre." + tend, nz)
        sc = np.abs(ref - marr(fix, "fix0." + tend, nz)).max()
        assert sc > 1e-7 and np.abs(interior(t) - interior(ref)).max() <= 1e-9 * sc, tend      # (earlier stage-3 substeps already pulled the mean close to Uinf)
    # other RK stages: the same keys, zero tables (so that a stage-3 table never lingers on the device)
    off = ls.tables(av, 1, dt)
    assert set(off) == {("up", 1), ("vp", 1)} and not off[("up", 1)][1].any()


@pytest.mark.gpu
def test_run_with_fixuinf2_matches_reference():
    """ifixuinf = 2: dp/dx follows the ODE d(dpdx)/dt = (freestreamav - Uinf)/tscale (fixuinf2 + tstep_integrate)."""
    import udcore
    from udcore import cold_start
    name, iexp = "run_fix2_16x8x12s", 44
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, pre_boundary=True))      # the fields as readinitfiles leaves them
    ls = LevelForcings(core, d)
    core.start_up(before_boundary=ls.capture_startup)             # thermodynamics (diagfld's averages), then boundary
    assert ls.active and ls.ifixuinf == 2
    dt = float(d.get("RUN", "dtmax"))
    nz = core.g.nz
    for isub in range(1, 10):
        rk = (isub - 1) % 3 + 1
        ls.update(rk, dt)
        core.substep(rk, dt, with_forces=True)
        if isub in (3, 6, 9):
            tag = f"s{isub:03d}"
            np.testing.assert_allclose(core.dpdxl + ls._pending, fix[tag + ".dpdxl"].data[:nz], rtol=1e-11, atol=0)
            for k in ("u0", "v0", "w0", "pres0"):
                ref = marr(fix, f"{tag}.{k}", nz)
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1])) <= 1e-9, (isub, k)
    assert abs(core.dpdxl[0] + ls._pending) > 0.3          # the ODE moved dp/dx a long way from its start value
    core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_run_with_shifted_pbcs_matches_reference(fused):
    """&BC ds > 0 (shiftedPBCs, src/modforces.f90:953-980): a spanwise drift vs(i, k) ~ u0av(k) sin(..) in the downstream half."""
    import udcore
    from udcore import cold_start
    name, iexp = "run_shift_16x8x12s", 49
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, pre_boundary=True))      # the fields as readinitfiles leaves them
    ls = LevelForcings(core, d)
    core.start_up(before_boundary=ls.capture_startup)             # thermodynamics (diagfld's averages), then boundary
    assert ls.active and ls.ds == 1.5 and ls.shift_sinx[:8].max() == 0. and ls.shift_sinx[8:].max() > 0.9
    dt = float(d.get("RUN", "dtmax"))
    for isub in range(1, 7):
        rk = (isub - 1) % 3 + 1
        ls.update(rk, dt)
        if fused:
            core.substep(rk, dt, with_forces=True)
        else:
            core.tstep_update(dt)
            core.advection(); core.shifted_pbcs(); core.subgrid(); core.bottom(); core.coriolis(); core.forces()
            core.masscorr(); core.poisson(); core.tstep_integrate(); core.halos(); core.boundary()
        if isub in (3, 6):
            for k in ("u0", "v0", "w0", "pres0"):
                ref = marr(fix, f"s{isub:03d}.{k}", core.g.nz)
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1])) <= 1e-9, (isub, k)
    # the drift is there: without it the run ends elsewhere
    core.close()
