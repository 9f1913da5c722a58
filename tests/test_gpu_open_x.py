"""Inflow / outflow in x (&BC BCxm = 2; u-dales_amd/csrc/udc_xopen.hip) against vectors of the reference's own Fortran: decks with
BCxm = 2 run through oracle/_ref/udales_ref (tests/golden/make_golden.py: k_xopen_*, run_xopen_*), fields dumped WITH their x
ghost columns ib-1 and ie+1 -- which are state here (the inlet's mirrored values, the convective outlet's v, w and the
prognostic u(ie+1)), so they are compared too.

The oracle restates the branch too (oracle/udcore_oracle.c orc_set_open_x, pinned on these decks by tests/test_oracle_open_x.py) and
checks the device on seeded fields at other sizes (test_against_oracle_seeded).  Tolerances as in tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from common import carr, deck_path, interior, load_fixture, marr, nocorner, relerr
from udcore import lib as L
from udcore import read_deck

pytestmark = pytest.mark.gpu

KERNEL_TOL = 1e-11
RUN_TOL = 1e-9
K_CASES = {"k_xopen_16x8x12": 90, "k_xopen_thl_16x8x12": 100, "k_xopen_sv_16x8x12": 102, "k_xopen_moist_16x8x12": 106}
# BCxm = 3: the inlet from a precursor run's planes (the reference's moddriver stays on the host: Fortran routes only)
D_CASES = {"run_xdriver_16x8x12s": 96, "run_xdriver_ibm_16x12x10": 98, "run_xdriver_moist_16x12x10": 110,
           "run_xdriver_scal_16x8x12s": 113}      # (BCxT = BCxq = BCxs = 3: the scalars' inlets from the precursor's planes too)
R_CASES = {"run_xopen_16x8x12s": 91, "run_xopen_vr_24x8x10": 92, "run_xopen_ibm_16x12x10": 93, "run_xopen_ibmwf3_16x12x10": 94,
           "run_xopen_volflow_16x8x12s": 99, "run_xopen_thl_16x8x12s": 101,
           "run_xopen_sv_16x8x12s": 103, "run_xopen_ibm_sv_16x12x10": 104, "run_xopen_ibm_thl_16x12x10": 105,
           # the temperature and the total water periodic in x (BCxT = BCxq = 1: the reference's defaults, its tests/cases/525) beside the inflow / outflow
           "run_xopen_moist_16x8x12s": 107, "run_xopen_ibm_moist_16x12x10": 108,
           "run_xopen_qt2_16x8x12s": 111}      # (BCxT = BCxq = 2: both enter with prof.inp's profiles)


def make_core(name, iexp):
    import udcore
    d = read_deck(deck_path(name, iexp))
    core = udcore.from_deck(d)
    assert core.open_x
    return d, core


def wlev(g):
    return g.dzf[1:g.nz + 1] / (g.zh[g.nz + 1] - g.zh[2])      # src/modboundary.f90:146-156


def xcols(a):
    """The interior rows and levels of an m-array, x ghost columns included."""
    return a[1:-1, 1:-1, :]


@pytest.mark.parametrize("name,iexp", sorted(K_CASES.items()))
def test_each_routine_matches_reference(name, iexp):
    fix = load_fixture(name)
    d, core = make_core(name, iexp)
    g, nz = core.g, core.g.nz
    thl = "in.thl0" in fix      # (BCxT = 2: the temperature enters with its profile and leaves convectively too; BCxT = 1: periodic)
    qt = "in.qt0" in fix
    tprof = thl and int(d.get("BC", "BCxT")) == 2
    for k in ("u0", "v0", "w0", "um", "vm", "wm", "pres0", "ekm", "ekh") + (("thl0", "thlm") if thl else ()) + (("qt0", "qtm") if qt else ()):
        core.upload(k, np.nan_to_num(marr(fix, "in." + k, nz)))
    if "thm.presf" in fix:      # what the reference's last thermodynamics call left behind
        core.thermo_state({n: np.concatenate(([0.], fix["thm." + n].data)) if "thm." + n in fix else np.zeros(nz + 2) for n in core.TH_TABLES})
    # what the upload took: the ghost columns come back as they went in
    for k in ("u0", "v0", "w0", "vm", "pres0") + (("thl0", "thlm") if thl else ()) + (("qt0", "qtm") if qt else ()):
        assert np.array_equal(xcols(core.download(k)), xcols(marr(fix, "in." + k, nz))), k
    nsv = core.nsv      # (BCxs = 2: the scalars' c-arrays carry two ghost columns either side)
    for n in range(nsv):
        core.upload(L.scalar_field(L.SV0, n), carr(fix, f"in.sv0_{n + 1:02d}", nz))
        core.upload(L.scalar_field(L.SVM, n), carr(fix, f"in.svm_{n + 1:02d}", nz))
        assert np.array_equal(core.download(L.scalar_field(L.SV0, n), halo=2)[2:-2, 2:-2, :], carr(fix, f"in.sv0_{n + 1:02d}", nz)[2:-2, 2:-2, :])
    zero = np.zeros(g.mshape())

    def zero_tend():
        for k in ("up", "vp", "wp") + (("thlp",) if thl else ()) + (("qtp",) if qt else ()):
            core.upload(k, zero)
        for n in range(nsv):
            core.upload(L.scalar_field(L.SVP, n), np.zeros(g.cshape()))

    def svp_is(tag):
        for n in range(nsv):
            got = core.download(L.scalar_field(L.SVP, n), halo=2)
            assert relerr(interior(got, 2), interior(carr(fix, f"{tag}.svp_{n + 1:02d}", nz), 2)) <= KERNEL_TOL, (tag, n)

    def thlp_is(tag):
        # (without the column ib: xTi_profile overwrites thl(ib, kb..ke) with the profile at every `boundary`, so its tendency is never
        #  used -- and at (ib, ke) it differs: `boundary` imposes the top ghost from thl(ib, ke) BEFORE xTi_profile replaces that value,
        #  the reference's subgrid re-imposes it from the new one (reassure_fluxtop_boundary) between advection and diffusion, the device
        #  re-imposes a zero-flux top only where a flux is prescribed)
        if thl:
            c0 = 1 if tprof else 0      # (periodic in x: every column)
            assert relerr(interior(core.download("thlp"))[:, :, c0:], interior(marr(fix, tag + ".thlp", nz))[:, :, c0:]) <= KERNEL_TOL, tag
        if qt:
            assert relerr(interior(core.download("qtp")), interior(marr(fix, tag + ".qtp", nz))) <= KERNEL_TOL, tag

    zero_tend()
    core.advection()
    for k in ("up", "vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "adv." + k, nz))) <= KERNEL_TOL, k
    thlp_is("adv")
    svp_is("adv")
    # (the test has teeth: the same sweep on periodic x gives something else in the columns next to the ends)
    ref = interior(marr(fix, "adv.vp", nz))
    assert np.abs(ref[:, :, 0]).max() > 0
    zero_tend()
    core.subgrid()
    assert relerr(core.download("ekm"), marr(fix, "sub.ekm", nz)) <= KERNEL_TOL          # every ghost included (closurebc)
    assert relerr(core.download("ekh"), marr(fix, "sub.ekh", nz)) <= KERNEL_TOL
    assert relerr(core.download("u0"), marr(fix, "sub.u0", nz)) <= KERNEL_TOL
    for k in ("up", "vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "sub." + k, nz))) <= KERNEL_TOL, k
    thlp_is("sub")
    svp_is("sub")
    core.bottom_diagnostics(True)
    core.bottom()
    for k in ("up", "vp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "bot." + k, nz))) <= KERNEL_TOL, k
    for nm, k in (("tau_x", "up"), ("tau_y", "vp")):      # [jtot][itot]: the deck's columns, not the device row's
        inc = interior(marr(fix, "bot." + k, nz))[0] - interior(marr(fix, "sub." + k, nz))[0]
        got = core.bottom_diag(nm)
        assert got.shape == inc.shape and np.abs(got - inc).max() <= 1e-12 * np.abs(inc).max(), nm
    zero_tend()
    core.advection(); core.subgrid(); core.bottom(); core.coriolis(); core.forces()
    core.rk3step, core.dt = int(fix["rk3"].data[0]), float(fix["rk3"].data[1])
    core.masscorr()
    for k in ("up", "vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "pre." + k, nz))) <= KERNEL_TOL, k
    thlp_is("pre")
    svp_is("pre")
    # bcpup reads the outlet's speed as the previous substep's `boundary` left it (dumped by the driver)
    core.set_open_x_outflow(None, float(fix["in.uouttot"].data[0]))
    core.poisson()
    rk3coef = core.dt / (4. - core.rk3step)
    pnat = 1e-2 * np.abs(marr(fix, "in.um", nz)).max() * g.dx / rk3coef
    pscale = max(np.abs(marr(fix, "poi.p", nz)).max(), np.abs(marr(fix, "poi.pres0", nz)).max(), pnat)
    # p with bcp's columns p(ib-1) = p(ib), p(ie+1) = p(ie)
    assert relerr(xcols(core.download("p")), xcols(marr(fix, "poi.p", nz)), pscale) <= KERNEL_TOL
    assert relerr(nocorner(core.download("pres0")[1:-1]), nocorner(marr(fix, "poi.pres0", nz)[1:-1]), pscale) <= KERNEL_TOL
    for k in ("vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "poi." + k, nz))) <= KERNEL_TOL, k
    # up with the inlet's column (zero) and the outlet's prognostic column up(ie+1)
    got, ref = xcols(core.download("up"))[:, :, 1:], xcols(marr(fix, "poi.up", nz))[:, :, 1:]
    assert relerr(got, ref) <= KERNEL_TOL
    assert np.all(got[:, :, 0] == 0.) and np.abs(ref[:, :, -1]).max() > 1e-3
    ref = marr(fix, "poi.wp", nz)
    assert relerr(core.download("wp")[nz + 1, 1:-1, 1:-1], ref[nz + 1, 1:-1, 1:-1], np.abs(ref).max()) <= KERNEL_TOL
    # ... and the `boundary` that ends the substep takes it from the slab averages of the state the substep started from
    core.set_open_x_outflow(wlev(g), 0.)
    core.tstep_integrate()
    core.halos()
    core.boundary()
    for k in ("u0", "v0", "w0", "um", "pres0"):
        ref = marr(fix, "out." + k, nz)
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]), pscale if k == "pres0" else None) <= KERNEL_TOL, k
    ref = marr(fix, "out.w0", nz)
    assert relerr(nocorner(core.download("w0"))[nz + 1], nocorner(ref)[nz + 1], np.abs(ref).max()) <= KERNEL_TOL
    for k in ("thl0", "thlm") if thl else ():      # with the inlet's and the outlet's columns; differences are O(1) K on a 288 K mean
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(marr(fix, "out." + k, nz)[1:-1]), 1.0) <= KERNEL_TOL, k
    for k in ("qt0", "qtm") if qt else ():
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(marr(fix, "out." + k, nz)[1:-1])) <= KERNEL_TOL, k
    for n in range(nsv):      # the interior rows and levels with ib-2, ib-1, ie+1, ie+2
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        assert relerr(got[2:-2, 2:-2, :], carr(fix, f"out.sv0_{n + 1:02d}", nz)[2:-2, 2:-2, :]) <= KERNEL_TOL, n
    core.close()


@pytest.mark.parametrize("fused", [True, False, "deferred"])
@pytest.mark.parametrize("name,iexp", sorted(R_CASES.items()))
def test_substeps_match_reference(name, iexp, fused, want_div_in_transform=False):
    """Chained substeps from the state the reference's start-up left (s000, x ghost columns included): the fused substep, the
    reference's routine-by-routine order, and that order with deferred execution."""
    fix = load_fixture(name)
    d, core = make_core(name, iexp)
    g = core.g
    thl = bool(d.get("PHYSICS", "ltempeq"))
    qt = bool(d.get("PHYSICS", "lmoist"))
    for k in ("u0", "v0", "w0", "um", "vm", "wm", "pres0") + (("thl0", "thlm") if thl else ()) + (("qt0", "qtm") if qt else ()):
        core.upload(k, marr(fix, "s000." + k, g.nz))
    if qt and fused is not True:
        core.thermodynamics()      # src/program.f90:120 (the fused substep makes the call itself on its first substep)
    for n in range(core.nsv):
        core.upload(L.scalar_field(L.SV0, n), carr(fix, f"s000.sv0_{n + 1:02d}", g.nz))
        core.upload(L.scalar_field(L.SVM, n), carr(fix, f"s000.svm_{n + 1:02d}", g.nz))
    # (the start-up's `boundary` has run: its speed is what bcpup reads first, and what the first substep's `boundary` still uses)
    if d.get("PHYSICS", "luvolflowr"):      # a prescribed volume flow names the outlet's speed, nothing else (masscorr is off with inflow / outflow)
        assert abs(core._uouttot - float(fix["s000.uouttot"].data[0])) < 1e-13      # (from_deck's ubulk)
    else:
        core.set_open_x_outflow(wlev(g), float(fix["s000.uouttot"].data[0]), hold_first=True)
    dt = float(d.get("RUN", "dtmax"))
    dumps = sorted(int(k[1:4]) for k in fix if k.endswith(".u0") and k != "s000.u0")
    if fused == "deferred":
        core.set_deferred(True)
    for isub in range(1, max(dumps) + 1):
        if fused is True:
            core.substep((isub - 1) % 3 + 1, dt, with_forces=True)
        else:
            core.tstep_update(dt)
            core.advection(); core.subgrid(); core.bottom(); core.coriolis(); core.forces(); core.ibmwallfun(); core.masscorr()
            core.ibmnorm(); core.scalsource(); core.poisson()
            core.tstep_integrate(); core.halos(); core.boundary()
            if core.moist_thermo:
                core.thermodynamics()
        if isub in dumps:
            tag = f"s{isub:03d}"
            for k in ("qt0", "qtm") if qt else ():      # (x ghost columns included: xq_periodic's)
                ref = marr(fix, f"{tag}.{k}", g.nz)
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1])) <= RUN_TOL, (tag, k)
            for k in ("u0", "v0", "w0", "pres0", "um", "vm", "wm"):
                ref = marr(fix, f"{tag}.{k}", g.nz)
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1])) <= RUN_TOL, (tag, k)
            for k in ("thl0", "thlm") if thl else ():
                ref = marr(fix, f"{tag}.{k}", g.nz)
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]), 1.0) <= RUN_TOL, (tag, k)
            for n in range(core.nsv):
                got = core.download(L.scalar_field(L.SV0, n), halo=2)
                assert relerr(got[2:-2, 2:-2, :], carr(fix, f"{tag}.sv0_{n + 1:02d}", g.nz)[2:-2, 2:-2, :]) <= RUN_TOL, (tag, n)
            ref = marr(fix, f"{tag}.w0", g.nz)
            assert relerr(nocorner(core.download("w0"))[g.nz + 1], nocorner(ref)[g.nz + 1], np.abs(ref).max()) <= RUN_TOL, (tag, "w0(ke+1)")
            if fused is True:
                assert abs(float(fix[f"{tag}.uouttot"].data[0])) > 0.1
    # the flow really is not periodic: what leaves differs from what enters
    u = core.download("u0")
    assert np.abs(u[1:-1, 1:-1, 1] - u[1:-1, 1:-1, -1]).max() > 1e-3
    if fused == "deferred":
        assert core.deferred_stats() == (max(dumps), 0)
    if fused is True:
        plan = core.last_plan()
        assert plan["pressure_total_form"] and not plan["slab_layout"]      # (under the open lid too: udc_plan.h)
        assert plan["divergence_in_x_transform"] == want_div_in_transform
    div = core.divergence()
    assert div[0] < 1e-12
    core.close()


@pytest.mark.parametrize("name", ["run_xopen_16x8x12s", "run_xopen_volflow_16x8x12s", "run_xopen_moist_16x8x12s", "run_xopen_sv_16x8x12s"])
def test_divergence_inside_the_doubled_rows_transform(name, monkeypatch):
    """UDC_OWN_FWD=1 (the library's default from 128 rows on): the fused substep writes no right-hand side -- the x transform of the solver's
    doubled row evaluates the divergence of the handle's tendencies itself, the interior columns and their mirror image (fftx_fwd_nat_kernel<LM,
    true>).  Same fixtures, same tolerance; 2 itot = 32 here (the 256^3 property test runs it by default)."""
    monkeypatch.setenv("UDC_OWN_FWD", "1")
    test_substeps_match_reference(name, R_CASES[name], True, want_div_in_transform=True)


@pytest.mark.parametrize("name", ["run_xopen_16x8x12s", "run_xopen_ibmwf3_16x12x10", "run_xopen_thl_16x8x12s", "run_xopen_ibm_sv_16x12x10", "run_xopen_moist_16x8x12s"])
def test_cold_start_matches_reference(name):
    """From the deck alone, the way run_case.py starts: the fields as readinitfiles leaves them (the x ghost columns hold the
    profile, no noise), the start-up's slab averages, its `boundary` (uouttot from those averages, one convective step with
    rk3step = 0 and dt = dtmax / 100), then the loop."""
    from udcore import cold_start
    iexp = R_CASES[name]
    fix = load_fixture(name)
    d, core = make_core(name, iexp)
    g = core.g
    dt = float(d.get("RUN", "dtmax"))
    core.load_state(cold_start(g, d, nsv=core.nsv, pre_boundary=True))
    core.halos()
    core.start_up(dtmax=dt)
    for k in ("u0", "v0", "w0", "um", "vm", "wm"):
        ref = marr(fix, "s000." + k, g.nz)
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1])) <= 1e-13, k
    for n in range(core.nsv):
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        assert relerr(got[2:-2, 2:-2, :], carr(fix, f"s000.sv0_{n + 1:02d}", g.nz)[2:-2, 2:-2, :]) <= 1e-13, n
    vin = float(d.v[0])
    assert np.abs(marr(fix, "s000.v0", g.nz)[1:-1, 1:-1, -1] - vin).max() > 1e-6      # (the start-up's convective step moved the outlet)
    for isub in range(1, 4):
        core.substep(isub, dt, with_forces=True)
    for k in ("u0", "v0", "w0", "pres0", "vm") + (("thl0",) if core.ltempeq else ()) + (("qt0",) if core.lmoist else ()):
        ref = marr(fix, "s003." + k, g.nz)
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]), 1.0 if k == "thl0" else None) <= RUN_TOL, k
    for n in range(core.nsv):
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        assert relerr(got[2:-2, 2:-2, :], carr(fix, f"s003.sv0_{n + 1:02d}", g.nz)[2:-2, 2:-2, :]) <= RUN_TOL, n
    core.close()


@pytest.mark.parametrize("residency", [0, 1, 2])
@pytest.mark.parametrize("name,iexp", sorted({**R_CASES, **D_CASES}.items()))
def test_fortran_driver_with_dropin_modules(name, iexp, residency, tmp_path):
    """The reference's own start-up and loop over the drop-in modules (oracle/_ref/udales_dropin), every residency mode: the start-up's
    `boundary` included (s000), x ghost columns included."""
    from test_gpu_fortran_dropin import run_dropin
    fix = load_fixture(name)
    got = run_dropin(name, iexp, "run", tmp_path, residency)
    checked = 0
    for key, ref in fix.items():
        if ".sv0_" in key:      # (c-arrays: two ghost cells; the interior rows and levels with their four ghost columns)
            assert relerr(got[key].data[2:-2, 2:-2, :], ref.data[2:-2, 2:-2, :]) <= RUN_TOL, key
            checked += 1
            continue
        if "." not in key or key.split(".")[1] not in ("u0", "v0", "w0", "pres0", "um", "vm", "wm", "thl0", "thlm", "qt0", "qtm"):
            continue
        a, b = got[key].data[1:-1], ref.data[1:-1]
        sc = 1.0 if (key.split(".")[1].startswith("thl") or np.abs(b).max() == 0) else None
        assert relerr(nocorner(a), nocorner(b), sc) <= RUN_TOL, key
        checked += 1
    assert checked >= 20
    assert abs(got["s000.uouttot"].data[0] - fix["s000.uouttot"].data[0]) <= 1e-13


@pytest.mark.parametrize("name,iexp", [("run_xopen_16x8x12s", 91), ("run_xopen_thl_16x8x12s", 101), ("run_xdriver_16x8x12s", 96), ("run_xdriver_ibm_16x12x10", 98),
                                       ("run_xopen_ibm_moist_16x12x10", 108), ("run_xdriver_moist_16x12x10", 110), ("run_xdriver_scal_16x8x12s", 113)])
@pytest.mark.parametrize("residency", [2, 0])
def test_through_the_reference_program(name, iexp, residency, tmp_path):
    """u-dales_amd/bin/udales_full_dropin -- the reference's own program.f90, start-up, time loop and writerestartfiles over the
    drop-in modules -- on a BCxm = 2 deck: the restart files it writes against the all-reference run's dump, x ghost columns
    (the outlet's state) included."""
    import os
    from test_full_reference import run_full
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "u-dales_amd", "bin", "udales_full_dropin")
    if not os.path.exists(exe):
        pytest.skip("u-dales_amd/bin/udales_full_dropin not built (needs the reference sources + flang)")
    (tmp_path / "dev").mkdir()
    fix, last, rs, _ = run_full(name, iexp, tmp_path / "dev", exe=exe, env=dict(os.environ, UDC_RESIDENCY=str(residency)))
    nz = int(fix["meta"].data[2])
    ref = {k: fix[f"{last}.{k}"].data for k in ("u0", "v0", "w0", "pres0") + (("thl0",) if f"{last}.thl0" in fix else ()) + (("qt0",) if f"{last}.qt0" in fix else ())}
    if name in D_CASES:
        # (the program stopped by `runtime` skips the last step's drivergen, src/moddriver.f90:216, which the fixture's driver makes:
        #  the all-reference program run the same way is the counterpart; it travels with the snapshot)
        from test_full_reference import FULL
        if not os.path.exists(FULL):
            pytest.skip("oracle/_ref/udales_full not built")
        (tmp_path / "ref").mkdir()
        ref = run_full(name, iexp, tmp_path / "ref", exe=FULL)[2]
    for k in [q for q in ("u0", "v0", "w0", "pres0", "thl0", "qt0") if f"{last}.{q}" in fix]:
        a, b = rs[k][1:nz + 1], ref[k][1:nz + 1]
        assert relerr(nocorner(a), nocorner(b), 1.0 if k == "thl0" else None) <= RUN_TOL, k
    assert np.abs(fix[f"{last}.v0"].data[1:nz + 1, 1:-1, -1] - fix[f"{last}.v0"].data[1:nz + 1, 1:-1, -2]).max() > 1e-4


@pytest.mark.parametrize("shape,sgs,stretch,floor", [
    ((20, 12, 10), 2, 1.00, False),      # 2 itot = 40: radix-5 lines in the doubled solve
    ((64, 48, 40), 2, 1.03, True),       # Vreman, stretched z, the floor
    ((32, 16, 24), 1, 1.05, True),       # Smagorinsky
    ((8, 8, 6), 0, 1.00, False),         # the smallest row the library takes, DNS
    ((126, 8, 6), 1, 1.10, False),       # itot + 2 = 128: the ghost columns end a full tile
])
def test_against_oracle_seeded(shape, sgs, stretch, floor):
    """Six substeps against the CPU oracle's restatement of the branch (oracle/udcore_oracle.c orc_set_open_x; pinned on the reference by
    tests/test_oracle_open_x.py) on seeded random fields with a sheared inflow, at sizes and line lengths the fixtures do not have;
    x ghost columns compared."""
    err = open_x_vs_oracle(shape, sgs, stretch, floor, nsub=6, seed=shape[0] * 100 + shape[2], route="fused")
    assert max(err.values()) <= RUN_TOL, err


def open_x_vs_oracle(shape, sgs, stretch, floor, nsub, seed, route):
    """-> {field: error} of the device (route: "fused", "routine", "deferred") against the oracle after nsub substeps (tests/fuzz_open_x.py
    draws its cases through here too)"""
    import ctypes as C
    import oracle_lib as ol
    from udcore.grid import Grid
    from udcore.core import DynCore
    nx, ny, nz = shape
    dz = 0.5 * stretch ** np.arange(nz)
    g = Grid.from_levels(nx, ny, nz, nx * 0.5, ny * 0.4, np.cumsum(dz) - 0.5 * dz)
    uprof = np.ascontiguousarray(np.concatenate(([0.], 0.8 + 0.4 * np.arange(nz) / nz, [0.])))
    vprof = np.ascontiguousarray(np.concatenate(([0.], 0.15 - 0.1 * np.arange(nz) / nz, [0.])))
    wl = np.ascontiguousarray(g.dzf[1:nz + 1] / (g.zh[nz + 1] - g.zh[2]))
    rng = np.random.default_rng(seed)
    st = {}
    for k, prof in (("u0", uprof), ("v0", vprof), ("w0", 0. * uprof)):
        a = np.zeros(g.mshape())
        a[1:nz + 1] = prof[1:nz + 1, None, None] + 0.05 * rng.standard_normal((nz, ny + 2, nx + 2))      # (ghost columns: state)
        a[:, 0, :] = a[:, -2, :]; a[:, -1, :] = a[:, 1, :]
        st[k] = a
    st["w0"][1] = 0.
    st["u0"][nz + 1] = st["u0"][nz]; st["v0"][nz + 1] = st["v0"][nz]
    for k, m in (("u0", "um"), ("v0", "vm"), ("w0", "wm")):
        st[m] = st[k].copy()
    st["pres0"] = np.zeros(g.mshape())
    core = DynCore(g, sgs=sgs, bctopm=3, lbottom=floor, z0=0.03, open_x=(uprof, vprof))
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=sgs, bctopm=3, csz=0.21658244510412, lbottom=floor, z0=0.03)
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
    dq = np.zeros(nz + 2); dq[1:nz + 1] = 2e-4
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], dq[1:nz + 1])
    core.set_open_x_outflow(wl, 1.0, hold_first=True)
    o.L.orc_set_open_x(1, ol.ptr(uprof), ol.ptr(vprof))
    o.L.orc_set_open_x_outflow(ol.ptr(wl), C.c_double(1.0), 1)
    err = {}
    try:
        ost = {k: v.copy() for k, v in st.items()}
        for k in ("up", "vp", "wp", "ekm", "ekh", "p", "pup", "pvp", "pwp"):
            ost[k] = np.zeros(g.mshape())
        ost["dpdxl"], ost["dpdyl"] = dp, dq
        # both start with a `boundary` (the oracle's with the substep's rk3coef = 0: the inlet columns, no convection)
        core.halos(); core.boundary()
        o.call("orc_boundary", ost["u0"], ost["v0"], ost["w0"], ost["um"], ost["vm"], ost["wm"], None, None)
        o.call("orc_boundary_open_x", 0., ost["u0"], ost["v0"], ost["w0"], ost["um"], ost["vm"], ost["wm"])
        dt = 0.05
        if route == "deferred":
            core.set_deferred(True)
        for isub in range(nsub):
            if route == "fused":
                core.substep(isub % 3 + 1, dt, with_forces=True)
            else:
                core.tstep_update(dt)
                core.advection(); core.subgrid(); core.bottom(); core.forces(); core.poisson()
                core.tstep_integrate(); core.halos(); core.boundary()
            o.substep(ost, isub % 3 + 1, dt)
        for k in ("u0", "v0", "w0", "pres0", "um", "vm", "wm"):
            err[k] = relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1]))
        ref = ost["w0"]
        err["w0(ke+1)"] = relerr(nocorner(core.download("w0"))[nz + 1], nocorner(ref)[nz + 1], np.abs(ref).max())
        err["divergence"] = core.divergence()[0] * 1e2      # (< 1e-11)
    finally:
        o.L.orc_set_open_x(0, None, None)
        o.L.orc_set_open_x_outflow(None, C.c_double(0.), 0)
        core.close()
    return err


def test_full_size_properties_256():
    """BASELINE.json's 256^3 with open x boundaries: size-independent properties -- the projection leaves the interior divergence-free
    (the outlet column included: u(ie+1) is part of the last cell's balance), the inlet column stays at the profile to the bit, the
    outlet's planes are what the arrays hold, and the routine-by-routine order gives what the fused substep gives."""
    from udcore.grid import Grid
    from udcore.core import DynCore
    from udcore import lib as L
    n = 256
    g = Grid.uniform(n, n, n, 2.0, 2.0, 2.0)
    prof = np.concatenate(([0.], 1.0 + 0.2 * np.arange(n) / n, [0.]))
    rng = np.random.default_rng(11)
    sh = g.mshape()
    st = {k: m + 0.05 * (rng.random(sh) - 0.5) for k, m in (("u0", 1.1), ("v0", 0.1), ("w0", 0.))}
    st["w0"][:2] = 0.
    out = []
    for fused in (True, False):
        core = DynCore(g, sgs=L.SGS_SMAGORINSKY, bctopm=3, lbottom=True, z0=0.05, open_x=(prof, 0.1 * (prof > 0)))
        core.set_forcing(np.zeros(n), np.zeros(n))
        core.set_open_x_outflow(g.dzf[1:n + 1] / (g.zh[n + 1] - g.zh[2]), 1.1)
        for k, a in st.items():
            core.upload(k, a); core.upload(k.replace("0", "m"), a)
        core.halos(); core.boundary()
        dt = 0.05
        for isub in range(1, 7):
            if fused:
                core.substep((isub - 1) % 3 + 1, dt)
            else:
                core.tstep_update(dt)
                core.advection(); core.subgrid(); core.bottom(); core.forces(); core.poisson()
                core.tstep_integrate(); core.halos(); core.boundary()
        u, v = core.download("u0"), core.download("v0")
        divmax, _ = core.divergence()
        assert divmax < 1e-12 * 1.1 / g.dx * 10
        assert np.array_equal(u[1:n + 2, 1:-1, 1], np.broadcast_to(prof[1:n + 2, None], (n + 1, n)))      # u(ib) = uprof(k), k = kb .. ke+1
        assert np.array_equal(u[1:n + 1, 1:-1, 0], 2 * prof[1:n + 1, None] - u[1:n + 1, 1:-1, 2])          # the inlet ghost mirrors about it
        assert np.abs(u[1:n + 1, 1:-1, -1] - u[1:n + 1, 1:-1, -2]).max() > 1e-4                             # an outlet that carries something
        out.append((u, v, core.download("pres0")))
        core.close()
    # (the fused substep runs the pressure-total form, the routines the reference's: pres0 is a fresh solve on one side and a sum of six
    #  increments on the other -- 9e-11 at this size, profiles/r06/open_lid_ptotal_ab.txt; the velocities agree to 2e-14)
    for (a, b), tol in zip(zip(*out), (1e-12, 1e-12, 1e-9)):
        assert relerr(nocorner(a[1:-1]), nocorner(b[1:-1])) <= tol


def test_reductions_leave_the_ghost_columns_out():
    """Courant / diffusion numbers (tstep_update), slab averages (diagfld) and the divergence check (chkdiv) run over ib..ie: whatever the
    two ghost columns of the device row hold (here: something huge) does not reach them."""
    import ctypes as C
    d, core = make_core("k_xopen_16x8x12", 90)
    fix = load_fixture("k_xopen_16x8x12")
    g, nz = core.g, core.g.nz
    st = {k: marr(fix, "in." + k, nz) for k in ("u0", "v0", "w0", "um", "vm", "wm", "ekm", "ekh")}
    st = {k: np.nan_to_num(a) for k, a in st.items()}
    ref_div = None
    for poke in (False, True):
        for k, a in st.items():
            b = a.copy()
            if poke and k in ("um", "vm", "wm", "ekm", "ekh", "v0", "w0"):
                b[:, :, 0] = 1e3; b[:, :, -1] = 1e3
            core.upload(k, b)
        dt = 0.25
        c, dd = C.c_double(), C.c_double()
        assert core.lib.udc_tstep_maxima(core.h, C.c_double(dt), C.byref(c), C.byref(dd)) == 0
        um, vm, wm, ekm, ekh = (interior(st[k]) for k in ("um", "vm", "wm", "ekm", "ekh"))
        dzh = g.dzh[1:nz + 1][:, None, None]
        cour = ((np.abs(um) / g.dx + np.abs(vm) / g.dy + np.abs(wm) / dzh) * dt).max()
        f = (1. / dzh ** 2 + 1. / g.dx ** 2 + 1. / g.dy ** 2) * dt
        dif = max(1e-5, (ekm * f).max(), (ekh * f).max())
        assert abs(c.value - cour) <= 1e-12 * cour and abs(dd.value - dif) <= 1e-12 * dif, poke
        av = core.slab_average("um")
        assert np.abs(av[1:nz + 1] - um.mean(axis=(1, 2))).max() <= 1e-13
        div = core.divergence()
        if ref_div is None:
            ref_div = div
        else:
            assert div == ref_div      # (v, w at ie+1 / ib-1 are no part of an interior cell's balance; u(ie+1) is, and was not poked)
    core.close()


def test_cli_run_writes_the_reference_state(tmp_path):
    """run_case.py on the BCxm = 2 deck with obstacles and wall functions: cold start, three steps, the restart file it writes."""
    import os, shutil, subprocess, sys
    from common import GOLDEN
    from udcore import restart as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name, iexp = "run_xopen_ibmwf3_16x12x10", 94
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    r = subprocess.run([sys.executable, os.path.join(root, "run_case.py"), f"namoptions.{iexp:03d}", "--steps", "3"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    fix = load_fixture(name)
    n = tuple(int(v) for v in fix["meta"].data[:3])
    got = R.read_initd(os.path.join(tmp_path, R.restart_name(3, 0, iexp)), *n)
    for k in ("u0", "v0", "w0", "pres0"):
        ref = marr(fix, f"s009.{k}", n[2])
        assert relerr(nocorner(got[k][1:-1]), nocorner(ref[1:-1])) <= RUN_TOL, k


@pytest.mark.parametrize("name,iexp", [("run_xdriver_16x8x12s", 96), ("run_xdriver_ibm_16x12x10", 98), ("run_xdriver_scal_16x8x12s", 113)])
def test_cli_run_with_driver_inflow(name, iexp, tmp_path):
    """run_case.py on BCxm = 3 decks: the Python runner reads the precursor's planes itself (udcore/driver.py restates readdriverfile and
    drivergen's interpolation, src/moddriver.f90:213-376, 752-932) and hands them over where the reference's `boundary` calls drivergen;
    cold start, three steps, the restart files against the reference's state -- with the scalars' planes too (BCxT = BCxq = BCxs = 3)."""
    import os, shutil, subprocess, sys
    from common import GOLDEN
    from udcore import restart as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    r = subprocess.run([sys.executable, os.path.join(root, "run_case.py"), f"namoptions.{iexp:03d}", "--steps", "3"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    fix = load_fixture(name)
    n = tuple(int(v) for v in fix["meta"].data[:3])
    got = R.read_initd(os.path.join(tmp_path, R.restart_name(3, 0, iexp)), *n)
    for k in ("u0", "v0", "w0", "pres0") + (("thl0", "qt0") if "s009.qt0" in fix else ()):
        ref = marr(fix, f"s009.{k}", n[2])
        assert relerr(nocorner(got[k][1:-1]), nocorner(ref[1:-1]), 1.0 if k == "thl0" else None) <= RUN_TOL, k
    nsv = int(fix["meta"].data[12])
    if nsv:
        sc = R.read_inits(os.path.join(tmp_path, R.restart_name(3, 0, iexp).replace("initd", "inits")), *n, nsv)
        for q in range(nsv):
            ref = carr(fix, f"s009.sv0_{q + 1:02d}", n[2])[2:n[2] + 2, 2:-2, 1:-1]      # (levels kb .. ke, rows jb .. je; the file keeps one ghost column)
            assert relerr(sc["sv0"][q][1:n[2] + 1, 1:-1, :], ref) <= RUN_TOL, q


@pytest.mark.parametrize("residency,prog_name,name,iexp", [
    (2, "udales_full_dropin", "run_xdriver_ibm_16x12x10", 98), (2, "udales_full_dropin", "run_xdriver_moist_16x12x10", 110),
    (2, "udales_full_dropin", "run_xopen_ibm_thl_16x12x10", 105), (2, "udales_full_dropin", "run_xopen_sv_16x8x12s", 103),
    (0, "udales_full_dropin", "run_xopen_ibm_thl_16x12x10", 105),
    (2, "udales_full_dropin_hoststats", "run_xdriver_ibm_16x12x10", 98), (0, "udales_full_dropin_hoststats", "run_xdriver_ibm_16x12x10", 98),
    (2, "udales_full_dropin_hoststats", "run_xopen_ibm_thl_16x12x10", 105)])
def test_statistics_of_an_inflow_outflow_run(residency, prog_name, name, iexp, tmp_path):
    """Statistics of an inflow / outflow run (the class of examples/950: driver inflow, obstacles, wall functions, tdump + xytdump), every
    record handed to NetCDF against the all-reference program's.  udales_full_dropin: the device's own accumulators (udc_stats.hip) --
    the reference samples between tstep_integrate / halos and `boundary` (src/program.f90:199-214), so stage 3 of the fused substep ends
    ahead of `boundary` there and the drop-in boundary / thermodynamics follow the sample; the slab sums leave the device row's ghost
    columns out.  udales_full_dropin_hoststats: the reference's own modstatsdump on the host arrays, which the drop-ins refresh on exactly
    the sampling steps.
    On a deck with BCxT = 2 that includes the one line of cells the state never shows: xTi_profile overwrites the first interior column with
    the profile in `boundary` (src/modboundary.f90:785-791), so what tstep_integrate left there reaches nothing but the sample -- and at its
    top level that value depends on reassure_fluxtop_boundary (src/modboundary.f90:392-431, inside closurebc) resetting thl0(ib, j, ke+1) to
    the profile between advection and diffusion (k_xo_thl_top)."""
    import os
    from common import BINDIR, GOLDEN
    from refdump import read_ncrec
    from test_full_reference import FULL, run_full
    exe = os.path.join(BINDIR, prog_name)
    if not (os.path.exists(exe) and os.path.exists(FULL)):
        pytest.skip(f"oracle/_ref/udales_full or u-dales_amd/bin/{prog_name} not built")
    out, rst = {}, {}
    for tag, prog in (("ref", FULL), ("dev", exe)):
        d = tmp_path / tag
        d.mkdir()
        txt = open(os.path.join(GOLDEN, "cases", name, f"namoptions.{iexp:03d}")).read()
        txt = txt.replace("&SCALARS", "&OUTPUT\nltdump = .true.\nlxytdump = .true.\nlxydump = .true.\nlytdump = .true.\nlydump = .true.\ntstatsdump = 0.5\ntsample = 0.25\n/\n&SCALARS")
        assert "ltdump" in txt
        rst[tag] = run_full(name, iexp, d, exe=prog, env=dict(os.environ, UDC_RESIDENCY=str(residency)), deck_text=txt)[2]
        out[tag] = {fn: read_ncrec(str(d / fn)) for fn in sorted(os.listdir(d)) if fn.endswith(".nc") and "dump" in fn}
    assert set(out["ref"]) == set(out["dev"]) and len(out["ref"]) >= 2, (sorted(out["ref"]), sorted(out["dev"]))
    # the restart file of the same run: the state AFTER the last `boundary` (the sample saw the one ahead of it), x ghost columns included
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0"):
        if k in rst["ref"] and np.abs(rst["ref"][k]).max() > 0.:
            assert relerr(nocorner(rst["dev"][k][1:-1]), nocorner(rst["ref"][k][1:-1]), 1.0 if k == "thl0" else None) <= RUN_TOL, k
    checked, bad = 0, []
    import re
    m = re.search(r"^\s*nsv\s*=\s*(\d+)", txt, re.M)
    nsv = int(m.group(1)) if m else 0
    for fn, ref in out["ref"].items():
        dev = out["dev"][fn]
        assert list(ref) == list(dev), fn
        for var, recs in ref.items():
            assert len(recs) == len(dev[var]) >= 1, (fn, var)
            # ytdump's / ydump's rows of fields the deck does not carry: the reference never assigns them (src/modstatsdump.f90:1109-1131,
            # 1483-1505 sit under ltempeq / lmoist / nsv > n) and writes what the arrays happen to hold
            if fn.startswith(("ytdump", "ydump")) and (("thl" in var and "ltempeq" not in txt) or ("qt" in var and "lmoist" not in txt)
                                                       or any(f"sca{q}" in var and nsv < q for q in (1, 2, 3))):
                continue
            for (s0, a), (s1, b) in zip(recs, dev[var]):
                assert s0 == s1 and a.shape == b.shape, (fn, var)
                hole = a < -900.
                if not np.array_equal(hole, b < -900.):
                    bad.append((fn, var + " (holes)", s0, float(np.abs(a - b).max())))
                    continue
                sc = max(np.abs(a[~hole]).max() if (~hole).any() else 0., 1e-3 if var.startswith("p") else 1e-6)
                err = np.abs(a - b)[~hole].max(initial=0.)
                if err > 1e-8 * sc:
                    bad.append((fn, var, s0, float(err / sc)))
                checked += 1
    if bad:
        pytest.fail("\n".join(f"{fn} {var} {s0}: {e:.3e}" for fn, var, s0, e in bad))
    assert checked >= 30


def test_warm_start_applies_the_start_up_boundary(tmp_path):
    """A warm start of an inflow / outflow run (udcore.restart.load_restart): the reference's start-up runs `boundary` on the fields of the
    restart file too (src/program.f90:118) -- the outlet's v, w take one more convective step with rk3step = 0 and the file's dt, at the speed
    diagfld's averages of the file's u give (src/modboundary.f90:141-160, 914), the inlet's ghost column is re-derived from the profile.
    The same expressions in numpy on the file's arrays."""
    from udcore import cold_start, restart as R
    name, iexp = "run_xopen_16x8x12s", 91
    d, core = make_core(name, iexp)
    g = core.g
    dt = float(d.get("RUN", "dtmax"))
    core.load_state(cold_start(g, d, nsv=0, pre_boundary=True))
    core.halos(); core.start_up(dtmax=dt)
    for isub in range(1, 4):
        core.substep(isub, dt)
    R.save_restart(core, str(tmp_path), iexp, 1, dt, dt)
    core.close()
    f = R.read_initd(str(tmp_path / R.restart_name(1, 0, iexp)), g.nx, g.ny, g.nz)
    d2, warm = make_core(name, iexp)
    R.load_restart(warm, str(tmp_path), iexp, 1)
    wl = wlev(g)
    uout = float(np.sum(f["u0"][1:g.nz + 1, 1:-1, 1:-1].mean(axis=(1, 2)) * wl))
    for k in ("v0", "w0"):
        a = f[k]
        want = a[:, :, -1] - (a[:, :, -1] - a[:, :, -2]) / g.dx * (f["dt"] / 4.) * uout
        got = warm.download(k)
        lev = slice(1, g.nz + 1)
        assert np.abs(got[lev, 1:-1, -1] - want[lev, 1:-1]).max() <= 1e-13, k
        assert np.abs(want[lev, 1:-1] - a[lev, 1:-1, -1]).max() > 1e-6      # (the step moved the outlet)
        assert np.array_equal(got[lev, 1:-1, 1:-1], a[lev, 1:-1, 1:-1])
    u = warm.download("u0")
    prof = np.asarray(d.u, dtype=float)[:g.nz]
    assert np.array_equal(u[1:g.nz + 1, 1:-1, 1], np.broadcast_to(prof[:, None], (g.nz, g.ny)))
    warm.close()


def test_example_950_through_the_reference_program(tmp_path):
    """examples/950 of the reference -- inflow from a precursor run's driver files (BCxm = 3), 256 x 128 x 128, 6612 facets with wall
    functions, a prescribed volume flow, the adaptive time step, tdump / xytdump / fielddump switched on -- as a user runs it, under the
    untouched program.f90 with the drop-in modules and the reference's own statsdump (udales_full_dropin_hoststats), device resident.
    The example ships without its precursor's planes: tests/golden/make_golden.py (make_example_950) derives a precursor deck from the
    example's own, runs it through the reference and keeps the four records with the case.  Deck changes: one rank, three steps,
    driverstore = 4.  Golden: the same through the all-reference executable -- the clock after each step, the restart file (slab means,
    rms, every 8th point, the two outlet columns)."""
    import gzip, os, subprocess, sys
    from common import BINDIR, GOLDEN
    from udcore import restart
    sys.path.insert(0, GOLDEN)
    from make_golden import ex950_decks
    exe = os.path.join(BINDIR, "udales_full_dropin_hoststats")
    if not os.path.exists(exe):
        pytest.skip("u-dales_amd/bin/udales_full_dropin_hoststats not built")
    fix = load_fixture("full_example_950")
    cdir = os.path.join(GOLDEN, "cases", "example_950")
    for fn in os.listdir(cdir):
        with gzip.open(os.path.join(cdir, fn), "rb") as f, open(tmp_path / fn[:-3], "wb") as o:
            o.write(f.read())
    deck = tmp_path / "namoptions.950"
    deck.write_text(ex950_decks(deck.read_text())[0])
    r = subprocess.run(f"ulimit -s unlimited; exec {exe} namoptions.950", shell=True, cwd=tmp_path, capture_output=True, text=True,
                       timeout=1500, executable="/bin/bash", env=dict(os.environ, UDC_RESIDENCY="2"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    mon = np.atleast_1d(np.loadtxt(tmp_path / "monitor000.txt"))
    ref = fix["monitor"].data
    assert len(mon) == len(ref) == 3
    np.testing.assert_allclose(mon, ref, rtol=2e-6)            # (the monitor file holds six digits)
    rst = [f for f in os.listdir(tmp_path) if f.startswith("initd") and f.endswith(".950")]
    assert len(rst) == 1 and float(rst[0][5:13]) == fix["rst.time"].data[2]
    nx, ny, nz = 256, 128, 128
    rs = restart.read_initd(str(tmp_path / rst[0]), nx, ny, nz)
    np.testing.assert_allclose((rs["timee"], rs["dt"]), fix["rst.time"].data[:2], rtol=1e-9)      # the adaptive step's history
    for k in ("u0", "v0", "w0", "pres0"):
        a = rs[k][1:nz + 1, 1:ny + 1, :]
        sc = np.abs(fix[f"rst.{k}.pts"].data).max()
        assert np.abs(a[::8, ::8, ::8] - fix[f"rst.{k}.pts"].data).max() <= 1e-8 * sc, k
        assert np.abs(a[:, :, -2:] - fix[f"rst.{k}.out"].data).max() <= 1e-8 * sc, k
        assert np.abs(a.mean(axis=(1, 2)) - fix[f"rst.{k}.mean"].data).max() <= 1e-9 * sc, k
        assert np.abs(np.sqrt((a ** 2).mean(axis=(1, 2))) - fix[f"rst.{k}.rms"].data).max() <= 1e-9 * sc, k


def test_what_open_x_does_not_offer_is_refused():
    from udcore import lib as L
    d, core = make_core("k_xopen_16x8x12", 90)
    with pytest.raises(L.UdcError, match="central scheme"):
        core.set_tempeq(iadv_thl=7)
    with pytest.raises(L.UdcError, match="open x"):
        core.set_masscorr_outflow(True, 1.0)
    core.close()
    import udcore
    d = read_deck(deck_path("k_xopen_16x8x12", 90))
    d.nml["BC"]["BCxm"] = 4
    with pytest.raises(ValueError, match="BCxm"):
        udcore.from_deck(d)


def test_runner_statistics_of_an_inflow_outflow_run(tmp_path):
    """tdump + xytdump of a BCxm = 2 deck through the Python runner (run_case.py: udc_stats_sample between the fused stage-3 substep, which
    ends ahead of `boundary` on such a handle, and udc_boundary) against the records of the all-reference program."""
    import os, shutil, subprocess, sys
    from common import GOLDEN
    from refdump import read_ncrec
    from test_full_reference import FULL, run_full
    if not os.path.exists(FULL):
        pytest.skip("oracle/_ref/udales_full not built")
    name, iexp = "run_xopen_16x8x12s", 91
    txt = open(os.path.join(GOLDEN, "cases", name, f"namoptions.{iexp:03d}")).read()
    assert "&SCALARS" in txt
    txt = txt.replace("&SCALARS", "&OUTPUT\nltdump = .true.\nlxytdump = .true.\nlytdump = .true.\ntstatsdump = 0.5\ntsample = 0.25\n/\n&SCALARS")
    (tmp_path / "ref").mkdir(); (tmp_path / "dev").mkdir()
    run_full(name, iexp, tmp_path / "ref", exe=FULL, deck_text=txt)
    ref = {fn: read_ncrec(str(tmp_path / "ref" / fn)) for fn in sorted(os.listdir(tmp_path / "ref")) if fn.endswith(".nc") and "dump" in fn}
    nrec = len(ref[f"tdump.000.000.{iexp:03d}.nc"]["ut"])
    assert nrec >= 1
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path / "dev")
    (tmp_path / "dev" / f"namoptions.{iexp:03d}").write_text(txt)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "run_case.py"), f"namoptions.{iexp:03d}", "--steps", str(2 * nrec)],
                       cwd=tmp_path / "dev", capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    z = np.load(tmp_path / "dev" / f"tdump.{iexp:03d}.npz")
    checked = 0
    for var, recs in ref[f"tdump.000.000.{iexp:03d}.nc"].items():
        if var == "time" or f"{var}.0" not in z.files:
            continue
        for q, (_, a) in enumerate(recs):
            sc = max(np.abs(a).max(), 1e-3 if var.startswith("p") else 1e-6)
            assert np.abs(z[f"{var}.{q}"] - a).max() <= 1e-8 * sc, (var, q)
            checked += 1
    assert checked >= 10 * nrec
    x = np.load(tmp_path / "dev" / f"xytdump.{iexp:03d}.npz")
    for var, recs in ref[f"xytdump.{iexp:03d}.nc"].items():
        if var == "time" or var not in x.files:
            continue
        for q, (_, a) in enumerate(recs):
            sc = max(np.abs(a[a > -900.]).max(initial=0.), 1e-3 if var.startswith("p") else 1e-6)
            assert np.abs(x[var][q] - a).max() <= 1e-8 * sc, (var, q)
            checked += 1
    assert checked >= 20 * nrec, checked
    y = np.load(tmp_path / "dev" / f"ytdump.{iexp:03d}.npz")
    for var, recs in ref[f"ytdump.{iexp:03d}.nc"].items():
        if var == "time" or var not in y.files or any(t in var for t in ("thl", "qt", "sca")):      # (rows the reference never assigns on this deck)
            continue
        for q, (_, a) in enumerate(recs):
            sc = max(np.abs(a[a > -900.]).max(initial=0.), 1e-6)
            assert y[var][q].shape == a.shape and np.abs(y[var][q] - a).max() <= 1e-8 * sc, (var, q)
            checked += 1
    assert checked >= 28 * nrec, checked


def test_sample_gap_is_the_state_ahead_of_boundary():
    """udc_set_open_x_sample_gap: the fused stage-3 substep that ends ahead of `boundary`, followed by udc_boundary, leaves what the substep
    with its `boundary` inside leaves -- and ahead of it the inlet's first thl column is NOT yet the profile while the interior of u equals
    the final one; one-shot: the substeps after it are whole again."""
    name, iexp = "run_xopen_thl_16x8x12s", 101
    out = {}
    for gap in (False, True):
        d, core = make_core(name, iexp)
        g = core.g
        dt = float(d.get("RUN", "dtmax"))
        from udcore import cold_start
        core.load_state(cold_start(g, d, nsv=0, pre_boundary=True))
        core.halos(); core.start_up(dtmax=dt)
        for step in range(2):
            for rk in (1, 2, 3):
                if gap and rk == 3 and step == 0:
                    core.set_open_x_sample_gap()
                core.substep(rk, dt)
                if gap and rk == 3 and step == 0:
                    ahead = {k: core.download(k) for k in ("u0", "thl0")}
                    core.boundary()
                    out["after"] = {k: core.download(k) for k in ("u0", "thl0")}
        out[gap] = {k: core.download(k) for k in ("u0", "v0", "w0", "thl0", "pres0")}
        core.close()
    for k in out[True]:
        assert np.array_equal(out[True][k], out[False][k]), k
    nz = out[True]["u0"].shape[0] - 2
    prof = np.asarray(d.thl, dtype=float)[:nz]
    assert np.array_equal(out["after"]["thl0"][1:nz + 1, 1:-1, 1], np.broadcast_to(prof[:, None], out["after"]["thl0"][1:nz + 1, 1:-1, 1].shape))
    assert np.abs(ahead["thl0"][1:nz + 1, 1:-1, 1] - prof[:, None]).max() > 1e-9      # (the integration's value, not the profile)
    assert np.array_equal(ahead["u0"][1:nz + 1, 1:-1, 2:-2], out["after"]["u0"][1:nz + 1, 1:-1, 2:-2])
