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
    core.load_state(cold_start(core.g, d))
    ls = LevelForcings(core, d)
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
