"""GPU parity tests: libudcore (through the C ABI) against
  (a) golden vectors produced by the reference's own Fortran (tests/golden),
  (b) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at BASELINE.json's 256^3 (divergence-free projection,
      periodic-shift equivariance).

Floating-point tolerances (real(8) everywhere, stated as error / max|reference field|):
  KERNEL_TOL 1e-11 : one routine; differences come from FMA contraction and, for the Poisson
                     solve, from rocFFT vs the FFTW-convention DFT used on the CPU side.
  RUN_TOL    1e-9  : a handful of chained substeps.
The north star's acceptance bar is 1e-6 after 100 steps (tests/test_gpu_long.py).
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
from common import (KERNEL_CASES, RUN_CASES, carr, deck_path, interior, load_fixture, marr, nocorner,
                    relerr)
from udcore import read_deck, sgs_from_deck, cold_start
from udcore.grid import Grid
from udcore import lib as L

pytestmark = pytest.mark.gpu

KERNEL_TOL = 1e-11
RUN_TOL = 1e-9


def core_from_deck(name, iexp):
    import udcore
    d = read_deck(deck_path(name, iexp))
    return d, udcore.from_deck(d)


def upload_inputs(core, fix, tag, g, nsv):
    for k in ("u0", "v0", "w0", "um", "vm", "wm", "pres0"):
        core.upload(k, marr(fix, f"{tag}.{k}", g.nz))
    for k in ("ekm", "ekh"):
        if f"{tag}.{k}" in fix:
            core.upload(k, np.nan_to_num(marr(fix, f"{tag}.{k}", g.nz)))
    for n in range(nsv):
        c = carr(fix, f"{tag}.sv0_{n + 1:02d}", g.nz)
        core.upload(L.scalar_field(L.SV0, n), c)
        key = f"{tag}.svm_{n + 1:02d}"
        core.upload(L.scalar_field(L.SVM, n), carr(fix, key, g.nz) if key in fix else c)
    if f"{tag}.thl0" in fix:
        core.upload("thl0", marr(fix, f"{tag}.thl0", g.nz))
        core.upload("thlm", marr(fix, f"{tag}.thlm", g.nz))
    if f"{tag}.qt0" in fix:
        core.upload("qt0", marr(fix, f"{tag}.qt0", g.nz))
        core.upload("qtm", marr(fix, f"{tag}.qtm", g.nz))
    if "thm.presf" in fix and tag == "in":      # what the reference's last thermodynamics call left behind
        core.thermo_state({n: np.concatenate(([0.], fix["thm." + n].data)) if "thm." + n in fix else np.zeros(g.nz + 2)
                           for n in core.TH_TABLES})
    if "thm.ql0" in fix and tag == "in" and f"{tag}.e120" in fix:      # moist dthvdz of the one-equation closure reads ql0
        core.upload("ql0", marr(fix, "thm.ql0", g.nz))
    if f"{tag}.e120" in fix:
        core.upload("e120", marr(fix, f"{tag}.e120", g.nz))
        core.upload("e12m", marr(fix, f"{tag}.e12m", g.nz))


@pytest.mark.parametrize("name,iexp", sorted(KERNEL_CASES.items()))
def test_each_routine_matches_reference(name, iexp):
    """udc_advection / udc_subgrid / udc_forces / udc_poisson / udc_tstep_integrate, one at a time,
    on the inputs and outputs the reference's own routines produced."""
    fix = load_fixture(name)
    d, core = core_from_deck(name, iexp)
    g, nsv, nz = core.g, core.nsv, core.g.nz
    upload_inputs(core, fix, "in", g, nsv)
    if int(d.get("DYNAMICS", "iadv_thl")) == 7:
        core.halos()          # kappa on thl reads two lateral ghost cells; the reference's m-arrays carry one
    zero = np.zeros(g.mshape())

    thl = "in.thl0" in fix
    tke = "in.e120" in fix
    qt = "in.qt0" in fix

    def zero_tend():
        for k in ("up", "vp", "wp") + (("thlp",) if thl else ()) + (("e12p",) if tke else ()) + (("qtp",) if qt else ()):
            core.upload(k, zero)
        for n in range(nsv):
            core.upload(L.scalar_field(L.SVP, n), np.zeros(g.cshape()))

    zero_tend()
    core.advection()
    for k in ("up", "vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "adv." + k, nz))) <= KERNEL_TOL, k
    for n in range(nsv):
        got = core.download(L.scalar_field(L.SVP, n), halo=2)
        assert relerr(interior(got, 2), interior(carr(fix, f"adv.svp_{n + 1:02d}", nz), 2)) <= KERNEL_TOL

    if thl:       # advecc_2nd
        assert relerr(interior(core.download("thlp")), interior(marr(fix, "adv.thlp", nz))) <= KERNEL_TOL
    if qt:
        assert relerr(interior(core.download("qtp")), interior(marr(fix, "adv.qtp", nz))) <= KERNEL_TOL
    # e12: the reference's x ghost columns of e120 are stale (never refreshed, src/modboundary.f90:527-536) while the
    # device wraps the index, so the columns next to the x boundary are not comparable
    def inx(a):
        return interior(a)[:, :, 1:-1]
    if tke:
        assert relerr(inx(core.download("e12p")), inx(marr(fix, "adv.e12p", nz))) <= KERNEL_TOL

    zero_tend()
    core.subgrid()
    if thl:       # top row re-imposed with the new ekh (reassure_fluxtop_boundary), then diffc
        assert relerr(core.download("thl0")[1:], marr(fix, "sub.thl0", nz)[1:], 1.0) <= KERNEL_TOL
        assert relerr(interior(core.download("thlp")), interior(marr(fix, "sub.thlp", nz))) <= KERNEL_TOL
    if qt:
        assert relerr(core.download("qt0")[1:], marr(fix, "sub.qt0", nz)[1:]) <= KERNEL_TOL
        assert relerr(interior(core.download("qtp")), interior(marr(fix, "sub.qtp", nz))) <= KERNEL_TOL
    if tke:       # closure from e120, then diffe + sources
        assert relerr(inx(core.download("e12p")), inx(marr(fix, "sub.e12p", nz))) <= KERNEL_TOL
    ekm, ekh = core.download("ekm"), core.download("ekh")
    assert relerr(ekm, marr(fix, "sub.ekm", nz)) <= KERNEL_TOL          # ghosts included (closurebc)
    assert relerr(ekh, marr(fix, "sub.ekh", nz)) <= KERNEL_TOL
    assert relerr(core.download("u0"), marr(fix, "sub.u0", nz)) <= KERNEL_TOL   # reassure_fluxtop row
    for k in ("up", "vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "sub." + k, nz))) <= KERNEL_TOL, k
    for n in range(nsv):
        got = core.download(L.scalar_field(L.SVP, n), halo=2)
        assert relerr(interior(got, 2), interior(carr(fix, f"sub.svp_{n + 1:02d}", nz), 2)) <= KERNEL_TOL
    if "bot.up" in fix:       # floor wall function on top of the subgrid tendencies (`bottom` -> wfmneutral)
        core.bottom_diagnostics(True)
        core.bottom()
        for k in ("up", "vp"):
            assert relerr(interior(core.download(k)), interior(marr(fix, "bot." + k, nz))) <= KERNEL_TOL, k
        # tau_x, tau_y, thl_flux: what `bottom` added (src/modibm.f90:2094-2097: `up - up_before`, same subtraction here),
        # from the reference's own before / after dumps; nonzero on the k = kb plane only
        for nm, k in (("tau_x", "up"), ("tau_y", "vp")) + ((("thl_flux", "thlp"),) if thl else ()):
            inc = interior(marr(fix, "bot." + k, nz))[0] - interior(marr(fix, "sub." + k, nz))[0]
            got = core.bottom_diag(nm)
            assert np.abs(inc).max() > 0 and np.abs(got - inc).max() <= 1e-12 * np.abs(inc).max(), nm
            assert np.all(interior(marr(fix, "bot." + k, nz))[1:] == interior(marr(fix, "sub." + k, nz))[1:])
        assert relerr(interior(core.download("up")), interior(marr(fix, "sub.up", nz))) > 1e-6
        if thl:   # floor flux wtsurf
            assert relerr(interior(core.download("thlp")), interior(marr(fix, "bot.thlp", nz))) <= KERNEL_TOL
        if qt:    # floor flux wqsurf
            assert relerr(interior(core.download("qtp")), interior(marr(fix, "bot.qtp", nz))) <= KERNEL_TOL
            assert relerr(interior(core.download("qtp")), interior(marr(fix, "sub.qtp", nz))) > 1e-6

    # full tendency as the reference driver had it, then forces (already inside pre.*), poisson
    zero_tend()
    core.advection()
    core.subgrid()
    core.bottom()
    core.coriolis()
    core.forces()
    core.rk3step, core.dt = int(fix["rk3"].data[0]), float(fix["rk3"].data[1])
    if "frc.up" in fix:      # masscorr: tendencies before / after the volume-flow correction
        for k in ("up", "vp"):
            assert relerr(interior(core.download(k)), interior(marr(fix, "frc." + k, nz))) <= KERNEL_TOL, k
    core.masscorr()
    if "src0.svp_01" in fix:      # scalsource: Gaussian point / line sources evaluated by udcore/sources.py
        core.scalsource()
        for n in range(nsv):
            got = core.download(L.scalar_field(L.SVP, n), halo=2)
            ref = carr(fix, f"pre.svp_{n + 1:02d}", nz)
            assert relerr(interior(got, 2), interior(ref, 2)) <= KERNEL_TOL
            assert relerr(interior(ref, 2), interior(carr(fix, f"src0.svp_{n + 1:02d}", nz), 2)) > 1e-3
    for k in ("up", "vp", "wp") + (("thlp",) if thl else ()) + (("qtp",) if qt else ()):
        assert relerr(interior(core.download(k)), interior(marr(fix, "pre." + k, nz))) <= KERNEL_TOL, k
    if tke:
        assert relerr(inx(core.download("e12p")), inx(marr(fix, "pre.e12p", nz))) <= KERNEL_TOL
    core.poisson()
    # p solves lap(p) = div(up + um/rk3coef): its round-off floor is set by the O(U/rk3coef) terms
    # that cancel in the divergence, so errors are measured against the natural pressure scale
    # U*dx/rk3coef (or |p| itself when that is larger), not against a possibly tiny |p|.
    rk3coef = core.dt / (4. - core.rk3step)
    pnat = 1e-2 * np.abs(marr(fix, "in.um", nz)).max() * g.dx / rk3coef
    pscale = max(np.abs(marr(fix, "poi.p", nz)).max(), np.abs(marr(fix, "poi.pres0", nz)).max(), pnat)
    assert relerr(interior(core.download("p")), interior(marr(fix, "poi.p", nz)), pscale) <= KERNEL_TOL
    pres = core.download("pres0")
    assert relerr(nocorner(pres[1:-1]), nocorner(marr(fix, "poi.pres0", nz)[1:-1]), pscale) <= KERNEL_TOL
    for k in ("up", "vp", "wp"):
        assert relerr(interior(core.download(k)), interior(marr(fix, "poi." + k, nz))) <= KERNEL_TOL, k
    lid = int(d.get("BC", "BCtopm")) == 3
    if lid:       # the open lid: bcpup's + tderive's row wp(ke+1) (src/modboundary.f90:1234-1243, src/modpois.f90:1058-1069)
        ref = marr(fix, "poi.wp", nz)
        assert relerr(core.download("wp")[nz + 1, 1:-1, 1:-1], ref[nz + 1, 1:-1, 1:-1], np.abs(ref).max()) <= KERNEL_TOL
    core.tstep_integrate()
    core.halos()
    core.boundary()
    if lid:       # ... and tstep_integrate's plane w0(ke+1), which `boundary` leaves alone (src/modtstep.f90:270-286, src/modboundary.f90:191-200)
        ref = marr(fix, "out.w0", nz)
        assert relerr(nocorner(core.download("w0"))[nz + 1], nocorner(ref)[nz + 1], np.abs(ref).max()) <= KERNEL_TOL
    for k in ("u0", "v0", "w0", "um", "pres0") + (("thl0", "thlm") if thl else ()) + (("qt0", "qtm") if qt else ()):
        ref = marr(fix, "out." + k, nz)
        sc = pscale if k == "pres0" else (1.0 if k.startswith("thl") else None)
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]), sc) <= KERNEL_TOL, k
    if tke:       # interior only: the reference never refreshes e120's lateral ghosts (src/modboundary.f90:527-536)
        for k in ("e120", "e12m"):
            assert relerr(inx(core.download(k)), inx(marr(fix, "out." + k, nz))) <= KERNEL_TOL, k
    if "thn.presf" in fix:      # the thermodynamics call that ends the substep (src/program.f90:214)
        core.thermodynamics()
        got = core.thermo_state()
        for n in ("presf", "presh", "exnf", "exnh", "thvh", "ql0av"):
            ref = fix["thn." + n].data
            lo = 1 if n == "thvh" else 0          # thvh(kb) is a diagnostic override the device does not keep
            hi = nz if n == "thvh" else nz + 1    # thvh(ke+kh) is never used (forces: k = kb+1..ke)
            assert np.abs(got[n][1 + lo:hi + 1] - ref[lo:hi]).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300), n
        assert fix["thn.ql0av"].data.max() > 1e-4
    core.close()


@pytest.mark.parametrize("fused", [True, False, "deferred"])
@pytest.mark.parametrize("name,iexp", sorted(RUN_CASES.items()))
def test_substeps_match_reference(name, iexp, fused):
    """Chained substeps from the cold start: through the fused udc_substep, through the reference's
    routine-by-routine call order, and through that same call order with deferred execution on (udc_set_deferred:
    the routines record, tstep_integrate launches the fused substep -- how an untouched driver gets it)."""
    fix = load_fixture(name)
    d, core = core_from_deck(name, iexp)
    g, nsv = core.g, core.nsv
    st = cold_start(g, d, nsv=nsv, pre_boundary=True)      # the fields as readinitfiles leaves them ...
    core.load_state(st)
    core.start_up()                                       # ... then thermodynamics and boundary, in the reference's order
    dt = float(d.get("RUN", "dtmax"))
    dumps = sorted(int(k[1:4]) for k in fix if k.endswith(".u0") and k != "s000.u0")
    if fused == "deferred":
        core._ensure_thermo()
        core.set_deferred(True)
    for isub in range(1, max(dumps) + 1):
        if fused is True:
            rk = (isub - 1) % 3 + 1
            core.substep(rk, dt, with_forces=True)
        else:
            core.tstep_update(dt)
            core.advection(); core.subgrid(); core.bottom(); core.coriolis(); core.forces(); core.ibmwallfun(); core.masscorr()
            core.ibmnorm(); core.scalsource(); core.poisson()
            core.tstep_integrate(); core.halos(); core.boundary()
            if core.moist_thermo:
                core.thermodynamics()
        if isub in dumps:
            tag = f"s{isub:03d}"
            for k in ("u0", "v0", "w0", "pres0") + (("thl0",) if core.ltempeq else ()) + (("qt0",) if core.lmoist else ()):
                ref = marr(fix, f"{tag}.{k}", g.nz)
                sc = 1.0 if k == "thl0" else None        # temperature differences are O(1) K on a 288 K mean
                assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]), sc) <= RUN_TOL, (tag, k)
            if int(d.get("BC", "BCtopm")) == 3:      # the open lid's prognostic plane w0(ke+1), ghost rows included
                ref = marr(fix, f"{tag}.w0", g.nz)
                assert relerr(nocorner(core.download("w0"))[g.nz + 1], nocorner(ref)[g.nz + 1], np.abs(ref).max()) <= RUN_TOL, (tag, "w0(ke+1)")
                if name == "run_ptop_ibm_16x12x10":   # (a block reaches the lid: the masked slab mean of the pressure is not zero there)
                    assert np.abs(ref[g.nz + 1]).max() > 1e-5
            for n in range(nsv):
                got = core.download(L.scalar_field(L.SV0, n), halo=2)
                ref = carr(fix, f"{tag}.sv0_{n + 1:02d}", g.nz)
                assert relerr(interior(got, 2), interior(ref, 2)) <= RUN_TOL
    if fused == "deferred":
        assert core.deferred_stats() == (max(dumps), 0)      # every substep took the fused path
    core.close()


def test_deferred_out_of_order_runs_routine_by_routine():
    """Deferred execution with a call order that is not the reference's (subgrid before advection, poisson missing
    from one substep): the record runs routine by routine and gives what immediate execution gives."""
    name, iexp = "run_16x16x8", RUN_CASES["run_16x16x8"]
    out = []
    for deferred in (False, True):
        d, core = core_from_deck(name, iexp)
        core.load_state(cold_start(core.g, d, nsv=core.nsv))
        dt = float(d.get("RUN", "dtmax"))
        core.set_deferred(deferred)
        for isub in range(3):
            core.tstep_update(dt)
            core.subgrid(); core.advection(); core.forces()
            if isub != 1:
                core.poisson()
            core.tstep_integrate(); core.halos(); core.boundary()
        if deferred:
            assert core.deferred_stats() == (0, 3)
        out.append([core.download(k) for k in ("u0", "v0", "w0", "pres0")])
        core.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def random_state(g, seed, nsv=0, amp=0.05):
    rng = np.random.default_rng(seed)
    st = {}
    for k, base in (("u0", 1.0), ("v0", 0.2), ("w0", 0.0)):
        a = np.zeros(g.mshape())
        a[1:-1, 1:-1, 1:-1] = base + amp * rng.standard_normal((g.nz, g.ny, g.nx))
        st[k] = a
    st["w0"][1] = 0.
    for k in ("u0", "v0", "w0"):
        a = st[k]
        a[:, :, 0] = a[:, :, -2]; a[:, :, -1] = a[:, :, 1]
        a[:, 0, :] = a[:, -2, :]; a[:, -1, :] = a[:, 1, :]
    st["u0"][-1] = st["u0"][-2]; st["v0"][-1] = st["v0"][-2]; st["w0"][-1] = 0.
    for k, m in (("u0", "um"), ("v0", "vm"), ("w0", "wm")):
        st[m] = st[k].copy()
    st["pres0"] = np.zeros(g.mshape())
    for n in range(nsv):
        c = np.zeros(g.cshape())
        zz = (np.arange(g.nz) + 0.5) / g.nz
        c[2:-2, :, :] = zz[:, None, None] * (n + 1) + 0.1 * rng.standard_normal((g.nz, 1, 1))
        c[2:-2, 2:-2, 2:-2] += 0.05 * rng.standard_normal((g.nz, g.ny, g.nx))
        for q in (c,):
            q[:, :, 0:2] = q[:, :, -4:-2]; q[:, :, -2:] = q[:, :, 2:4]
            q[:, 0:2, :] = q[:, -4:-2, :]; q[:, -2:, :] = q[:, 2:4, :]
        c[0] = c[2]; c[1] = c[2]; c[-2] = c[-3]; c[-1] = c[-3]
        st[f"sv0_{n}"] = c
        st[f"svm_{n}"] = c.copy()
    return st


def oracle_state(st, g, nsv):
    o = {k: v.copy() for k, v in st.items() if not k.startswith("sv")}
    for k in ("up", "vp", "wp", "ekm", "ekh", "p", "pup", "pvp", "pwp"):
        o[k] = np.zeros(g.mshape())
    if nsv:
        o["sv0"] = np.stack([st[f"sv0_{n}"] for n in range(nsv)])
        o["svm"] = o["sv0"].copy()
        o["svp"] = np.zeros_like(o["sv0"])
    return o


@pytest.mark.parametrize("shape,sgs,nsv,stretch,floor", [
    ((64, 48, 40), 2, 0, 1.03, False),      # Vreman, stretched z
    ((48, 64, 24), 1, 1, 1.00, False),      # Smagorinsky + kappa scalar
    ((20, 12, 10), 2, 0, 1.00, False),      # non power-of-two FFT lengths (radix 5, 3)
    ((4, 4, 3), 0, 0, 1.00, False),         # smallest grid the library accepts, DNS
    ((128, 8, 6), 1, 2, 1.10, False),       # ragged aspect, two scalars
    ((40, 24, 16), 2, 1, 1.05, True),       # floor wall function (lbottom, BCbotm = 3) + scalar floor
    ((12, 8, 6), 0, 0, 1.00, True),         # floor under DNS viscosity
    ((16, 16, 300), 2, 0, 1.01, False),     # tall columns: 512-thread workgroups (eight levels per thread) / sixteen levels per thread unpaired
    ((8, 8, 530), 1, 0, 1.005, True),       # ... above 512 levels: two systems per thread do not fit, sixteen levels per thread everywhere
])
@pytest.mark.parametrize("thomas", ["stream", "reg", "reg-nopair", "slab-mirror"])
def test_against_oracle_seeded(shape, sgs, nsv, stretch, floor, thomas, monkeypatch):
    """Three substeps (one RK3 step) vs the CPU oracle on seeded random fields, with every variant of the tridiagonal solve:
    the streaming kernel (UDC_THOMAS=0: one thread per mode, solmpj's own order), register-resident segments (the default:
    partitioned recurrences, rows ky and ny - ky of the one-GPU layout solved together), the same without the pairing
    (UDC_THOMAS_PAIR=0) and the slab ranks' layout with the mirrored runs of a line solved together (from lines of 256 by
    default; here from 16: the shapes with 48 and 64 rows take it, the others the unpaired kernel of the slab path -- and every
    shape the radix-8 x backward transform, lines of 2 to 64 complex)."""
    if thomas == "stream":
        monkeypatch.setenv("UDC_THOMAS", "0")
    if thomas == "reg-nopair":
        monkeypatch.setenv("UDC_THOMAS_PAIR", "0")
    if thomas == "slab-mirror":
        monkeypatch.setenv("UDC_FORCE_SLAB", "1")
        monkeypatch.setenv("UDC_THOMAS_MIRROR_MIN", "16")
    nx, ny, nz = shape
    dz = 0.5 * stretch ** np.arange(nz)
    zf = np.cumsum(dz) - 0.5 * dz
    g = Grid.from_levels(nx, ny, nz, nx * 0.5, ny * 0.4, zf)
    from udcore.core import DynCore
    core = DynCore(g, sgs=sgs, nsv=nsv, lbottom=floor, z0=0.03)
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=sgs, nsv=nsv, csz=0.21658244510412, lbottom=floor, z0=0.03)
    st = random_state(g, seed=nx * 1000 + ny, nsv=nsv)
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
    dq = np.zeros(nz + 2); dq[1:nz + 1] = 2e-4
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], dq[1:nz + 1])
    ost = oracle_state(st, g, nsv)
    ost["dpdxl"], ost["dpdyl"] = dp, dq
    dt = 0.05
    for rk in (1, 2, 3):
        core.substep(rk, dt, with_forces=True)
        o.substep(ost, rk, dt)
    for k in ("u0", "v0", "w0", "pres0", "um"):
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1])) <= RUN_TOL, k
    for n in range(nsv):
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        assert relerr(interior(got, 2), interior(ost["sv0"][n], 2)) <= RUN_TOL
    divmax, _ = core.divergence()
    assert divmax < 1e-11
    core.close()


@pytest.mark.parametrize("shape,sgs,nsv,cor", [
    ((36, 20, 14), 2, 1, 1),     # radix-3/5 FFT lengths, Vreman, lcoriol
    ((24, 16, 18), 1, 2, 2),     # Smagorinsky, two scalars, lprofforc
    ((16, 12, 8), 0, 0, 1),      # DNS viscosity
])
def test_all_forcings_together_against_oracle(shape, sgs, nsv, cor):
    """Six substeps with every optional term switched on at once -- floor wall function, Coriolis / geostrophic
    relaxation, buoyant temperature with a radiative source, top value / floor flux, prescribed volume flow in u and
    v, kappa scalars -- through the fused substep, against the CPU oracle on a stretched grid."""
    nx, ny, nz = shape
    dz = 0.4 * 1.06 ** np.arange(nz)
    zf = np.cumsum(dz) - 0.5 * dz
    g = Grid.from_levels(nx, ny, nz, nx * 0.45, ny * 0.5, zf)
    from udcore.core import DynCore
    om22, om23 = 2. * 7.292e-5 * np.cos(0.9), 2. * 7.292e-5 * np.sin(0.9)
    kw = dict(lbottom=True, z0=0.02)
    core = DynCore(g, sgs=sgs, nsv=nsv, **kw)
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=sgs, nsv=nsv, csz=0.21658244510412,
                  luvolflowr=True, uflowrate=1.02, lvvolflowr=True, vflowrate=-0.01,
                  ltempeq=True, bctopt=2, wttop=0., thl_top=290.2, wtsurf=0.04, lbuoyancy=True,
                  coriolis_mode=cor, om22=om22 * 50, om23=om23 * 50, **kw)
    ug = np.zeros(nz + 2); ug[1:nz + 1] = 1.0 + 0.01 * np.arange(nz)
    tc = np.zeros(nz + 2); tc[1:nz + 1] = 1e-3 * np.sin(np.arange(nz))
    core.set_masscorr(True, 1.02, True, -0.01)
    core.set_tempeq(bctopt=2, thl_top=290.2, wtsurf=0.04, thlpcar=tc[1:nz + 1])
    core.set_buoyancy(True)
    core.set_coriolis(cor, om22 * 50, om23 * 50, ug[1:nz + 1])      # x50: make the terms visible in six substeps
    st = random_state(g, seed=7 * nx + ny, nsv=nsv)
    rng = np.random.default_rng(11)
    t = np.zeros(g.mshape())
    t[1:-1, 1:-1, 1:-1] = 288. + 0.3 * g.zf[1:nz + 1, None, None] + 0.05 * rng.standard_normal((nz, ny, nx))
    t[:, 0, :] = t[:, ny, :]; t[:, ny + 1, :] = t[:, 1, :]
    t[:, :, 0] = t[:, :, nx]; t[:, :, nx + 1] = t[:, :, 1]
    t[0] = t[1]; t[nz + 1] = 2 * 290.2 - t[nz]
    st["thl0"], st["thlm"] = t, t.copy()
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -2e-3
    dq = np.zeros(nz + 2); dq[1:nz + 1] = 1e-4
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], dq[1:nz + 1])
    ost = oracle_state(st, g, nsv)
    ost.update(dpdxl=dp, dpdyl=dq, thl0=t.copy(), thlm=t.copy(), thlp=np.zeros(g.mshape()), thlpcar=tc, ug=ug)
    dt = 0.04
    for isub in range(6):
        rk = isub % 3 + 1
        core.substep(rk, dt, with_forces=True)
        o.substep(ost, rk, dt)
    for k in ("u0", "v0", "w0", "pres0", "um", "thl0"):
        sc = 1.0 if k == "thl0" else None
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1]), sc) <= RUN_TOL, k
    for n in range(nsv):
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        assert relerr(interior(got, 2), interior(ost["sv0"][n], 2)) <= RUN_TOL
    # the prescribed volume flow is met: <u> = uflowrate, <v> = vflowrate (dzf-weighted)
    w = g.dzf[1:nz + 1, None, None] / g.dzf[1:nz + 1].sum()
    assert abs((core.download("u0")[1:-1, 1:-1, 1:-1] * w).sum() / (nx * ny) - 1.02) < 1e-12
    assert abs((core.download("v0")[1:-1, 1:-1, 1:-1] * w).sum() / (nx * ny) + 0.01) < 1e-12
    divmax, _ = core.divergence()
    assert divmax < 1e-11
    core.close()


@pytest.mark.parametrize("thls,bcbotm,bcbott", [(288.6, 2, 2), (287.2, 2, 2), (288.1, 3, 2), (288.1, 2, 1)],
                         ids=["unstable", "stable", "neutral-mom+uno-T", "uno-mom+flux-T"])
def test_uno_floor_against_oracle(thls, bcbotm, bcbott):
    """Floor with the stability-dependent wall function (wfuno): a noisy near-floor temperature puts cells on both sides
    of Ri = 0, so both branches of the transfer functions run; six fused substeps against the CPU oracle."""
    nx, ny, nz = 40, 24, 12
    dz = 0.4 * 1.06 ** np.arange(nz)
    zf = np.cumsum(dz) - 0.5 * dz
    g = Grid.from_levels(nx, ny, nz, nx * 0.45, ny * 0.5, zf)
    from udcore.core import DynCore
    kw = dict(lbottom=True, z0=0.03)
    core = DynCore(g, sgs=2, **kw)
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=2, ltempeq=True, bctopt=2, wttop=0., thl_top=290., wtsurf=0.02,
                  lbuoyancy=True, thls=thls, bcbotm=bcbotm, bcbott=bcbott, z0h=0.004, **kw)
    core.set_tempeq(bctopt=2, thl_top=290., bcbott=bcbott, wtsurf=0.02)
    core.set_floor_wf(bcbotm, bcbott, thls, 0.004)
    core.set_buoyancy(True)
    st = random_state(g, seed=23)
    rng = np.random.default_rng(4)
    t = np.zeros(g.mshape())
    t[1:-1, 1:-1, 1:-1] = 288. + 0.2 * g.zf[1:nz + 1, None, None] + 0.3 * rng.standard_normal((nz, ny, nx))
    t[:, 0, :] = t[:, ny, :]; t[:, ny + 1, :] = t[:, 1, :]
    t[:, :, 0] = t[:, :, nx]; t[:, :, nx + 1] = t[:, :, 1]
    t[0] = t[1]; t[nz + 1] = 2 * 290. - t[nz]
    st.update(thl0=t, thlm=t.copy())
    if bcbott == 2 and bcbotm == 2:
        assert ((t[1, 1:-1, 1:-1] - thls) > 0).any() and ((t[1, 1:-1, 1:-1] - thls) < 0).any()
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], np.zeros(nz))
    ost = oracle_state(st, g, 0)
    ost.update(dpdxl=dp, dpdyl=np.zeros(nz + 2), thl0=t.copy(), thlm=t.copy(), thlp=np.zeros(g.mshape()))
    for isub in range(6):
        core.substep(isub % 3 + 1, 0.04, with_forces=True)
        o.substep(ost, isub % 3 + 1, 0.04)
    for k in ("u0", "v0", "w0", "pres0", "thl0"):
        sc = 1.0 if k == "thl0" else None
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1]), sc) <= RUN_TOL, k
    core.close()


def test_moist_buoyancy_against_oracle():
    """Moist thermodynamics in the loop: a partly saturated layer (condensate from thermo, pressures from diagfld /
    fromztop, moist thv0h in the buoyancy term) through six fused substeps against the CPU oracle."""
    nx, ny, nz = 48, 40, 20
    dz = 0.4 * 1.05 ** np.arange(nz)
    zf = np.cumsum(dz) - 0.5 * dz
    g = Grid.from_levels(nx, ny, nz, nx * 0.45, ny * 0.5, zf)
    from udcore.core import DynCore
    kw = dict(lbottom=True, z0=0.03)
    qkw = dict(bctopq=2, wqtop=0., qt_top=0.0105, wqsurf=5e-5)
    core = DynCore(g, sgs=2, **kw)
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=2, ltempeq=True, bctopt=2, wttop=0., thl_top=291., wtsurf=0.04,
                  lbuoyancy=True, lmoist=True, thls=288., qts=0.0105, zf=g.zf, zh=g.zh, **qkw, **kw)
    core.set_tempeq(bctopt=2, thl_top=291., wtsurf=0.04)
    core.set_moisture(**qkw)
    core.set_moist_thermo(288., 0.0105)
    core.set_buoyancy(True)
    st = random_state(g, seed=17)
    rng = np.random.default_rng(8)

    def field(mean, grad, amp, top):
        a = np.zeros(g.mshape())
        a[1:-1, 1:-1, 1:-1] = mean + grad * g.zf[1:nz + 1, None, None] + amp * rng.standard_normal((nz, ny, nx))
        a[:, 0, :] = a[:, ny, :]; a[:, ny + 1, :] = a[:, 1, :]
        a[:, :, 0] = a[:, :, nx]; a[:, :, nx + 1] = a[:, :, 1]
        a[nz + 1] = top(a[nz])
        return a
    t = field(288., 0.25, 0.05, lambda r: 2 * 291. - r)
    t[0] = t[1]
    q = field(0.0118, -8e-5, 2e-4, lambda r: 2 * 0.0105 - r)
    st.update(thl0=t, thlm=t.copy(), qt0=q, qtm=q.copy())
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], np.zeros(nz))
    ost = oracle_state(st, g, 0)
    ost.update(dpdxl=dp, dpdyl=np.zeros(nz + 2), thl0=t.copy(), thlm=t.copy(), thlp=np.zeros(g.mshape()),
               qt0=q.copy(), qtm=q.copy(), qtp=np.zeros(g.mshape()), thermo=o.thermo_tables(), ql0=np.zeros(g.mshape()))
    o.thermodynamics(ost)                              # src/program.f90:120; the device does the same on its first substep
    dt = 0.04
    for isub in range(6):
        core.substep(isub % 3 + 1, dt, with_forces=True)
        o.substep(ost, isub % 3 + 1, dt)
    ql = ost["ql0"][0:nz + 1, 1:-1, 1:-1]
    assert 0.05 < (ql > 0).mean() < 0.9                 # partly cloudy: both branches of thermo are exercised
    got, ref = core.thermo_state(), ost["thermo"]
    for q_, name in enumerate(core.TH_TABLES[:5]):
        r = ref[q_ * (nz + 2):(q_ + 1) * (nz + 2)]
        lo, hi = (2, nz) if name == "thvh" else (1, nz + 1)
        assert np.abs(got[name][lo:hi + 1] - r[lo:hi + 1]).max() <= 1e-12 * np.abs(r).max(), name
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0"):
        sc = 1.0 if k == "thl0" else None
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1]), sc) <= RUN_TOL, k
    core.close()


@pytest.mark.parametrize("bctopq", [1, 2], ids=["fluxtop", "valuetop"])
def test_moisture_against_oracle(bctopq):
    """Total water (lmoist) through six fused substeps next to a passive temperature and two kappa scalars, floor
    flux wqsurf and both top conditions, against the CPU oracle."""
    nx, ny, nz = 48, 40, 20
    dz = 0.4 * 1.05 ** np.arange(nz)
    zf = np.cumsum(dz) - 0.5 * dz
    g = Grid.from_levels(nx, ny, nz, nx * 0.45, ny * 0.5, zf)
    from udcore.core import DynCore
    kw = dict(lbottom=True, z0=0.03)
    qkw = dict(bctopq=bctopq, wqtop=-2e-5, qt_top=0.003, wqsurf=4e-5)
    core = DynCore(g, sgs=2, nsv=2, **kw)
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=2, nsv=2, ltempeq=True, bctopt=1, wttop=0., wtsurf=0.03,
                  lmoist=True, **qkw, **kw)
    core.set_tempeq(bctopt=1, wttop=0., wtsurf=0.03)
    core.set_moisture(**qkw)
    st = random_state(g, seed=91, nsv=2)
    rng = np.random.default_rng(5)

    def field(mean, grad, amp, top):
        a = np.zeros(g.mshape())
        a[1:-1, 1:-1, 1:-1] = mean + grad * g.zf[1:nz + 1, None, None] + amp * rng.standard_normal((nz, ny, nx))
        a[:, 0, :] = a[:, ny, :]; a[:, ny + 1, :] = a[:, 1, :]
        a[:, :, 0] = a[:, :, nx]; a[:, :, nx + 1] = a[:, :, 1]
        a[nz + 1] = top(a[nz])
        return a
    t = field(288., 0.3, 0.05, lambda r: r)
    t[0] = t[1]
    q = field(0.008, -2e-4, 2e-4, (lambda r: 2 * 0.003 - r) if bctopq == 2 else (lambda r: r))   # floor ghost stays zero
    st.update(thl0=t, thlm=t.copy(), qt0=q, qtm=q.copy())
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], np.zeros(nz))
    ost = oracle_state(st, g, 2)
    ost.update(dpdxl=dp, dpdyl=np.zeros(nz + 2), thl0=t.copy(), thlm=t.copy(), thlp=np.zeros(g.mshape()),
               qt0=q.copy(), qtm=q.copy(), qtp=np.zeros(g.mshape()))
    dt = 0.04
    for isub in range(6):
        core.substep(isub % 3 + 1, dt, with_forces=True)
        o.substep(ost, isub % 3 + 1, dt)
    assert np.abs(ost["qt0"][1:-1] - q[1:-1]).max() > 1e-5          # moisture moved
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0", "qtm"):
        sc = 1.0 if k == "thl0" else None
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1]), sc) <= RUN_TOL, k
    for n in range(2):
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        assert relerr(interior(got, 2), interior(ost["sv0"][n], 2)) <= RUN_TOL
    core.close()


@pytest.mark.parametrize("strat", [False, True, "moist"], ids=["neutral", "stratified", "moist"])
def test_tke_closure_runs_against_oracle(strat):
    """One-equation closure (loneeqn) over six fused substeps against the CPU oracle (both with correct periodic
    e120 ghosts; the reference's own multi-substep loneeqn runs are not a target, see tests/golden/make_golden.py)."""
    nx, ny, nz = 24, 16, 12
    dz = 0.5 * 1.05 ** np.arange(nz)
    zf = np.cumsum(dz) - 0.5 * dz
    g = Grid.from_levels(nx, ny, nz, nx * 0.5, ny * 0.5, zf)
    from udcore.core import DynCore, tke_constants
    tk = tke_constants()
    tk.update(thvs=288., ldelta=0)
    core = DynCore(g, sgs=L.SGS_ONEEQN, lbottom=True, z0=0.03)
    kw = dict(ltempeq=True, bctopt=2, thl_top=291., wtsurf=0.02, lbuoyancy=True) if strat else {}
    moist = strat == "moist"      # calthv's moist dthvdz: saturated branch inside a cloud layer near the floor
    qkw = dict(bctopq=2, wqtop=0., qt_top=0.0105, wqsurf=4e-5)
    if moist:
        kw.update(lmoist=True, thls=288., qts=0.0105, zf=g.zf, zh=g.zh, **qkw)
    o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=3, tke=tk, lbottom=True, z0=0.03, **kw)
    st = random_state(g, seed=99)
    rng = np.random.default_rng(3)
    e = np.zeros(g.mshape())
    e[1:-1, 1:-1, 1:-1] = 0.06 + 0.02 * rng.random((nz, ny, nx))
    e[:, 0, :] = e[:, ny, :]; e[:, ny + 1, :] = e[:, 1, :]
    e[:, :, 0] = e[:, :, nx]; e[:, :, nx + 1] = e[:, :, 1]
    e[0] = e[1]; e[nz + 1] = 5e-5
    st["e120"], st["e12m"] = e, e.copy()
    if strat:
        t = np.zeros(g.mshape())
        t[1:-1, 1:-1, 1:-1] = 288. + 0.3 * g.zf[1:nz + 1, None, None] + 0.05 * rng.standard_normal((nz, ny, nx))
        t[:, 0, :] = t[:, ny, :]; t[:, ny + 1, :] = t[:, 1, :]
        t[:, :, 0] = t[:, :, nx]; t[:, :, nx + 1] = t[:, :, 1]
        t[0] = t[1]; t[nz + 1] = 2 * 291. - t[nz]
        st["thl0"], st["thlm"] = t, t.copy()
        core.set_tempeq(bctopt=2, thl_top=291., wtsurf=0.02)
        if moist:
            q = np.zeros(g.mshape())
            q[1:-1, 1:-1, 1:-1] = 0.0118 - 8e-5 * g.zf[1:nz + 1, None, None] + 2e-4 * rng.standard_normal((nz, ny, nx))
            q[:, 0, :] = q[:, ny, :]; q[:, ny + 1, :] = q[:, 1, :]
            q[:, :, 0] = q[:, :, nx]; q[:, :, nx + 1] = q[:, :, 1]
            q[nz + 1] = 2 * 0.0105 - q[nz]
            st["qt0"], st["qtm"] = q, q.copy()
            core.set_moisture(**qkw)
            core.set_moist_thermo(288., 0.0105)
        core.set_buoyancy(True)
    core.set_tke(thvs=288.)
    dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
    dq = np.zeros(nz + 2)
    core.load_state(st)
    core.set_forcing(dp[1:nz + 1], dq[1:nz + 1])
    ost = oracle_state(st, g, 0)
    ost.update(dpdxl=dp, dpdyl=dq, e12p=np.zeros(g.mshape()))
    if strat:
        ost["thlp"] = np.zeros(g.mshape())
    if moist:
        ost.update(qt0=st["qt0"].copy(), qtm=st["qt0"].copy(), qtp=np.zeros(g.mshape()), thermo=o.thermo_tables(),
                   ql0=np.zeros(g.mshape()))
        o.thermodynamics(ost)
    for isub in range(6):
        rk = isub % 3 + 1
        core.substep(rk, 0.05, with_forces=True)
        o.substep(ost, rk, 0.05)
    if moist:
        cloud = ost["ql0"][1:nz + 1, 1:-1, 1:-1] > 0
        assert 0.05 < cloud.mean() < 0.9
    for k in ("u0", "v0", "w0", "pres0", "e120", "e12m") + (("thl0",) if strat else ()) + (("qt0",) if moist else ()):
        sc = 1.0 if k == "thl0" else None
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1]), sc) <= RUN_TOL, k
    assert core.download("e120")[1:-1, 1:-1, 1:-1].min() >= 5e-5
    divmax, _ = core.divergence()
    assert divmax < 1e-11
    core.close()


def test_upload_download_roundtrip_and_x_ghosts():
    g = Grid.uniform(16, 8, 6)
    from udcore.core import DynCore
    core = DynCore(g)
    rng = np.random.default_rng(3)
    a = rng.standard_normal(g.mshape())
    core.upload("u0", a)
    b = core.download("u0")
    np.testing.assert_array_equal(b[:, :, 1:-1], a[:, :, 1:-1])          # bit exact, y/z ghosts included
    np.testing.assert_array_equal(b[:, :, 0], a[:, :, -2])               # x ghosts rebuilt as periodic images
    np.testing.assert_array_equal(b[:, :, -1], a[:, :, 1])
    c = rng.standard_normal(g.cshape())
    core2 = DynCore(g, nsv=1)
    core2.upload(L.scalar_field(L.SV0, 0), c)
    d = core2.download(L.scalar_field(L.SV0, 0), halo=2)
    np.testing.assert_array_equal(d[:, :, 2:-2], c[:, :, 2:-2])
    core.close(); core2.close()


def test_error_behaviour():
    """Bad arguments return an error (never exit): odd itot, unknown field, wrong forcing length."""
    from udcore.core import DynCore
    with pytest.raises(L.UdcError, match="even"):
        DynCore(Grid.uniform(9, 8, 6))
    core = DynCore(Grid.uniform(8, 8, 6))
    with pytest.raises(L.UdcError, match="field"):
        core.upload(L.scalar_field(L.SV0, 0), np.zeros((10, 12, 12)))      # nsv = 0: not allocated
    with pytest.raises(L.UdcError, match="levels"):
        core.set_forcing(np.zeros(3), np.zeros(3))
    core.close()


def test_tstep_maxima_matches_numpy():
    g = Grid.uniform(32, 16, 12)
    from udcore.core import DynCore
    core = DynCore(g)
    st = random_state(g, 11)
    core.load_state(st)
    core.substep(3, 0.05, with_forces=False)     # produces ekm / ekh (RK stage 3: the one whose fields tstep_update looks at; on
    # stages 1, 2 of a deck without scalars nothing reads ekh and the closure does not write it)
    um, vm, wm = (interior(core.download(k)) for k in ("um", "vm", "wm"))
    ekm, ekh = interior(core.download("ekm")), interior(core.download("ekh"))
    dt = 0.05
    dzh = g.dzh[1:g.nz + 1][:, None, None]
    cour = ((np.abs(um) / g.dx + np.abs(vm) / g.dy + np.abs(wm) / dzh) * dt).max()
    f = (1. / dzh ** 2 + 1. / g.dx ** 2 + 1. / g.dy ** 2) * dt
    dif = max(1e-5, (ekm * f).max(), (ekh * f).max())
    core.dt = dt
    import ctypes as C
    c, d = C.c_double(), C.c_double()
    assert core.lib.udc_tstep_maxima(core.h, C.c_double(dt), C.byref(c), C.byref(d)) == 0
    assert abs(c.value - cour) <= 1e-12 * cour and abs(d.value - dif) <= 1e-12 * dif
    core.close()


@pytest.mark.parametrize("stage", [3, 2], ids=["after-stage-3-um-is-u0", "after-stage-2-um-distinct"])
def test_checksim_numbers_match_numpy(stage, monkeypatch):
    """udc_checksim (src/modchecksim.f90:102-203: calccourant's SIGNED maximum of um dxhi + vm dyi + wm dzhi, calcdiffnr, chkdiv's divmax
    and divtot) against the same expressions in numpy -- after RK stage 3 of a fused substep (um is u0: the one-sweep kernel) and after
    stage 2 (um distinct: the two kernels) -- and the two halves udc_checksim_begin / _end against the blocking call."""
    import ctypes as C
    g = Grid.from_levels(32, 16, 12, 16., 6.4, np.cumsum(0.5 * 1.04 ** np.arange(12)) - 0.25 * 1.04 ** np.arange(12))
    from udcore.core import DynCore
    if stage == 2:      # (on stages 1, 2 of a deck without scalars nothing of the reference's loop reads ekh and the closure does not write it)
        monkeypatch.setenv("UDC_EK_ALWAYS", "1")
    core = DynCore(g)
    core.load_state(random_state(g, 12))
    for rk in ((1, 2, 3) if stage == 3 else (1, 2, 3, 1, 2)):
        core.substep(rk, 0.05, with_forces=False)
    dtmn = 0.04
    out = (C.c_double * 4)()
    assert core.lib.udc_checksim(core.h, C.c_double(dtmn), out) == 0
    um, vm, wm = (interior(core.download(k)) for k in ("um", "vm", "wm"))
    ekm, ekh = interior(core.download("ekm")), interior(core.download("ekh"))
    dzh = g.dzh[1:g.nz + 1][:, None, None]
    dzf = g.dzf[1:g.nz + 1][:, None, None]
    cour = ((um / g.dx + vm / g.dy + wm / dzh) * dtmn).max()
    f = (1. / dzh ** 2 + 1. / g.dx ** 2 + 1. / g.dy ** 2) * dtmn
    dif = max((ekm * f).max(), (ekh * f).max())
    u, v, w = (core.download(k) for k in ("u0", "v0", "w0"))
    div = ((u[1:-1, 1:-1, 2:] - u[1:-1, 1:-1, 1:-1]) / g.dx + (v[1:-1, 2:, 1:-1] - v[1:-1, 1:-1, 1:-1]) / g.dy
           + (w[2:, 1:-1, 1:-1] - w[1:-1, 1:-1, 1:-1]) / dzf)
    assert abs(out[0] - cour) <= 1e-12 * abs(cour) and abs(out[1] - dif) <= 1e-12 * dif
    # (the divergence of a projected field is rounding noise: compared on the scale of its terms, |u| / dx)
    scale = np.abs(u).max() / g.dx
    assert abs(out[2] - np.abs(div).max()) <= 1e-13 * scale
    assert abs(out[3] - (div * g.dx * g.dy * dzf).sum()) <= 1e-13 * scale * (g.dx * g.dy * dzf).sum() * div.size / g.nz
    out2 = (C.c_double * 4)()
    assert core.lib.udc_checksim_begin(core.h, C.c_double(dtmn)) == 0
    core.substep(3 if stage == 2 else 1, 0.05, with_forces=False)      # the report is picked up after more work has been queued
    assert core.lib.udc_checksim_end(core.h, out2) == 0
    assert list(out2) == list(out)
    assert core.lib.udc_checksim_end(core.h, out2) != 0                    # nothing pending any more
    core.close()


def test_full_size_properties_256():
    """BASELINE config 2 (256^3): projection leaves a divergence-free field, and a periodic shift
    of the input by (sx, sy) cells shifts the output (checks FFT/halo wiring at full size)."""
    from udcore.core import DynCore
    n = 256
    g = Grid.uniform(n, n, n)
    core = DynCore(g, sgs=L.SGS_VREMAN)
    st = random_state(g, 5, amp=0.02)
    core.load_state(st)
    for rk in (1, 2, 3):
        core.substep(rk, 0.1, with_forces=False)
    divmax, divtot = core.divergence()
    assert divmax < 1e-10, divmax
    u_ref = core.download("u0")
    sx, sy = 37, 101
    st2 = {}
    for k, a in st.items():
        inner = np.roll(a[:, 1:-1, 1:-1], (sy, sx), axis=(1, 2))
        b = a.copy()
        b[:, 1:-1, 1:-1] = inner
        b[:, :, 0] = b[:, :, -2]; b[:, :, -1] = b[:, :, 1]
        b[:, 0, :] = b[:, -2, :]; b[:, -1, :] = b[:, 1, :]
        st2[k] = b
    core.load_state(st2)
    for k in ("up", "vp", "wp"):
        core.upload(k, np.zeros(g.mshape()))
    for rk in (1, 2, 3):
        core.substep(rk, 0.1, with_forces=False)
    u_sh = core.download("u0")
    exp = np.roll(u_ref[1:-1, 1:-1, 1:-1], (sy, sx), axis=(1, 2))
    assert relerr(u_sh[1:-1, 1:-1, 1:-1], exp) <= 1e-10
    core.close()


def test_um_alias_and_mixed_api():
    """After RK stage 3 the fused substep leaves um,vm,wm aliased to u0,v0,w0 (no copy) and rotates the
    buffers at the next stage 1.  Observing um, or switching to the routine-by-routine API in the middle,
    must give the same answers as the all-fused run."""
    from udcore.core import DynCore
    g = Grid.uniform(32, 16, 12)
    st = random_state(g, 21)
    dt = 0.05
    ref = DynCore(g)
    ref.load_state(st)
    for isub in range(9):
        ref.substep(isub % 3 + 1, dt, True)
    mix = DynCore(g)
    mix.load_state(st)
    for isub in range(3):
        mix.substep(isub % 3 + 1, dt, True)
    np.testing.assert_array_equal(mix.download("um"), mix.download("u0"))      # materialised alias
    np.testing.assert_array_equal(mix.download("wm"), mix.download("w0"))
    for isub in range(3, 6):                                                   # routine by routine
        mix.tstep_update(dt)
        mix.advection(); mix.subgrid(); mix.forces(); mix.poisson()
        mix.tstep_integrate(); mix.halos(); mix.boundary()
    for isub in range(6, 9):                                                   # fused again
        mix.substep(isub % 3 + 1, dt, True)
    for k in ("u0", "v0", "w0", "pres0", "um", "vm", "wm"):
        a, b = mix.download(k)[1:-1], ref.download(k)[1:-1]
        assert relerr(nocorner(a), nocorner(b)) <= 1e-12, k
    ref.close(); mix.close()


def test_pressure_total_form_and_in_sweep_scalar_update_are_what_runs(monkeypatch):
    """The two byte savings of round 5 are the default of the fused substep and equal the reference's form to round-off: the
    pressure-total form (the sweep leaves grad pres0 out, the solve's output becomes pres0) and a plain passive kappa
    scalar's RK3 update inside its own sweep.  The executed plan says so (no silent fall-back), nine substeps agree with the same
    library run in the reference's form (UDC_PTOTAL=0, UDC_SV_INLINE=0) at 1e-11, and the planner backs off where it must: with the
    outflow-rate mass correction (the tendencies are summed over one plane) and with a source on the scalar."""
    from udcore.core import DynCore
    g = Grid.uniform(64, 32, 16)
    st = random_state(g, 5, nsv=1)
    st["pres0"] = np.zeros_like(st["pres0"])      # (a pres0 that no solve produced would keep an arbitrary constant in the reference's form only)
    dt = 0.05

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        core = DynCore(g, sgs=1, nsv=1, lbottom=True, z0=0.03)
        core.load_state(st)
        for isub in range(9):
            core.substep(isub % 3 + 1, dt, True)
        plan = core.last_plan()
        out = {k: core.download(k)[1:-1].copy() for k in ("u0", "v0", "w0", "pres0")}
        out["sv0"] = core.download(L.scalar_field(L.SV0, 0), halo=2)[2:-2, 2:-2, 2:-2].copy()
        core.close()
        for k in env:
            monkeypatch.delenv(k)
        return out, plan
    new, plan = run()
    assert plan["pressure_total_form"] and plan["scalars_updated_in_their_sweep"] == 1
    old, plan0 = run(UDC_PTOTAL="0", UDC_SV_INLINE="0")
    assert not plan0["pressure_total_form"] and plan0["scalars_updated_in_their_sweep"] == 0
    for k in new:
        a, b = (new[k], old[k]) if k == "sv0" else (nocorner(new[k]), nocorner(old[k]))
        assert relerr(a, b) <= 1e-11, k
    # where the forms differ the planner keeps the reference's
    core = DynCore(g, sgs=1, nsv=1)
    core.load_state(st)
    core.set_masscorr_outflow(True, 1.0)
    core.substep(1, dt, True)
    assert not core.last_plan()["pressure_total_form"]
    core.close()


def test_pressure_total_form_from_a_pres0_with_a_foreign_constant():
    """A warm start whose pres0 did not come from this solver (a smooth field + 3.7): the reference's form carries the constant along,
    the pressure-total form re-pins it with every solve (include/udcore.h, udc_substep) -- the velocities and the GRADIENT of pres0 agree
    to round-off all the same, pres0 itself up to one constant per run; and p, scratch after a pressure-total substep, is not handed out."""
    from udcore.core import DynCore
    g = Grid.uniform(32, 32, 12)
    st = random_state(g, 9)
    k, j, i = np.meshgrid(np.arange(g.nz + 2), np.arange(g.ny + 2), np.arange(g.nx + 2), indexing="ij")
    st["pres0"] = 3.7 + 0.02 * np.sin(2 * np.pi * (i - 0.5) / g.nx) * np.cos(2 * np.pi * (j - 0.5) / g.ny) * np.cos(np.pi * (k - 0.5) / g.nz)
    st["pres0"][0] = 0.
    dt = 0.05
    out = {}
    for form in ("1", "0"):
        os.environ["UDC_PTOTAL"] = form
        try:
            core = DynCore(g, sgs=2, lbottom=True, z0=0.03)
            core.load_state(st)
            for isub in range(6):
                core.substep(isub % 3 + 1, dt, True)
            assert core.last_plan()["pressure_total_form"] == (form == "1")
            if form == "1":
                with pytest.raises(L.UdcError, match="scratch"):
                    core.download("p")
            out[form] = {q: core.download(q)[1:-1, 1:-1, 1:-1].copy() for q in ("u0", "v0", "w0", "pres0")}
            core.close()
        finally:
            del os.environ["UDC_PTOTAL"]
    for q in ("u0", "v0", "w0"):
        assert relerr(out["1"][q], out["0"][q]) <= 1e-11, q
    d = out["1"]["pres0"] - out["0"]["pres0"]
    assert abs(d.mean() + 3.7) < 0.1 and np.abs(d - d.mean()).max() <= 1e-11 * max(np.abs(out["0"]["pres0"] - out["0"]["pres0"].mean()).max(), 1e-3)


@pytest.mark.parametrize("shape,pmode", [((64, 32, 16), (5, 2, 1)), ((32, 48, 12), (16, 24, 0)), ((20, 12, 10), (3, 1, 7))])
def test_poisson_returns_an_analytic_eigenmode(shape, pmode):
    """p* = cos(2 pi m x) cos(2 pi n y) cos(pi l z) is an eigenvector of the discrete operator `poisson` inverts (uniform
    grid, periodic x / y, Neumann floor and top): fed its discrete gradient as tendencies (um = 0), udc_poisson must
    return p* and project the tendencies to zero -- an answer that involves no FFT (same check as on the reference's
    own poisson, tests/test_oracle_vs_reference.py)."""
    from udcore.core import DynCore
    from udcore.grid import Grid
    nx, ny, nz = shape
    g = Grid.uniform(nx, ny, nz)
    core = DynCore(g, sgs=L.SGS_VREMAN)
    i = np.arange(-1, nx + 1)[None, None, :]
    j = np.arange(-1, ny + 1)[None, :, None]
    k = np.arange(0, nz + 2)[:, None, None]
    pe = np.cos(2 * np.pi * pmode[0] * i / nx) * np.cos(2 * np.pi * pmode[1] * j / ny) * np.cos(np.pi * pmode[2] * (k - 0.5) / nz)
    up, vp, wp = (np.zeros(g.mshape()) for _ in range(3))
    up[1:-1, 1:-1, 1:-1] = (pe[1:-1, 1:-1, 1:-1] - pe[1:-1, 1:-1, :-2]) / g.dx
    vp[1:-1, 1:-1, 1:-1] = (pe[1:-1, 1:-1, 1:-1] - pe[1:-1, :-2, 1:-1]) / g.dy
    wp[2:-1, 1:-1, 1:-1] = (pe[2:-1, 1:-1, 1:-1] - pe[1:-2, 1:-1, 1:-1]) / g.dzh[2:nz + 1, None, None]
    for name, a in (("up", up), ("vp", vp), ("wp", wp)):
        core.upload(name, a)
    core.rk3step, core.dt = 1, 0.3
    core.poisson()
    p = core.download("p")
    assert np.abs(interior(p) - interior(pe)).max() <= 5e-13
    for name in ("up", "vp", "wp"):
        assert np.abs(interior(core.download(name))).max() <= 5e-12
    core.close()
