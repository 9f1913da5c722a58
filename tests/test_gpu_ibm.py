"""Immersed boundary, sparse corrections on the device (udc_ibm.hip) against the reference's own routines: per routine
on the k_ibm fixture (tendencies before / after ibmwallfun and ibmnorm, dumped around the reference's diffu/v/w/c_corr and
solid, oracle/ref_driver.f90), plus properties of a longer run.  The 9-substep run fixture run_ibm_16x12x10 goes through
the generic run tests (tests/test_gpu_parity.py, tests/test_gpu_fortran_dropin.py)."""
import numpy as np
import pytest

from common import carr, deck_path, interior, load_fixture, marr, relerr
from udcore import cold_start, read_deck
from udcore import lib as L
from udcore.ibm import read_ibm

pytestmark = pytest.mark.gpu


def _core(name, iexp):
    import udcore
    d = read_deck(deck_path(name, iexp))
    return d, udcore.from_deck(d)


@pytest.mark.parametrize("name,iexp", [("k_ibm_16x12x10", 54), ("k_ibm_thl_16x12x10", 58), ("k_ibm_wf3_16x12x10", 66), ("k_ibm_wf2_16x12x10", 67),
                                       ("k_ibm_wh2_16x12x10", 69), ("k_ibm_wh1_16x12x10", 71), ("k_ibm_wq2_16x12x10", 81)])
def test_ibm_routines_match_reference(name, iexp):
    """The wf decks: the facet wall functions for momentum (wallfunmom, neutral and with the stability functions on the facet
    temperatures; sections with reconstruction points and an oblique facet normal) ahead of the diffusion corrections.
    wh2: also the heat wall function on the facet temperatures (wallfunheat with iwalltemp = 2) ahead of diffc_corr on thl.
    wq2: also its latent part on the vegetated facets (iwallmoist = 2: moist_flux against the facets' humidity) on qtp.
    The second deck adds temperature with buoyancy: diffc_corr and solid (volume-mean value) on thl, advecc2nd_corr_liberal,
    and the slab averages over the fluid cells (thl0av; thvh through the buoyancy term of the run fixtures)."""
    fix = load_fixture(name)
    d, core = _core(name, iexp)
    nz, nsv = core.g.nz, core.nsv
    for k, rec in (("u0", "sub.u0"), ("v0", "in.v0"), ("w0", "in.w0"), ("um", "in.um"), ("vm", "in.vm"), ("wm", "in.wm"),
                   ("ekm", "sub.ekm"), ("ekh", "sub.ekh")):
        core.upload(k, marr(fix, rec, nz))
    for n in range(nsv):
        core.upload(L.scalar_field(L.SV0, n), carr(fix, f"in.sv0_{n + 1:02d}", nz))
        core.upload(L.scalar_field(L.SVM, n), carr(fix, f"in.svm_{n + 1:02d}", nz))
    thl = "ibw.thlp" in fix
    if thl:
        core.upload("thl0", marr(fix, "sub.thl0", nz)); core.upload("thlm", marr(fix, "in.thlm", nz))
        # diagfld's thl0av over the fluid cells (what ibmnorm's solid value and the host forcings are built from)
        av = core.slab_averages(["thl0"])["thl0"]
        assert np.abs(av[1:nz + 1] - fix["ibm.thl0av"].data[:nz]).max() <= 1e-12 * 288.
        core.upload("thlp", marr(fix, "ibw0.thlp", nz))
    qt = "ibw.qtp" in fix
    if qt:
        core.upload("qt0", marr(fix, "sub.qt0", nz)); core.upload("qtm", marr(fix, "in.qtm", nz))
        core.upload("qtp", marr(fix, "ibw0.qtp", nz))
    # --- ibmwallfun: tendencies as the reference had them before the call
    for t in ("up", "vp", "wp"):
        core.upload(t, marr(fix, f"ibw0.{t}", nz))
    for n in range(nsv):
        core.upload(L.scalar_field(L.SVP, n), carr(fix, f"ibw0.svp_{n + 1:02d}", nz))
    core.ibmwallfun()
    wf = int(d.get("WALLS", "iwallmom")) > 1
    for t in ("up", "vp", "wp"):
        ref, before = marr(fix, f"ibw.{t}", nz), marr(fix, f"ibw0.{t}", nz)
        # without wall functions (iwallmom = 1) the reference never reads the velocity grids' fluid-boundary points
        # (src/modibm.f90:166-179): diffu/v/w_corr change nothing, here as there
        assert (np.abs(ref - before).max() > 1e-9) == wf
        assert relerr(interior(core.download(t)), interior(ref)) <= 1e-12, t
    for n in range(nsv):
        got = core.download(L.scalar_field(L.SVP, n), halo=2)
        assert relerr(interior(got, 2), interior(carr(fix, f"ibw.svp_{n + 1:02d}", nz), 2)) <= 1e-12
    if thl:
        ref, before = marr(fix, "ibw.thlp", nz), marr(fix, "ibw0.thlp", nz)
        assert np.abs(ref - before).max() > 1e-9
        assert relerr(interior(core.download("thlp")), interior(ref)) <= 1e-12
        core.upload("thlp", marr(fix, "ibn0.thlp", nz))
    if qt:
        ref, before = marr(fix, "ibw.qtp", nz), marr(fix, "ibw0.qtp", nz)
        assert np.abs(ref - before).max() > 1e-6
        assert relerr(interior(core.download("qtp")), interior(ref)) <= 1e-11
        core.upload("qtp", marr(fix, "ibn0.qtp", nz))
    # --- ibmnorm
    for t in ("up", "vp", "wp"):
        core.upload(t, marr(fix, f"ibn0.{t}", nz))
    for n in range(nsv):
        core.upload(L.scalar_field(L.SVP, n), carr(fix, f"ibn0.svp_{n + 1:02d}", nz))
    core.ibmnorm()
    for t, vm in (("up", "um"), ("vp", "vm"), ("wp", "wm")):
        assert np.array_equal(interior(core.download(t)), interior(marr(fix, f"ibn.{t}", nz))), t
        assert np.array_equal(interior(core.download(vm)), interior(marr(fix, f"ibn.{vm}", nz))), vm
    for n in range(nsv):
        got = core.download(L.scalar_field(L.SVP, n), halo=2)
        assert relerr(interior(got, 2), interior(carr(fix, f"ibn.svp_{n + 1:02d}", nz), 2)) <= 1e-13
        got = core.download(L.scalar_field(L.SVM, n), halo=2)
        assert relerr(interior(got, 2), interior(carr(fix, f"ibn.svm_{n + 1:02d}", nz), 2)) <= 1e-13
    if thl:       # solid with the volume mean of thl0av, then advecc2nd_corr_liberal
        assert relerr(interior(core.download("thlm")), interior(marr(fix, "ibn.thlm", nz))) <= 1e-13
        ref, before = marr(fix, "ibn.thlp", nz), marr(fix, "ibn0.thlp", nz)
        assert np.abs(interior(ref) - interior(before)).max() > 1e-9
        assert relerr(interior(core.download("thlp")), interior(ref)) <= 1e-12
    if qt:        # solid (mean of the fluid neighbours) and advecc2nd_corr on qt
        assert relerr(interior(core.download("qtm")), interior(marr(fix, "ibn.qtm", nz))) <= 1e-13
        assert relerr(interior(core.download("qtp")), interior(marr(fix, "ibn.qtp", nz))) <= 1e-11
    core.close()


def test_ibm_run_properties():
    """60 substeps around two blocks: the flow stays divergence-free in the fluid, bounded, and the velocities at the solid
    points stay at the size of one pressure correction (ibmnorm zeroes um and the tendency, the projection adds
    rk3coef dp/dx back, as in the reference)."""
    name, iexp = "run_ibm_16x12x10", 55
    d, core = _core(name, iexp)
    lists = read_ibm(d)
    core.load_state(cold_start(core.g, d, nsv=core.nsv))
    dt = float(d.get("RUN", "dtmax"))
    core.run(60, dt)
    divmax, _ = core.divergence()
    assert divmax < 1e-12
    u = core.download("u0")
    assert np.isfinite(u).all() and np.abs(u).max() < 3.
    sol = lists["u"][0]
    inside = np.abs(u[sol[:, 2], sol[:, 1], sol[:, 0]])
    fluid = np.abs(interior(u)).mean()
    assert inside.mean() < 0.2 * fluid
    core.close()


def test_ibm_refusals():
    import udcore
    name, iexp = "run_ibm_16x12x10", 55
    d = read_deck(deck_path(name, iexp))
    d.nml["WALLS"]["iwallmom"] = 2                       # stability functions without an air temperature
    with pytest.raises(ValueError, match="iwallmom = 2"):
        udcore.from_deck(d)
    d.nml["WALLS"]["iwallmom"] = 3                       # wall functions without the facets of the pre-processing
    with pytest.raises(ValueError, match="nfcts"):
        udcore.from_deck(d)
    d.nml["WALLS"]["iwallmom"] = 1
    d.nml.setdefault("PHYSICS", {})["ltempeq"] = True
    d.nml["WALLS"]["iwalltemp"] = 2                      # wall temperatures: the heat wall function needs the facets too
    with pytest.raises(ValueError, match="nfcts"):
        udcore.from_deck(d)
    d.nml["WALLS"]["iwalltemp"] = 1
    d.nml.setdefault("BC", {})["bctfz"] = 0.01           # a prescribed wall flux acts on the facet sections: facets needed
    with pytest.raises(ValueError, match="nfcts"):
        udcore.from_deck(d)


def test_masked_slab_averages():
    """diagfld's slab averages with an immersed boundary run over the fluid cells of each level only (avexy_ibm with IIu,
    IIv, IIc: src/modthermodynamics.f90:271-301, src/modmpi.f90:623-664)."""
    import oracle_lib as ol
    name, iexp = "run_ibm_volflow_16x12x10", 56
    d, core = _core(name, iexp)
    g = core.g
    lists = read_ibm(d)
    core.load_state(cold_start(g, d, nsv=core.nsv))
    dt = float(d.get("RUN", "dtmax"))
    core.run(6, dt)
    o = ol.Oracle(g.nx, g.ny, g.nz, g.dx, g.dy, g.dzf, g.dzh)
    masks = o.set_ibm(lists)
    o.set_ibm(None)
    av = core.slab_averages(["u0", "v0", "sv0_0"])
    sv = core.download(L.scalar_field(L.SV0, 0), halo=2)[1:-1, 1:-1, 1:-1]
    for nm, fld, mk in (("u0", core.download("u0"), masks["u"]), ("v0", core.download("v0"), masks["v"]), ("sv0_0", sv, masks["c"])):
        for k in range(1, g.nz + 2):
            m = mk[k, 1:-1, 1:-1]
            ref = (fld[k, 1:-1, 1:-1] * m).sum() / m.sum()
            assert abs(av[nm][k] - ref) <= 1e-13 * max(abs(ref), 1.), (nm, k)
        # the blocks do make a difference on the levels they occupy
        plain = fld[1, 1:-1, 1:-1].mean()
        assert abs(plain - av[nm][1]) > 1e-6
    core.close()
