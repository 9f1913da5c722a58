"""udcore: MI355X-native dynamical core for uDALES (host-side Python mirror of the call surface).

The package directory is `u-dales_amd/` (not an importable name): add it to sys.path and
`import udcore` (tests/conftest.py, bench.py and __graft_entry__.py do that).
"""
from .grid import Grid, cold_start, sgs_from_deck  # noqa: F401
from .namoptions import read_deck, Deck  # noqa: F401


def from_deck(deck, device=0, rank=0, nranks=1):
    """Build a DynCore from a parsed deck (same namoptions the reference reads)."""
    from .core import DynCore
    g = Grid.from_deck(deck)
    sgs, csz, c_vreman, prandtli = sgs_from_deck(deck)
    lbottom = bool(deck.get("WALLS", "lbottom"))
    bcbotm, bcbott = int(deck.get("BC", "BCbotm")), int(deck.get("BC", "BCbotT"))
    if lbottom and bcbotm not in (2, 3):
        raise ValueError("lbottom: BCbotm must be 2 (wfuno) or 3 (wfmneutral), src/modibm.f90:2021-2029")
    import numpy as np
    bcxm = int(deck.get("BC", "BCxm"))
    if bcxm not in (1, 2, 3):
        raise ValueError("&BC BCxm: 1 (periodic), 2 (inflow profile, convective outflow) or 3 (inflow from a precursor's planes) are on the device path")
    driver = bcxm == 3      # (the planes come from udcore.driver.DriverInlet, handed over by the runner; everything else is BCxm = 2's)
    bcx3 = (3,) if driver else ()
    if driver:
        bcxm = 2
    open_x = None
    if bcxm == 2:      # inflow from prof.inp's u, v (xmi_profile); uprof(ke+1) = vprof(ke+1) = 0 as allocated, src/modfields.f90:556
        open_x = (np.concatenate(([0.], np.asarray(deck.u, dtype=float)[:g.nz], [0.])),
                  np.concatenate(([0.], np.asarray(deck.v, dtype=float)[:g.nz], [0.])))
    core = DynCore(g, sgs=sgs, bctopm=3 if bcxm == 2 else int(deck.get("BC", "BCtopm")),      # (src/modstartup.f90:845-848: BCxm = 2 opens the lid)
                   nsv=int(deck.get("SCALARS", "nsv")),
                   prandtli=prandtli, c_vreman=c_vreman, csz=csz, device=device, rank=rank, nranks=nranks,
                   lbottom=lbottom, z0=float(deck.get("BC", "z0")),
                   uinf=float(deck.get("INLET", "Uinf")), vinf=float(deck.get("INLET", "Vinf")), open_x=open_x)
    if bcxm == 2:      # the outlet's speed (src/modboundary.f90:141-160): ubulk of a prescribed flow, else the mean of u's slab averages
        ubulk = float(np.sum(np.asarray(deck.u)[:g.nz] * g.dzf[1:g.nz + 1]) / (g.zh[g.nz + 1] - g.zh[1]))      # src/modstartup.f90:1336-1341
        core.set_open_x_outflow(None if deck.get("PHYSICS", "luvolflowr") else g.dzf[1:g.nz + 1] / (g.zh[g.nz + 1] - g.zh[2]), ubulk)
    core.driver_inflow = driver      # (BCxm = 3: the runner hands the precursor's planes over, udcore.driver)
    if int(deck.get("BC", "BCzp")) != 1:
        core.set_poisson_bczp(int(deck.get("BC", "BCzp")))
    core.set_masscorr(bool(deck.get("PHYSICS", "luvolflowr")), float(deck.get("PHYSICS", "uflowrate")),
                      bool(deck.get("PHYSICS", "lvvolflowr")), float(deck.get("PHYSICS", "vflowrate")))
    if deck.get("PHYSICS", "luoutflowr"):      # masscorr's outflow-rate branch for u (src/modforces.f90:352-387)
        core.set_masscorr_outflow(True, float(deck.get("PHYSICS", "uflowrate")))
    if deck.get("PHYSICS", "ltempeq"):
        iadv = int(deck.get("DYNAMICS", "iadv_thl"))
        core.set_tempeq(iadv_thl=int(deck.get("DYNAMICS", "iadv_mom")) if iadv < 0 else iadv,
                        bctopt=int(deck.get("BC", "BCtopT")), wttop=float(deck.get("BC", "wttop")),
                        thl_top=float(deck.get("BC", "thl_top")), bcbott=int(deck.get("BC", "BCbotT")),
                        wtsurf=float(deck.get("BC", "wtsurf")), thlpcar=getattr(deck, "thlpcar", None))
        if bcxm == 2:      # inflow / outflow: BCxT = 2, the temperature enters with prof.inp's profile too (xTi_profile), thlprof(ke+1) = 0 as
            # allocated; BCxT = 1 (the reference's default, its tests/cases/525): the temperature stays periodic (halos' xT_periodic)
            if int(deck.get("BC", "BCxT")) not in (1, 2) + bcx3:
                raise ValueError("&BC BCxm = 2 with the temperature equation: BCxT = 1 (periodic) or 2 (inflow profile, convective outflow) is what the device path has")
            if int(deck.get("BC", "BCxT")) == 2:
                core.set_open_x_thl(np.concatenate(([0.], np.asarray(deck.thl, dtype=float)[:g.nz], [0.])))
    if lbottom and bcbotm == 2 and not deck.get("PHYSICS", "ltempeq"):
        # the reference's thl0 stays at prof.inp's profile when the temperature equation is off; wfuno reads its first level
        core.set_floor_air_temperature(float(deck.thl[0]))
    prt = float(deck.get("WALLS", "prandtlturb"))      # &WALLS prandtlturb, fkar (src/modstartup.f90:152-153; defaults prandtlmol, 0.41)
    if deck.is_set("WALLS", "fkar"):
        core.set_fkar(float(deck.get("WALLS", "fkar")))
    if lbottom and (bcbotm == 2 or (bcbott == 2 and deck.get("PHYSICS", "ltempeq"))):
        core.set_floor_wf(bcbotm, bcbott, float(deck.get("BC", "thls")), float(deck.get("BC", "z0h")), prt)
    if deck.get("PHYSICS", "lmoist"):
        iadv = int(deck.get("DYNAMICS", "iadv_qt"))
        core.set_moisture(iadv_qt=int(deck.get("DYNAMICS", "iadv_mom")) if iadv < 0 else iadv,
                          bctopq=int(deck.get("BC", "BCtopq")), wqtop=float(deck.get("BC", "wqtop")),
                          qt_top=float(deck.get("BC", "qt_top")), bcbotq=int(deck.get("BC", "BCbotq")),
                          wqsurf=float(deck.get("BC", "wqsurf")))
        if bcxm == 2:      # inflow / outflow: BCxq = 1 periodic (xq_periodic), 2 mirrored about prof.inp's profile (xqi_profile), qtprof(ke+1) = 0 as allocated
            if int(deck.get("BC", "BCxq")) not in (1, 2) + bcx3:
                raise ValueError("&BC BCxm = 2 with moisture: BCxq = 1 (periodic) or 2 (inflow profile, convective outflow) is what the device path has")
            if int(deck.get("BC", "BCxq")) == 2:
                core.set_open_x_qt(np.concatenate(([0.], np.asarray(deck.qt, dtype=float)[:g.nz], [0.])))
        if deck.get("PHYSICS", "lbuoyancy") or sgs == 3:      # the moist thermodynamics feed the buoyancy and calthv's dthvdz
            if not deck.get("PHYSICS", "ltempeq"):
                raise ValueError("lmoist with lbuoyancy or loneeqn needs ltempeq on the device path")
            core.set_moist_thermo(float(deck.get("BC", "thls")), float(deck.get("BC", "qts")), float(deck.get("PHYSICS", "ps")),
                                  lqlnr=bool(deck.get("DYNAMICS", "lqlnr")))
    if deck.get("PHYSICS", "ltempeq") and deck.get("PHYSICS", "lbuoyancy"):
        core.set_buoyancy(True)
        if sgs == 2 and deck.get("NAMSUBGRID", "lbuoycorr"):      # src/modsubgrid.f90:332
            core.set_buoycorr(True, float(deck.get("NAMSUBGRID", "Rigc")))
    if sgs == 3:      # after set_tempeq: the closure reads thl0 when the temperature equation is on
        thls, qts = float(deck.get("BC", "thls")), float(deck.get("BC", "qts"))
        core.set_tke(cf=float(deck.get("NAMSUBGRID", "cf")), cn=float(deck.get("NAMSUBGRID", "cn")),
                     Rigc=float(deck.get("NAMSUBGRID", "Rigc")), Prandtl=float(deck.get("NAMSUBGRID", "Prandtl")),
                     thvs=thls * (1. + (461.5 / 287.04 - 1.) * qts), ldelta=bool(deck.get("NAMSUBGRID", "ldelta")))
    import numpy as np
    # dpdxl, dpdyl: src/modstartup.f90:2071-2081 (lcoriol false => om23_gs terms still present:
    # dpdxl = om23_gs*vg - pgx - dpdx with om23_gs = 2*omega*sin(lat); ug = vg = 0 in our decks)
    import math
    phi = float(deck.get("DOMAIN", "xlat")) * 3.141592653589793116 / 180.      # src/modglobal.f90:660-675
    om23_gs = 2. * 7.292e-5 * math.sin(phi)
    # coriolis (src/modforces.f90:600-717): om22, om23 of src/modglobal.f90:666-673
    mode = 1 if deck.get("PHYSICS", "lcoriol") else (2 if deck.get("PHYSICS", "lprofforc") else 0)
    if mode:
        core.set_coriolis(mode, 2. * 7.292e-5 * math.cos(phi), 2. * 7.292e-5 * math.sin(phi), np.array(deck.ug))
    dpdx = float(deck.get("PHYSICS", "dpdx"))
    if deck.get("PHYSICS", "lprofforc"):
        dpdxl = [-pg - dpdx for pg in deck.pgx]
        dpdyl = [-pg for pg in deck.pgy]
    else:
        dpdxl = [om23_gs * vg - pg - dpdx for vg, pg in zip(deck.vg, deck.pgx)]
        dpdyl = [-om23_gs * ug - pg for ug, pg in zip(deck.ug, deck.pgy)]
    core.set_forcing(np.array(dpdxl), np.array(dpdyl))
    if deck.get("CHEMISTRY", "lchem"):
        core.set_chem(True, float(deck.get("CHEMISTRY", "k1")), float(deck.get("CHEMISTRY", "JNO2")))
    if core.nsv:      # top condition of the scalars (src/modboundary.f90:236-247; sv_top = svprof(ke), src/modstartup.f90:1574)
        bctops = int(deck.get("BC", "BCtops"))
        w = deck.get("BC", "wsvtopdum")
        w = list(w) if isinstance(w, (list, tuple)) else [w]
        if bctops == 2:
            from .grid import scalar_profiles
            prof = scalar_profiles(g, deck, core.nsv)
            for n in range(core.nsv):
                core.set_scalar_top(n, 2, prof[n][g.nz])
        else:
            for n in range(core.nsv):
                core.set_scalar_top(n, 1, float(w[n]) if n < len(w) else 0.)
    bcxs = int(deck.get("BC", "BCxs"))
    if bcxs not in (1, 2) + bcx3:
        raise ValueError("&BC BCxs: 1 (periodic), 2 (inflow profile, convective outflow) or, with BCxm = 3, 3 (the precursor's planes) are on the device path")
    if core.nsv and bcxm == 2:      # inflow / outflow for the flow: the scalars enter and leave with it (the rows carry their ghost columns)
        if bcxs not in (2,) + bcx3:
            raise ValueError("&BC BCxm = 2 / 3 with passive scalars: BCxs = 2 (inflow profile, convective outflow) or 3 (driver planes) is what the device path has")
        from .grid import scalar_profiles
        core.set_open_x_scalars(np.array(scalar_profiles(g, deck, core.nsv)))
    elif core.nsv and bcxs == 2:      # scalars enter at the low-x side with svprof and leave at the high-x side (src/modboundary.f90:844, 983)
        if deck.is_set("PHYSICS", "luoutflowr") and deck.nml["PHYSICS"][[k for k in deck.nml["PHYSICS"] if k.lower() == "luoutflowr"][0]]:
            raise ValueError("BCxs = 2 with luoutflowr is not on the device path")
        from .grid import scalar_profiles
        ubulk = float(np.sum(np.asarray(deck.u) * g.dzf[1:g.nz + 1]) / (g.zh[g.nz + 1] - g.zh[1]))      # src/modstartup.f90:1336-1341
        core.set_scalar_bcx(2, np.array(scalar_profiles(g, deck, core.nsv)), ubulk)      # luvolflowr: the outlet convects with ubulk (:159)
        if not deck.get("PHYSICS", "luvolflowr"):      # else with the mean of diagfld's slab averages (:143-156)
            core.set_scalar_bcx_outflow(g.dzf[1:g.nz + 1] / (g.zh[g.nz + 1] - g.zh[2]))
    from .ibm import apply_ibm
    apply_ibm(core, deck)
    if core.nsv and (deck.get("SCALARS", "lscasrc") or deck.get("SCALARS", "lscasrcl")):
        from .sources import apply_sources
        apply_sources(core, deck, j0=rank * core.nyl)
    return core
