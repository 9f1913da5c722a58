"""Reader for uDALES input decks: namoptions.NNN (Fortran namelists), prof.inp.NNN, lscale.inp.NNN.

Mirrors the subset of src/modstartup.f90:105-172 (readnamelists) that the dynamical core
consumes, with the reference's defaults (src/modglobal.f90, src/modsubgriddata.f90:39-61),
so that the *same* namoptions file drives the reference CPU path and this library.
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, field

# group -> {name: default}; names and defaults as in the reference
DEFAULTS = {
    "RUN": dict(iexpnr=0, runtime=300., dtmax=20., ladaptive=False, irandom=43, randu=0.01,
                krand=2 ** 31 - 1, courant=-1., diffnr=0.25, libm=True, lles=True, lrandomize=True,
                nprocx=1, nprocy=1, lwarmstart=False, startfile="", trestart=10000.),
    "DOMAIN": dict(itot=96, jtot=96, ktot=96, xlen=-1., ylen=-1., xlat=52., ksp=-1),
    "PHYSICS": dict(lmoist=False, lcoriol=False, lbuoyancy=False, ltempeq=False, lprofforc=False, ps=101325.,
                    dpdx=0., igrw_damp=0, lnudge=False, lnudgevel=True, tnudge=60., nnudge=0, luvolflowr=False, uflowrate=1., lvvolflowr=False, vflowrate=1., luoutflowr=False, lvoutflowr=False,
                    ifixuinf=0, lvinf=False, tscale=0., lconservativeibm=False),
    "CHEMISTRY": dict(lchem=False, k1=0., JNO2=0.),
    "INLET": dict(Uinf=0., Vinf=0., inletav=0.),
    "DRIVER": dict(idriver=0, tdriverstart=0., driverjobnr=0, dtdriver=0.1, driverstore=0, iplane=-1, lchunkread=False, chunkread_size=100),
    "DYNAMICS": dict(lqlnr=False, ipoiss=0, iadv_mom=2, iadv_tke=-1, iadv_thl=-1, iadv_qt=-1),
    "BC": dict(BCxs=1, BCxm=1, BCxT=1, BCxq=1, BCym=1, BCtopm=1, BCbotm=2, BCzp=1, z0=-1., z0h=-1., BCtopT=1, BCbotT=1, BCbots=1, BCtops=1,
               wttop=0., thl_top=-1., wtsurf=-1., thls=-1., qts=-1.,
               BCtopq=1, BCbotq=1, wqtop=0., qt_top=-1., wqsurf=-1., wsvtopdum=0., ds=0.,
               bctfxm=0., bctfxp=0., bctfym=0., bctfyp=0., bctfz=0., bcqfxm=0., bcqfxp=0., bcqfym=0., bcqfyp=0., bcqfz=0.),
    "SCALARS": dict(nsv=0, lreadscal=False, lscasrc=False, nscasrc=0, lscasrcl=False, nscasrcl=0),
    "NAMSUBGRID": dict(lsmagorinsky=False, lvreman=True, loneeqn=False, c_vreman=0.07, cs=-1.,
                       cf=2.5, cn=0.76, Rigc=0.25, Prandtl=0.333, ldelta=False, lbuoycorr=False),
    "OUTPUT": dict(ltdump=False, lxytdump=False, lmintdump=False, tstatsdump=10000., tsample=5., tstatstart=0.,
                   lfielddump=False, tfielddump=10000., fieldvars="", lydump=False, lytdump=False, lxydump=False, ltkedump=False,
                   lkslicedump=False, lislicedump=False, ljslicedump=False),
    "WALLS": dict(nfcts=-1, lbottom=False, iwallmom=2, iwalltemp=1, iwallmoist=1, nsolpts_u=0, nsolpts_v=0, nsolpts_w=0, nsolpts_c=0,
                  nbndpts_u=0, nbndpts_v=0, nbndpts_w=0, nbndpts_c=0, nfctsecs_u=0, nfctsecs_v=0, nfctsecs_w=0, nfctsecs_c=0,
                  lnorec=False, prandtlturb=0.71, fkar=0.41, lwritefac=False),      # src/modglobal.f90:304 (= prandtlmol), 317; src/modibm.f90:50
    "ENERGYBALANCE": dict(wsoil=0., wfc=313.),      # src/modglobal.f90:286, 292 (read for the vegetated facets' humidity, src/initfac.f90:355)
    "ORACLE": dict(nsub=3, nspin=2, lforces=True, scal_a=1.0, scal_b=0.0),      # (scal_a, scal_b: the linear stand-in profile of decks without a scalar.inp)
}
GEODAMPTIME = 7200.      # src/modglobal.f90 (not a namelist variable)

# Every variable the reference's readnamelists accepts, per group (src/modstartup.f90:105-172, src/modsubgrid.f90:89-90).
# A deck that sets a name outside its group's list would stop the reference ("Problem in namoptions"): read_deck does
# the same.  (wqtop, the top moisture flux, is in no namelist of the reference: it stays at its default.)  &ORACLE is the test
# driver's own group (oracle/ref_driver.f90); the reference's reader never looks at it.
KNOWN = {
    "RUN": "iexpnr lwarmstart lstratstart startfile runmode runtime dtmax trestart ladaptive irandom randu randthl randqt krand "
           "courant diffnr author libm lles lper2inout lwalldist lreadmean nprocx nprocy lrandomize",
    "DOMAIN": "itot jtot ktot xlen ylen xlat xlon xday xtime ksp",
    "PHYSICS": "ps igrw_damp lmoist lcoriol lbuoyancy ltempeq lprofforc ifixuinf lvinf tscale dpdx luoutflowr lvoutflowr "
               "luvolflowr lvvolflowr uflowrate vflowrate lnudge lnudgevel tnudge nnudge ltimedepsurf ntimedepsurf "
               "ltimedepnudge ntimedepnudge ltimedeplw ntimedeplw ltimedepsw ntimedepsw lconservativeibm",
    "DYNAMICS": "lqlnr ipoiss iadv_mom iadv_tke iadv_thl iadv_qt iadv_sv",
    "BC": "BCxm BCxT BCxq BCxs BCym BCyT BCyq BCys BCtopm BCtopT BCtopq BCtops BCbotm BCbotT BCbotq BCbots bctfxm bctfxp "
          "bctfym bctfyp bctfz bcqfxm bcqfxp bcqfym bcqfyp bcqfz wttop thl_top qt_top qts wsvsurfdum wsvtopdum wtsurf "
          "wqsurf thls z0 z0h BCzp ds",
    "INLET": "Uinf Vinf di dti inletav linletRA lstoreplane lreadminl lfixinlet lfixutauin lwallfunc",
    "DRIVER": "idriver tdriverstart driverjobnr dtdriver driverstore iplane iangledeg lchunkread chunkread_size",
    "WALLS": "nfcts iwallmom iwalltemp iwallmoist iwallscal nsolpts_u nsolpts_v nsolpts_w nsolpts_c nbndpts_u nbndpts_v "
             "nbndpts_w nbndpts_c nfctsecs_u nfctsecs_v nfctsecs_w nfctsecs_c lbottom lnorec prandtlturb fkar lwritefac dtfac",
    "ENERGYBALANCE": "lEB lwriteEBfiles lperiodicEBcorr sinkbase lconstW dtEB bldT flrT wsoil wgrmax wwilt wfc skyLW GRLAI "
                     "rsmin nfaclyrs lfacTlyrs lvfsparse nnz fraction",
    "SCALARS": "lreadscal lscasrc lscasrcl lscasrcr nsv nscasrc nscasrcl",
    "CHEMISTRY": "lchem k1 JNO2",
    "OUTPUT": "lfielddump tfielddump fieldvars ltdump lydump lytdump lxydump lxytdump lmintdump lkslicedump kslice "
              "lislicedump islice ljslicedump jslice ltkedump tstatsdump tsample tstatstart",
    "TREES": "ltrees ntrees cd dec ud lad Qstar dQdt lsize r_s ltreedump itree_mode",
    "PURIFS": "lpurif npurif Qpu epu",
    "HEATPUMP": "lheatpump lfan_hp nhppoints Q_dot_hp QH_dot_hp",
    "NAMSUBGRID": "ldelta lmason cf cn Rigc Prandtl lsmagorinsky lvreman loneeqn c_vreman cs nmason lbuoycorr",
    "ORACLE": "nsub nspin nwarm dump_at lforces scal_a scal_b",
}
KNOWN = {g: {n.lower() for n in v.split()} for g, v in KNOWN.items()}
# Switches of features that have no device implementation: a deck that turns one on is refused (udcore.run,
# check_supported) instead of silently running different physics.  (group, name, value that means "off")
UNSUPPORTED = [("WALLS", "lwritefac", False),      # facet statistics (fac.NNN.nc): host code of the reference, not on the device path
               ("RUN", "lstratstart", False), ("RUN", "lper2inout", False), ("RUN", "lreadmean", False),
               ("PHYSICS", "ltimedepsurf", False), ("PHYSICS", "ltimedepnudge", False), ("PHYSICS", "ltimedeplw", False),
               ("PHYSICS", "ltimedepsw", False),
               ("PHYSICS", "lvoutflowr", False),      # (the reference's own call overruns its array there, src/modforces.f90:441-442)
               ("DRIVER", "idriver", 0), ("INLET", "linletRA", False), ("INLET", "lstoreplane", False),
               ("INLET", "lreadminl", False), ("INLET", "lfixinlet", False), ("INLET", "lfixutauin", False),
               ("ENERGYBALANCE", "lEB", False), ("ENERGYBALANCE", "lperiodicEBcorr", False),
               ("SCALARS", "lscasrcr", False),
               ("TREES", "ltrees", False), ("PURIFS", "lpurif", False), ("HEATPUMP", "lheatpump", False)]
# &NAMSUBGRID lmason / nmason are read and broadcast by the reference (src/modsubgrid.f90:90-106) and used nowhere: accepted, no effect


def _expand(tokens):
    """Fortran repeat counts: `3*0.` -> three values."""
    out = []
    for t in tokens:
        m = re.fullmatch(r"(\d+)\*(.*)", t)
        if m:
            out += [m.group(2)] * int(m.group(1))
        else:
            out.append(t)
    return out


def _value(tok: str):
    t = tok.strip().rstrip(",")
    tl = t.lower()
    if tl in (".true.", "t", ".t."):
        return True
    if tl in (".false.", "f", ".f."):
        return False
    if (t.startswith("'") and t.endswith("'")) or (t.startswith('"') and t.endswith('"')):
        return t[1:-1]
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(tl.replace("d", "e"))
    except ValueError:
        return t


def _split_values(s: str):
    """Value tokens of a namelist line: separated by commas / blanks, except inside a quoted string
    (fieldvars = 'u0,v0,w0' is one value)."""
    return [m.group(0) for m in re.finditer(r"'[^']*'|\"[^\"]*\"|[^,\s]+", s)]


def parse_namelists(text: str) -> dict:
    """Returns {GROUP: {name: value-or-list}} for every &GROUP ... / block in the text.

    Line oriented (one or more `name = values` per line, values may continue on the
    following lines), which covers every deck under the reference's examples/ and tests/.
    """
    out = {}
    grp = None
    last = None
    for raw in text.splitlines():
        line = raw.split("!")[0].strip()
        if not line:
            continue
        if line.startswith("&"):
            grp = out.setdefault(line[1:].split()[0].upper(), {})
            last = None
            line = " ".join(line.split()[1:])
            if not line:
                continue
        if grp is None:
            continue
        if line == "/" or line.lower() in ("&end", "$end"):
            grp, last = None, None
            continue
        if line.endswith("/"):
            line, close = line[:-1], True
        else:
            close = False
        parts = re.split(r"(\w+)\s*(?:\([^)]*\))?\s*=", line)
        # parts = [pre, name1, vals1, name2, vals2, ...]
        pre = parts[0].strip()
        if pre and last is not None:
            grp[last] = _as_list(grp[last]) + [_value(v) for v in _expand(_split_values(pre))]
        for q in range(1, len(parts), 2):
            name, vals = parts[q], parts[q + 1]
            vv = [_value(v) for v in _expand(_split_values(vals.strip()))]
            grp[name] = vv[0] if len(vv) == 1 else vv
            last = name
        if close:
            grp, last = None, None
    return out


def _as_list(v):
    return v if isinstance(v, list) else [v]


@dataclass
class Deck:
    """A parsed case: namelist values (with reference defaults) + profile files."""
    path: str
    nml: dict
    zf: list = field(default_factory=list)
    thl: list = field(default_factory=list)
    qt: list = field(default_factory=list)
    u: list = field(default_factory=list)
    v: list = field(default_factory=list)
    tke: list = field(default_factory=list)
    pgx: list = field(default_factory=list)
    pgy: list = field(default_factory=list)
    ug: list = field(default_factory=list)
    vg: list = field(default_factory=list)

    def get(self, group, name):
        g = self.nml.get(group, {})
        # namelist variable names are case-insensitive in Fortran
        for k, v in g.items():
            if k.lower() == name.lower():
                return v
        return DEFAULTS[group][name]

    def is_set(self, group, name):
        return any(k.lower() == name.lower() for k in self.nml.get(group, {}))

    def set(self, group, name, value):
        """Overrides a value whatever the case its name was written in (Deck.get returns the first case-insensitive match)."""
        g = self.nml.setdefault(group, {})
        for k in list(g):
            if k.lower() == name.lower():
                del g[k]
        g[name] = value

    def apply_checkinitvalues(self):
        """The values checkinitvalues re-routes before anything is initialised (src/modstartup.f90:811-816): without the
        temperature equation, or with prescribed wall heat fluxes (iwalltemp = 1, the default), the stability-dependent wall
        function (iwallmom = 2, the default) becomes the neutral one -- on the facets AND on the floor (BCbotm = 3), whatever the
        deck says about BCbotm."""
        if (not self.get("PHYSICS", "ltempeq") or int(self.get("WALLS", "iwalltemp")) == 1) and int(self.get("WALLS", "iwallmom")) == 2:
            self.set("WALLS", "iwallmom", 3)
            self.set("BC", "BCbotm", 3)

    def validate(self):
        """A name the reference's namelist of that group does not hold stops the reference; so it does here."""
        for grp, vals in self.nml.items():
            if grp not in KNOWN:
                # a Fortran namelist read looks for the group it names and skips every other one: groups only the
                # reference's pre-processing reads (&INP, &INPS of tools/) or that no routine reads any more pass unseen
                continue
            for k in vals:
                if k.lower() not in KNOWN[grp]:
                    where = [g for g, names in KNOWN.items() if k.lower() in names]
                    hint = f" (it belongs to &{where[0]})" if where else ""
                    raise ValueError(f"{self.path}: &{grp} has no variable '{k}'{hint} -- Problem in namoptions {grp}")


def _read_table(path, ncol, nrows):
    rows = []
    with open(path) as f:
        lines = f.readlines()[2:]
    for ln in lines:
        p = ln.split()
        if len(p) >= ncol:
            rows.append([float(x) for x in p[:ncol]])
        if len(rows) == nrows:
            break
    if len(rows) != nrows:
        raise ValueError(f"{path}: expected {nrows} rows, found {len(rows)}")
    return rows


def read_deck(namoptions_path: str) -> Deck:
    with open(namoptions_path) as f:
        nml = parse_namelists(f.read())
    d = Deck(path=namoptions_path, nml=nml)
    d.validate()
    d.apply_checkinitvalues()
    exp = d.get("RUN", "iexpnr")
    base = os.path.dirname(os.path.abspath(namoptions_path))
    ktot = d.get("DOMAIN", "ktot")
    prof = _read_table(os.path.join(base, f"prof.inp.{exp:03d}"), 6, ktot)
    d.zf = [r[0] for r in prof]
    d.thl = [r[1] for r in prof]
    d.qt = [r[2] for r in prof]
    d.u = [r[3] for r in prof]
    d.v = [r[4] for r in prof]
    d.tke = [r[5] for r in prof]
    ls = _read_table(os.path.join(base, f"lscale.inp.{exp:03d}"), 10, ktot)
    d.ug = [r[1] for r in ls]
    d.vg = [r[2] for r in ls]
    d.pgx = [r[3] for r in ls]
    d.pgy = [r[4] for r in ls]
    d.wfls = [r[5] for r in ls]         # large-scale vertical velocity -> whls, src/modstartup.f90:2125-2129
    d.dqtdxls, d.dqtdyls, d.dqtdtls = [r[6] for r in ls], [r[7] for r in ls], [r[8] for r in ls]
    d.thlpcar = [r[9] for r in ls]      # dthlrad column -> thlpcar(k), src/modstartup.f90:2060-2097
    nsv = int(d.get("SCALARS", "nsv"))
    sc = os.path.join(base, f"scalar.inp.{exp:03d}")
    d.svprof = None
    if nsv > 0 and os.path.exists(sc):  # z sv(1) .. sv(nsv), src/modstartup.f90:1541-1548
        rows = _read_table(sc, 1 + nsv, ktot)
        d.svprof = [[r[1 + n] for r in rows] for n in range(nsv)]
    return d


def smagorinsky_constant(cs, cf=2.5, alpha_kolm=1.5):
    """csz of src/modsubgrid.f90:65-77."""
    if cs != -1.:
        return cs
    pi = 3.141592653589793116
    cm = cf / (2. * pi) * (1.5 * alpha_kolm) ** (-1.5)
    ceps = 2. * pi / cf * (1.5 * alpha_kolm) ** (-1.5)
    return (cm ** 3 / ceps) ** 0.25
