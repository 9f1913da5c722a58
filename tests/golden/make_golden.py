#!/usr/bin/env python3
"""Generate the golden fixtures in this directory.

Runs oracle/_ref/udales_ref -- the reference's own UNMODIFIED src/ tree (every file, compiled from
/root/reference/src by oracle/Makefile against single-rank stand-ins for MPI / 2DECOMP / FFTW / NetCDF) under
oracle/ref_driver.f90, a main program with the start-up and loop of src/program.f90 plus dump records -- on
the small decks defined below, and stores the inputs/outputs it dumps.  The fixtures are data
only (fields in, fields out); no reference source text is stored.

    python tests/golden/make_golden.py [case ...]   # needs oracle/_ref/udales_ref (make -C oracle ref)

Fixture format: the record stream of oracle/ref_driver.f90 (see tests/refdump.py), gzip'ed,
plus the input deck files of each case under cases/<name>/ so that the same namoptions drive
the device library.
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refdump import read_dump, write_dump  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "udales_ref")


def deck(iexpnr, nx, ny, nz, dx=0.5, dy=0.5, dtmax=0.25, sgs="vreman", nsv=0, bctopm=1,
         oracle="", lles=True, randu=0.01, floor=False, z0=0.05, physics="", bc="", bcbotm=3, scalars="", dynamics="", inlet="", ladaptive=False, chemistry="",
         ibm=None, output="", iwallmom=1, walls="", extra=""):
    sub = {"oneeqn": "loneeqn = .true.\nlvreman = .false.\nlsmagorinsky = .false.",
           "vreman": "lvreman = .true.\nlsmagorinsky = .false.",
           "vreman_bc": "lvreman = .true.\nlsmagorinsky = .false.\nlbuoycorr = .true.",
           "smag": "lsmagorinsky = .true.\nlvreman = .false.",
           "dns": "lvreman = .false.\nlsmagorinsky = .false."}[sgs]
    return f"""&RUN
iexpnr = {iexpnr}
runtime = 1000.
dtmax = {dtmax}
ladaptive = {'.true.' if ladaptive else '.false.'}
irandom = 43
randu = {randu}
nprocx = 1
nprocy = 1
libm = {'.true.' if ibm else '.false.'}
lles = {'.true.' if lles else '.false.'}
/
&DOMAIN
itot = {nx}
jtot = {ny}
ktot = {nz}
xlen = {nx * dx}
ylen = {ny * dy}
/
&PHYSICS
{physics}
/
&DYNAMICS
ipoiss = 0{(chr(10) + dynamics) if dynamics else ''}
/
&BC
BCtopm = {bctopm}
{('BCbotm = ' + str(bcbotm) + chr(10) + 'z0 = ' + repr(z0)) if floor else ''}
{bc}
/
{('&WALLS' + chr(10) + 'nfcts = 0' + chr(10) + ('lbottom = .true.' + chr(10) if floor else '') + ((ibm_walls(ibm, nx, ny, nz) if iwallmom == 1 else ibm_walls_wf(ibm, nx, ny, nz, dx, dy, 0.5, iwallmom)) if ibm else '') + (walls + chr(10) if walls else '') + '/') if (floor or ibm) else ''}
&SCALARS
nsv = {nsv}{(chr(10) + scalars) if scalars else ''}
/
&NAMSUBGRID
{sub}
/{(chr(10) + '&OUTPUT' + chr(10) + output + chr(10) + '/') if output else ''}{(chr(10) + '&INLET' + chr(10) + inlet + chr(10) + '/') if inlet else ''}{(chr(10) + '&CHEMISTRY' + chr(10) + chemistry + chr(10) + '/') if chemistry else ''}
{(extra + chr(10)) if extra else ''}&ORACLE
{oracle}
/
"""


GROUND_CASES = set()      # blocks == "ground": no obstacles, the floor itself is the immersed boundary (the reference's examples/001)


def ibm_lists(blocks, nx, ny, nz):
    """Point lists of the reference's IBM input files for axis-aligned blocks [(i0, i1, j0, j1, k1), ...] of solid cells
    (1-based, inclusive; blocks stand on the floor).  A u point is solid when the face between cells i-1 and i touches a
    solid cell (likewise v, w); fluid-boundary points are the fluid points with a solid point of their own grid among
    their six neighbours (x, y periodic).  Rows in k, j, i order."""
    import numpy as np
    if blocks == "ground":      # the floor as facets: the w points of the first level are solid, the first cells of u, v, c (and
        # the second level of w) look at it -- the lists of the reference's examples/001
        full = [(i + 1, j + 1, 1) for j in range(ny) for i in range(nx)]
        return {"u": ([], full), "v": ([], full), "c": ([], full), "w": (full, [(i, j, 2) for (i, j, _) in full])}
    c = np.zeros((nz + 2, ny, nx), dtype=bool)
    for (i0, i1, j0, j1, k1) in blocks:
        c[1:k1 + 1, j0 - 1:j1, i0 - 1:i1] = True
    u = c | np.roll(c, 1, axis=2)
    v = c | np.roll(c, 1, axis=1)
    w = c.copy(); w[1:] |= c[:-1]
    out = {}
    for name, sol in (("u", u), ("v", v), ("w", w), ("c", c)):
        nb = np.zeros_like(sol)
        for ax, sh in ((2, 1), (2, -1), (1, 1), (1, -1)):
            nb |= np.roll(sol, sh, axis=ax)
        nb[1:] |= sol[:-1]; nb[:-1] |= sol[1:]
        bnd = nb & ~sol
        lo = 2 if name == "w" else 1
        pts = lambda m: [(i + 1, j + 1, k) for k in range(lo, nz + 1) for j in range(ny) for i in range(nx) if m[k, j, i]]   # noqa: E731
        out[name] = (pts(sol) if name != "w" else [(i + 1, j + 1, k) for k in range(1, nz + 1) for j in range(ny) for i in range(nx) if sol[k, j, i]],
                     pts(bnd))
    return out


def ibm_walls(blocks, nx, ny, nz):
    L = ibm_lists(blocks, nx, ny, nz)
    # (src/modibm.f90:180-186 reads facet_sections_c.txt whenever a c grid exists, wall functions or not: an empty list then)
    return "iwallmom = 1\n" + "".join(f"nsolpts_{g} = {len(L[g][0])}\nnbndpts_{g} = {len(L[g][1])}\n" for g in "uvwc") + "nfctsecs_c = 0\n"


def write_ibm_files(d, blocks, nx, ny, nz):
    with open(os.path.join(d, "facet_sections_c.txt"), "w") as f:      # overwritten by write_facet_files for the wall-function decks
        f.write(" # facet      area flux point distance\n")
    for g, (sol, bnd) in ibm_lists(blocks, nx, ny, nz).items():
        for fn, pts in ((f"solid_{g}.txt", sol), (f"fluid_boundary_{g}.txt", bnd)):
            with open(os.path.join(d, fn), "w") as f:
                f.write("# position (i,j,k)\n")
                for p in pts:
                    f.write("%5d %5d %5d\n" % p)


# ---- facets of the blocks for the reference's wall functions (iwallmom = 2, 3): facets.inp, factypes.inp, Tfacinit.inp and the
#      facet_sections_<grid>.txt lists (facet, area, boundary point, distance) the pre-processing would write.  The geometry is
#      that of the blocks -- one facet per block face; a section wherever a fluid-boundary point has a solid neighbour -- bent
#      on purpose in two places so that every branch of initibmwallfun / wallfunmom runs: the second block's east face carries
#      an oblique normal (rotation of the stress, a reconstruction line that leaves the cell sideways) and the second facet
#      type is so rough that the log law is ill-defined at half a cell (reconstruction points, trilinear interpolation).
FACET_TYPES = [(1, 0.01, 0.001), (2, 0.12, 0.0035)]      # id, z0, z0h


WF_NO_OBLIQUE = set()      # cases whose facets are all grid-aligned (prescribed wall heat fluxes are defined for those only)


def facet_files(blocks, nx, ny, nz, dx, dy, dz, oblique=True, green=False):
    """-> (files {name: text}, counts {grid: nfctsecs}, nfcts).  green: the second facet type is vegetated (lGR)."""
    import numpy as np
    L = ibm_lists(blocks, nx, ny, nz)
    if blocks == "ground":      # one facet per column pair, normal +z, half a cell below the u, v, c points and a cell below w(2)
        nf = 4
        files = {"facets": "# type, normal\n" + "".join("%d 0.0000 0.0000 1.0000\n" % (1 + q % 2) for q in range(nf)),
                 "factypes": "# walltype\n# -\n# wallid lGR z0 z0h al em d1 d2 d3 C1 C2 C3 l1 l2 l3 k1 k2 k3 k4\n" +
                             "".join("%d 0 %.4f %.5f 0.5 0.85 0.1 0.2 0.2 1875000 1875000 1875000 0.75 0.75 0.75 4e-7 4e-7 4e-7 4e-7\n" % t
                                     for t in ((1, 0.05, 0.00035), (2, 0.02, 0.0002))),
                 "Tfacinit": "# initial facet temperatures\n" + "".join("%.2f\n" % (288.5 + 0.5 * q) for q in range(nf))}
        counts = {}
        for g in "uvwc":
            rows = [(1 + (i + j) % nf, dx * dy, q + 1, dz if g == "w" else 0.5 * dz) for q, (i, j, k) in enumerate(L[g][1])]
            counts[g] = len(rows)
            files[f"sections_{g}"] = " # facet      area flux point distance\n" + "".join("%8d %9.4f %10d %8.4f\n" % r for r in rows)
        return files, counts, nf
    owner = np.zeros((nz + 2, ny, nx), dtype=int)          # block number (1-based) of a solid c cell
    for b, (i0, i1, j0, j1, k1) in enumerate(blocks):
        owner[1:k1 + 1, j0 - 1:j1, i0 - 1:i1] = b + 1
    # a solid u / v / w point belongs to the block of one of the cells it touches
    own = {"c": owner,
           "u": np.maximum(owner, np.roll(owner, 1, axis=2)), "v": np.maximum(owner, np.roll(owner, 1, axis=1))}
    ow = owner.copy(); ow[1:] = np.maximum(owner[1:], owner[:-1]); own["w"] = ow
    faces = ["top", "west", "east", "south", "north"]      # facet number = 5 (block - 1) + 1 + index
    normal = {"top": (0., 0., 1.), "west": (-1., 0., 0.), "east": (1., 0., 0.), "south": (0., -1., 0.), "north": (0., 1., 0.)}
    facets = []
    for b in range(len(blocks)):
        for f in faces:
            n = normal[f]
            if b == 1 and f == "east" and oblique:
                n = (0.8, 0.6, 0.)
            facets.append((2 if (b == 1 and f in ("top", "north")) else 1, n))
    nfcts = len(facets)
    files = {"facets": "# type, normal\n" + "".join("%d %.4f %.4f %.4f\n" % ((t,) + n) for t, n in facets),
             "factypes": "# walltype\n# -\n# wallid lGR z0 z0h al em d1 d2 d3 C1 C2 C3 l1 l2 l3 k1 k2 k3 k4\n" +
                         "".join("%d %d %.4f %.5f 0.5 0.85 0.1 0.2 0.2 1875000 1875000 1875000 0.75 0.75 0.75 4e-7 4e-7 4e-7 4e-7\n"
                                 % (t[0], 1 if (green and t[0] == 2) else 0, t[1], t[2]) for t in FACET_TYPES),
             "Tfacinit": "# initial facet temperatures\n" + "".join("%.2f\n" % ((292.0 if green else 289.5) + 0.25 * (q % 4)) for q in range(nfcts))}
    counts = {}
    # direction to the solid neighbour -> the face of that neighbour's block the point looks at
    dirs = [((1, 0, 0), "west", dy * dz, dx), ((-1, 0, 0), "east", dy * dz, dx), ((0, 1, 0), "south", dx * dz, dy),
            ((0, -1, 0), "north", dx * dz, dy), ((0, 0, -1), "top", dx * dy, dz)]
    for g in "uvwc":
        sol = own[g]
        rows = []
        for q, (i, j, k) in enumerate(L[g][1]):
            for (di, dj, dk), face, area, h in dirs:
                ii, jj, kk = (i - 1 + di) % nx, (j - 1 + dj) % ny, k + dk
                if kk < 1 or kk > nz:
                    continue
                b = sol[kk, jj, ii]
                if g == "w" and kk == 1 and b == 0:
                    continue
                if b:
                    # half a cell from the face, a whole one along the grid's own staggering direction
                    stag = {"u": (1, 0, 0), "v": (0, 1, 0), "w": (0, 0, 1)}.get(g, (0, 0, 0))
                    dist = h if (abs(di), abs(dj), abs(dk)) == stag else 0.5 * h
                    rows.append((5 * (b - 1) + 1 + faces.index(face), area, q + 1, dist))
        counts[g] = len(rows)
        files[f"sections_{g}"] = " # facet      area flux point distance\n" + "".join("%8d %9.4f %10d %8.4f\n" % r for r in rows)
    return files, counts, nfcts


def ibm_walls_wf(blocks, nx, ny, nz, dx, dy, dz, iwallmom):
    L = ibm_lists(blocks, nx, ny, nz)
    _, counts, nfcts = facet_files(blocks, nx, ny, nz, dx, dy, dz)      # (the counts do not depend on the oblique normal)
    return (f"iwallmom = {iwallmom}\nnfcts = {nfcts}\n" + "".join(f"nsolpts_{g} = {len(L[g][0])}\nnbndpts_{g} = {len(L[g][1])}\n" for g in "uvwc")
            + "".join(f"nfctsecs_{g} = {counts[g]}\n" for g in "uvwc"))


def write_facet_files(d, iexp, blocks, nx, ny, nz, dx, dy, dz, oblique=True, green=False):
    files, _, _ = facet_files(blocks, nx, ny, nz, dx, dy, dz, oblique, green)
    for name, text in files.items():
        fn = f"facet_{name}.txt" if name.startswith("sections_") else f"{name}.inp.{iexp:03d}"
        with open(os.path.join(d, fn), "w") as f:
            f.write(text)


def zlevels(nz, dz0=0.5, stretch=1.0):
    zf, zh = [], 0.0
    dz = dz0
    for _ in range(nz):
        zf.append(zh + 0.5 * dz)
        zh += dz
        dz *= stretch
    return zf


def write_case(d, iexpnr, text, zf, u=1.0, v=0.0, pgx=1e-4, dthl=0.0, dthlrad=0.0, ug=0.0, tke=0.0, wtop=0.0, qt=0.0, dqt=0.0, dqtdx=0.0, dqtdy=0.0, dqtdt=0.0,
               psrc=None, lsrc=None, nsv=0, scal_a=1.0, scal_b=0.0):
    with open(os.path.join(d, f"namoptions.{iexpnr:03d}"), "w") as f:
        f.write(text)
    if nsv > 0:      # scalar.inp (src/modstartup.f90:1541-1548): linear profiles, scalar n = scal_b + scal_a n z / zsize
        zh = [0.0]
        for z in zf:
            zh.append(2.0 * z - zh[-1])
        zsize = zh[-1]
        with open(os.path.join(d, f"scalar.inp.{iexpnr:03d}"), "w") as f:
            f.write("# golden\n# z sv(1..nsv)\n")
            for z in zf:
                f.write(f"{z:.15f} " + " ".join(repr(scal_b + scal_a * (n + 1) * z / zsize) for n in range(nsv)) + "\n")
    with open(os.path.join(d, f"prof.inp.{iexpnr:03d}"), "w") as f:
        f.write("# golden\n# z thl qt u v tke\n")
        for z in zf:
            f.write(f"{z:.15f} {288.0 + dthl * z!r} {(qt + dqt * z) if (qt or dqt) else 0.0!r} {u} {v} {tke!r}\n")
    with open(os.path.join(d, f"lscale.inp.{iexpnr:03d}"), "w") as f:
        f.write("# golden\n# z uq vq pqx pqy wfls dqtdxls dqtdyls dqtdtls dthlrad\n")
        for z in zf:
            wf = wtop * z / zf[-1] if wtop else 0.0
            f.write(f"{z:.15f} {ug!r} 0.0 {pgx} 0.0 {wf!r} {dqtdx!r} {dqtdy!r} {dqtdt!r} {dthlrad!r}\n")


    # scalar point / line sources: one file per scalar (src/modscalsource.f90:292-375)
    for kind, rows in (("p", psrc), ("l", lsrc)):
        for n, r in enumerate(rows or []):
            with open(os.path.join(d, f"scalarsource{kind}.inp.{n + 1}.{iexpnr:03d}"), "w") as f:
                f.write("# golden\n# " + ("xS yS zS SS sigS" if kind == "p" else "xSb ySb zSb xSe ySe zSe SS sigS") + "\n")
                for row in r:
                    f.write(" ".join(repr(float(x)) for x in row) + "\n")


KEEP_KERNELS = ("meta dzf dzh zf dpdxl dpdyl rk3 in.uouttot in.u0 in.v0 in.w0 in.um in.vm in.wm in.pres0 "
                "in.ekm in.ekh adv.up adv.vp adv.wp sub.ekm sub.ekh sub.u0 sub.up sub.vp sub.wp bot.up bot.vp frc.up frc.vp "
                "in.thl0 in.thlm adv.thlp sub.thlp sub.thl0 bot.thlp pre.thlp out.thl0 out.thlm "
                "thm.presf thm.presh thm.exnf thm.exnh thm.thvh thm.ql0av thm.ql0 thm.thv0h "
                "thn.presf thn.presh thn.exnf thn.exnh thn.thvh thn.ql0av thn.ql0 thn.thv0h "
                "in.qt0 in.qtm adv.qtp sub.qtp sub.qt0 bot.qtp pre.qtp out.qt0 out.qtm "
                "in.e120 in.e12m adv.e12p sub.e12p pre.e12p out.e120 out.e12m "
                "frc0.up frc0.vp frc0.wp frc0.thlp lsf.up lsf.vp lsf.wp lsf.thlp u0av thl0av frc0.qtp lsf.qtp qt0av "
                "src0.up fix0.up fix0.vp "
                "ibw0.up ibw0.vp ibw0.wp ibw.up ibw.vp ibw.wp ibn0.up ibn0.vp ibn0.wp ibn.up ibn.vp ibn.wp ibn.um ibn.vm ibn.wm "
                "ibw0.thlp ibw.thlp ibn0.thlp ibn.thlp ibn.thlm ibn.thl0 ibm.thvh ibm.thl0av ibw0.qtp ibw.qtp ibn0.qtp ibn.qtp ibn.qtm "
                "pre.up pre.vp pre.wp poi.p poi.pres0 poi.up poi.vp poi.wp out.u0 out.v0 out.w0 "
                "out.um out.pres0").split()

CASES = {
    # name: (mode, iexpnr, nx, ny, nz, kwargs, stretch)
    "k_vreman_12x8x6": ("kernels", 11, 12, 8, 6, dict(sgs="vreman", oracle="nspin = 2"), 1.0),
    "k_smag_8x12x10s": ("kernels", 12, 8, 12, 10, dict(sgs="smag", oracle="nspin = 4"), 1.08),
    "k_noslip_8x8x6": ("kernels", 13, 8, 8, 6, dict(sgs="vreman", bctopm=2, oracle="nspin = 3"), 1.0),
    "k_scalar_8x8x8": ("kernels", 14, 8, 8, 8, dict(sgs="smag", nsv=2, oracle="nspin = 5"), 1.05),
    "k_dns_8x8x6": ("kernels", 15, 8, 8, 6, dict(sgs="dns", lles=False, oracle="nspin = 2"), 1.0),
    "run_16x16x8": ("run", 21, 16, 16, 8, dict(sgs="vreman", oracle="nsub = 9\ndump_at = 1, 3, 9"), 1.0),
    "run_smag_scalar_16x8x12s": ("run", 22, 16, 8, 12,
                                 dict(sgs="smag", nsv=1, oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
    # floor wall function (lbottom, BCbotm = 3: `bottom` -> wfmneutral), the floor of BASELINE configs 1-3
    "k_floor_12x8x6": ("kernels", 16, 12, 8, 6, dict(sgs="vreman", floor=True, oracle="nspin = 3"), 1.0),
    "run_floor_scalar_16x8x12s": ("run", 23, 16, 8, 12,
                                  dict(sgs="smag", nsv=1, floor=True, dx=0.3, oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
    # masscorr: prescribed volume-flow rate in u (and v), src/modforces.f90:328-497
    "k_volflow_12x8x6": ("kernels", 17, 12, 8, 6,
                         dict(sgs="vreman", floor=True, physics="luvolflowr = .true.\nuflowrate = 1.2",
                              oracle="nspin = 3"), 1.04),
    "run_volflow_uv_16x16x8": ("run", 24, 16, 16, 8,
                               dict(sgs="vreman", physics="luvolflowr = .true.\nuflowrate = 1.1\n"
                                    "lvvolflowr = .true.\nvflowrate = 0.05",
                                    oracle="nsub = 6\ndump_at = 1, 3, 6"), 1.0),
    # passive temperature equation (ltempeq, iadv_thl = cd2 -> advecc_2nd): stratified profile, radiative source,
    # floor flux wtsurf, top flux wttop / top value thl_top
    "k_thl_12x8x6": ("kernels", 18, 12, 8, 6,
                     dict(sgs="vreman", floor=True, physics="ltempeq = .true.\nlbuoyancy = .false.",
                          bc="BCtopT = 1\nwttop = -0.002\nBCbotT = 1\nwtsurf = 0.01", oracle="nspin = 3"), 1.04),
    "run_thl_16x8x12s": ("run", 25, 16, 8, 12,
                         dict(sgs="smag", nsv=1, floor=True, physics="ltempeq = .true.\nlbuoyancy = .false.",
                              bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.02",
                              oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
}
CASES.update({
    # buoyancy (lbuoyancy, dry air): forces' wp += grav (thv0h - thvh)/thvh with the reference's thermodynamics
    "k_buoy_12x8x6": ("kernels", 19, 12, 8, 6,
                      dict(sgs="vreman", floor=True, physics="ltempeq = .true.\nlbuoyancy = .true.",
                           bc="BCtopT = 1\nBCbotT = 1\nwtsurf = 0.05\nthls = 288.0", oracle="nspin = 3"), 1.04),
    "run_buoy_16x8x12s": ("run", 26, 16, 8, 12,
                          dict(sgs="smag", floor=True, physics="ltempeq = .true.\nlbuoyancy = .true.",
                               bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.05\nthls = 288.0",
                               oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
})
CASES.update({
    # coriolis (lcoriol) and the relaxation towards a geostrophic profile (lprofforc, as examples/024)
    "k_coriol_12x8x6": ("kernels", 20, 12, 8, 6,
                        dict(sgs="vreman", floor=True, physics="lcoriol = .true.", oracle="nspin = 3"), 1.04),
    "run_profforc_16x16x8": ("run", 27, 16, 16, 8,
                             dict(sgs="vreman", physics="lprofforc = .true.", oracle="nsub = 6\ndump_at = 3, 6"), 1.0),
})
CASES.update({
    # one-equation TKE closure (loneeqn): closure from e120, advecc_2nd + diffe + sources on e12, clipped integration;
    # neutral, and with a stratified buoyant temperature (stability-limited length scale, buoyancy production)
    "k_tke_12x8x6": ("kernels", 21, 12, 8, 6,
                     dict(sgs="oneeqn", floor=True, bc="thls = 288.0\nqts = 0.0", oracle="nspin = 3"), 1.04),
    # (kernel vectors only: the reference's xm_periodic / ym_periodic index e120 with the loop variable AFTER the
    #  loop -- src/modboundary.f90:527-536, 615-624 -- so its e120 ghost cells are never refreshed and a multi-substep
    #  reference run of loneeqn is not a meaningful target; each routine is pinned on the inputs the reference had)
    "k_tke_thl_12x8x6": ("kernels", 28, 12, 8, 6,
                         dict(sgs="oneeqn", floor=True, physics="ltempeq = .true.\nlbuoyancy = .true.",
                              bc="BCtopT = 2\nthl_top = 291.5\nBCbotT = 1\nwtsurf = 0.03\nthls = 288.0\nqts = 0.0",
                              oracle="nspin = 3"), 1.06),
})
CASES.update({
    # per-level forcings: large-scale subsidence (lstend, wfls), nudging, gravity-wave sponge (grwdamp)
    "k_lsf_12x8x24": ("kernels", 29, 12, 8, 24,
                      dict(sgs="vreman", physics="ltempeq = .true.\nlbuoyancy = .true.\nlnudge = .true.\ntnudge = 50.\n"
                           "nnudge = 2\nigrw_damp = 2", bc="BCtopT = 2\nthl_top = 295.\nthls = 288.0\nqts = 0.0",
                           oracle="nspin = 3"), 1.03),
    "run_lsf_16x8x24s": ("run", 30, 16, 8, 24,
                         dict(sgs="smag", physics="ltempeq = .true.\nlbuoyancy = .true.\nlnudge = .true.\ntnudge = 40.\n"
                              "nnudge = 3\nigrw_damp = 1\nlcoriol = .true.",
                              bc="BCtopT = 2\nthl_top = 295.\nthls = 288.0\nqts = 0.0", oracle="nsub = 6\ndump_at = 3, 6"), 1.04),
})
CASES.update({
    # total water (lmoist) as a transported field: advecc_2nd + diffc, floor flux wqsurf, top flux wqtop / top value qt_top
    "k_qt_12x8x6": ("kernels", 32, 12, 8, 6,
                    dict(sgs="vreman", floor=True, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .false.",
                         bc="BCtopT = 1\nBCbotT = 1\nwtsurf = 0.01\nthls = 288.0\nqts = 0.008\n"
                            "BCtopq = 1\nBCbotq = 1\nwqsurf = 2.e-5", oracle="nspin = 3"), 1.04),
    "run_qt_16x8x12s": ("run", 33, 16, 8, 12,
                        dict(sgs="smag", nsv=1, floor=True, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .false.",
                             bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.02\nthls = 288.0\nqts = 0.008\n"
                                "BCtopq = 2\nqt_top = 0.004\nBCbotq = 1\nwqsurf = 3.e-5",
                             oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
CASES.update({
    # lstend / nudge / grwdamp acting on total water, with large-scale moisture gradients and tendency
    "k_lsfq_12x8x20": ("kernels", 34, 12, 8, 20,
                       dict(sgs="vreman", nsv=1, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .false.\nlnudge = .true.\n"
                            "tnudge = 45.\nnnudge = 1\nigrw_damp = 3",
                            bc="BCtopT = 2\nthl_top = 295.\nthls = 288.0\nqts = 0.008\nBCtopq = 2\nqt_top = 0.002",
                            oracle="nspin = 3"), 1.03),
})
CASES.update({
    # moist thermodynamics (lmoist with lbuoyancy): condensate from thermo, hydrostatic pressures / exner functions from
    # diagfld + fromztop, moist thv0h in forces' buoyancy; a profile that is saturated near the floor and dry above
    "k_moist_12x8x8": ("kernels", 35, 12, 8, 8,
                       dict(sgs="vreman", floor=True, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                            bc="BCtopT = 1\nBCbotT = 1\nwtsurf = 0.03\nthls = 288.0\nqts = 0.0105\n"
                               "BCtopq = 1\nBCbotq = 1\nwqsurf = 4.e-5", oracle="nspin = 4"), 1.05),
    "run_moist_16x8x12s": ("run", 36, 16, 8, 12,
                           dict(sgs="smag", floor=True, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                                bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.04\nthls = 288.0\nqts = 0.0105\n"
                                   "BCtopq = 2\nqt_top = 0.0104\nBCbotq = 1\nwqsurf = 5.e-5",
                                oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
})
CASES.update({
    # floor with the stability-dependent wall function (wfuno, Uno et al. 1995): BCbotm = 2 for momentum, BCbotT = 2 for
    # temperature against a wall at thls; an unstable floor (thls above the air) and a stable one (below)
    "k_uno_12x8x6": ("kernels", 37, 12, 8, 6,
                     dict(sgs="vreman", floor=True, bcbotm=2, physics="ltempeq = .true.\nlbuoyancy = .true.", walls="iwalltemp = 2",
                          bc="BCtopT = 1\nBCbotT = 2\nthls = 291.0\nz0h = 0.0067\nqts = 0.0", oracle="nspin = 4"), 1.04),
    "run_uno_16x8x12s": ("run", 38, 16, 8, 12,
                         dict(sgs="smag", floor=True, bcbotm=2, physics="ltempeq = .true.\nlbuoyancy = .true.", walls="iwalltemp = 2",
                              bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 2\nthls = 286.5\nz0h = 0.005\nqts = 0.0",
                              oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
})
CASES.update({
    # Gaussian point and line sources of the scalars (scalsource)
    "k_src_12x8x8": ("kernels", 39, 12, 8, 8,
                     dict(sgs="vreman", nsv=2, floor=True, scalars="lscasrc = .true.\nnscasrc = 2\nlscasrcl = .true.\nnscasrcl = 1",
                          oracle="nspin = 3"), 1.04),
    "run_src_16x8x12s": ("run", 40, 16, 8, 12,
                         dict(sgs="smag", nsv=1, floor=True, scalars="lscasrc = .true.\nnscasrc = 1\nlscasrcl = .true.\nnscasrcl = 2",
                              oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
CASES.update({
    # kappa advection of the temperature (iadv_thl = 7: advecc_kappa on the wide copy thl0c), flux top and value top
    "k_thlk_12x8x6": ("kernels", 41, 12, 8, 6,
                      dict(sgs="vreman", floor=True, physics="ltempeq = .true.\nlbuoyancy = .true.", dynamics="iadv_thl = 7",
                           bc="BCtopT = 1\nwttop = -0.002\nBCbotT = 1\nwtsurf = 0.02\nthls = 288.0\nqts = 0.0", oracle="nspin = 4"), 1.04),
    "run_thlk_16x8x12s": ("run", 42, 16, 8, 12,
                          dict(sgs="smag", nsv=1, floor=True, physics="ltempeq = .true.\nlbuoyancy = .true.", dynamics="iadv_thl = 7",
                               bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.03\nthls = 288.0\nqts = 0.0",
                               oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
CASES.update({
    # fixuinf1 / fixuinf2: the top-level mean velocity pulled back to Uinf, directly or through the dp/dx ODE
    "k_fix1_12x8x6": ("kernels", 43, 12, 8, 6,
                      dict(sgs="vreman", floor=True, physics="ifixuinf = 1\nlvinf = .true.", inlet="Uinf = 1.1\nVinf = 0.02",
                           oracle="nspin = 5"), 1.04),
    "run_fix2_16x8x12s": ("run", 44, 16, 8, 12,
                          dict(sgs="smag", floor=True, physics="ifixuinf = 2\ntscale = 3.0", inlet="Uinf = 1.15\ninletav = 2.0",
                               oracle="nsub = 9\ndump_at = 3, 6, 9"), 1.06),
})
CASES.update({
    # the condensate by Newton-Raphson on the temperature (lqlnr) instead of the one-step formula
    "run_moistnr_16x8x12s": ("run", 45, 16, 8, 12,
                             dict(sgs="smag", floor=True, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.", dynamics="lqlnr = .true.",
                                  bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.04\nthls = 288.0\nqts = 0.0105\n"
                                     "BCtopq = 2\nqt_top = 0.0104\nBCbotq = 1\nwqsurf = 5.e-5",
                                  oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
CASES.update({
    # adaptive time step (tstep_update: Courant / diffusion-number limits, dt = dtmax/100 at the cold start)
    "run_adaptive_16x8x12s": ("run", 46, 16, 8, 12,
                              dict(sgs="smag", nsv=1, floor=True, ladaptive=True, dtmax=2.0, oracle="nsub = 18\ndump_at = 3, 9, 18"), 1.06),
})
CASES.update({
    # top conditions of the scalars: a prescribed value (BCtops = 2, sv_top = svprof(ke)) and prescribed fluxes (wsvtopdum)
    "k_svtop_8x8x8": ("kernels", 47, 8, 8, 8,
                      dict(sgs="smag", nsv=2, bc="BCtops = 2", oracle="nspin = 4"), 1.05),
    "run_svflux_16x8x12s": ("run", 48, 16, 8, 12,
                            dict(sgs="smag", nsv=2, floor=True, bc="BCtops = 1\nwsvtopdum = -0.004, 0.006", oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
CASES.update({
    # shifted periodic boundary conditions (&BC ds > 0, shiftedPBCs): a spanwise drift in the downstream half
    "run_shift_16x8x12s": ("run", 49, 16, 8, 12,
                           dict(sgs="smag", floor=True, bc="ds = 1.5", oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
CASES.update({
    # one-equation closure in moist air: calthv's dthvdz takes its saturated branch inside the cloud
    # (kernel vectors only, like the other loneeqn cases: the reference never refreshes e120's lateral ghosts)
    "k_tke_moist_12x8x8": ("kernels", 50, 12, 8, 8,
                           dict(sgs="oneeqn", floor=True, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                                bc="BCtopT = 1\nBCbotT = 1\nwtsurf = 0.03\nthls = 288.0\nqts = 0.0105\n"
                                   "BCtopq = 1\nBCbotq = 1\nwqsurf = 4.e-5", oracle="nspin = 4"), 1.05),
})
CASES.update({
    # NO - NO2 - O3 chemistry on scalars 1-3 (lchem): fully implicit step on RK stage 3
    "run_chem_16x8x12s": ("run", 51, 16, 8, 12,
                          dict(sgs="smag", nsv=3, floor=True, chemistry="lchem = .true.\nk1 = 0.4\nJNO2 = 0.008",
                               oracle="nsub = 6\ndump_at = 3, 6\nscal_a = 30.\nscal_b = 20."), 1.06),
})
CASES.update({
    # Vreman closure with the buoyancy correction for stable stratification (lbuoycorr with lbuoyancy): a weakly stable
    # layer, so that Rig <= 0 (no correction), 0 < Rig < Rigc (damped) and Rig >= Rigc (ekm = molecular) all occur
    "k_vreman_buoycorr_12x8x10": ("kernels", 52, 12, 8, 10,
                                  dict(sgs="vreman_bc", floor=True, randu=0.05, physics="ltempeq = .true.\nlbuoyancy = .true.",
                                       bc="BCtopT = 1\nBCbotT = 1\nwtsurf = -0.002\nthls = 288.0", oracle="nspin = 3"), 1.04),
    "run_vreman_buoycorr_16x8x12s": ("run", 53, 16, 8, 12,
                                     dict(sgs="vreman_bc", floor=True, randu=0.05, physics="ltempeq = .true.\nlbuoyancy = .true.",
                                          bc="BCtopT = 2\nthl_top = 288.03\nBCbotT = 1\nwtsurf = -0.002\nthls = 288.0",
                                          oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
})
IBM_BLOCKS = {"k_ibm_16x12x10": [(5, 8, 4, 7, 4), (11, 13, 8, 10, 2)], "run_ibm_16x12x10": [(5, 8, 4, 7, 4), (11, 13, 8, 10, 2)]}
CASES.update({
    # immersed boundary, sparse corrections (libm, iwallmom = 1: no facet wall functions): two blocks on the floor, one
    # kappa-advected scalar.  Per-routine dumps around ibmwallfun (diff*_corr) and ibmnorm (solid), and a 9-substep run.
    "k_ibm_16x12x10": ("kernels", 54, 16, 12, 10, dict(sgs="smag", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["k_ibm_16x12x10"],
                                                       oracle="nspin = 4"), 1.04),
    "run_ibm_16x12x10": ("run", 55, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                     oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
})
IBM_BLOCKS["run_ibm_volflow_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
CASES.update({
    # immersed boundary with a prescribed volume flow: masscorr's avexy_ibm averages over the fluid cells (IIu, IIv)
    "run_ibm_volflow_16x12x10": ("run", 56, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                             physics="luvolflowr = .true.\nuflowrate = 1.1\nlvvolflowr = .true.\nvflowrate = 0.04",
                                                             oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
})
# immersed boundary against the periodic boundaries: one block in the first rows of y (its solid v points include j = 1,
# whose periodic image is the ghost row fillps' divergence reads), one in the last rows of y and the last columns of x
IBM_BLOCKS["run_ibm_edge_16x12x10"] = [(4, 7, 1, 3, 4), (14, 16, 10, 12, 3)]
CASES.update({
    "run_ibm_edge_16x12x10": ("run", 57, 16, 12, 10, dict(sgs="vreman", nsv=0, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_edge_16x12x10"],
                                                          oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
})
# masscorr's outflow-rate branch (luoutflowr, src/modforces.f90:352-387): u is corrected so that the flow through the plane i = ie
# (over its fluid u points, per area of its fluid c cells -- uoutletarea :499) stays at uflowrate; plain channel, and with a block
# standing in that plane.  (The v counterpart, lvoutflowr :424-465, hands a (kb:ke) array to sumy_ibm's (ib:ie, kb:ke) dummy -- it
# overruns: not a usable branch of the reference.)
IBM_BLOCKS["run_ibm_uoutflow_16x12x10"] = IBM_BLOCKS["run_ibm_edge_16x12x10"]
CASES.update({
    "run_uoutflow_16x16x8": ("run", 82, 16, 16, 8, dict(sgs="vreman", floor=True, randu=0.05, physics="luoutflowr = .true.\nuflowrate = 1.1",
                                                        oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
    "run_ibm_uoutflow_16x12x10": ("run", 83, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_edge_16x12x10"],
                                                              physics="luoutflowr = .true.\nuflowrate = 1.05", oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
})
# BCzp = 2: the Poisson solve with a cosine transform in z (equidistant levels) instead of the tridiagonal solve, src/modpois.f90:179-191, 559-590
CASES.update({
    "k_bczp2_12x8x8": ("kernels", 88, 12, 8, 8, dict(sgs="vreman", floor=True, randu=0.05, bc="BCzp = 2", oracle="nspin = 3"), 1.0),
    "run_bczp2_16x16x8": ("run", 89, 16, 16, 8, dict(sgs="smag", nsv=1, floor=True, randu=0.05, bc="BCzp = 2", oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
})
# the open lid, BCtopm = 3 (BCtopm_pressure): w(ke+1) is prognostic -- bcpup's row from the slab mean of pres0(ke), tderive's from the
# mean of p(ke), tstep_integrate's plane; per-routine vectors after 4 substeps (so that pres0 and w(ke+1) are no longer zero), a run
# with a kappa-advected scalar (its top flux now sees w(ke+1)), and a run with an immersed boundary one block of which reaches the
# lid: avexy_ibm's mean then runs over the fluid c cells of level ke only (and, run_ibmtall, the same obstacles with a scalar under a
# closed lid)
IBM_BLOCKS["run_ptop_ibm_16x12x10"] = [(5, 8, 4, 7, 10), (11, 13, 8, 10, 2)]
IBM_BLOCKS["run_ibmtall_16x12x10"] = IBM_BLOCKS["run_ptop_ibm_16x12x10"]      # the same obstacles under a closed (free-slip) lid
CASES.update({
    # inflow / outflow in x (&BC BCxm = 2: xmi_profile, xmo_convective, bcpup's and bcp's profile branches, the cosine transform in x;
    # the reference opens the lid itself, src/modstartup.f90:845-848), prof.inp's u = 1, v = 0.1
    "k_xopen_16x8x12": ("kernels", 90, 16, 8, 12, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, bc="BCxm = 2", oracle="nspin = 4"), 1.04),
    "run_xopen_16x8x12s": ("run", 91, 16, 8, 12, dict(sgs="smag", floor=True, bctopm=3, randu=0.05, bc="BCxm = 2", oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
    "run_xopen_volflow_16x8x12s": ("run", 99, 16, 8, 12, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, bc="BCxm = 2", physics="luvolflowr = .true.\nuflowrate = 1.05",
                                                              oracle="nsub = 9\ndump_at = 3, 9"), 1.06),      # masscorr's volume flow; the outlet convects with ubulk
    # ... with the temperature: BCxT = 2 (xTi_profile, xTo_convective), buoyancy, a heated floor, a stratified inflow
    "k_xopen_thl_16x8x12": ("kernels", 100, 16, 8, 12, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                            bc="BCxm = 2\nBCxT = 2\nBCtopT = 1\nwttop = 0.\nBCbotT = 1\nwtsurf = 0.02\nthls = 288.0", oracle="nspin = 4"), 1.04),
    "run_xopen_thl_16x8x12s": ("run", 101, 16, 8, 12, dict(sgs="smag", floor=True, bctopm=3, randu=0.05, physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                           bc="BCxm = 2\nBCxT = 2\nBCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.02\nthls = 288.0",
                                                           oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
    # ... with the temperature and the total water left periodic in x (BCxT = BCxq = 1, the reference's defaults -- what its
    # tests/cases/525 runs with next to BCxm = 3): halos' xT_periodic / xq_periodic refresh their x ghosts after every integration, the
    # moist thermodynamics' slab averages run over ib .. ie
    "k_xopen_moist_16x8x12": ("kernels", 106, 16, 8, 12, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                                                              bc="BCxm = 2\nBCtopT = 1\nBCbotT = 1\nwtsurf = 0.03\nthls = 288.0\nqts = 0.0105\nBCtopq = 1\nBCbotq = 1\nwqsurf = 4.e-5",
                                                              oracle="nspin = 4"), 1.04),
    "run_xopen_moist_16x8x12s": ("run", 107, 16, 8, 12, dict(sgs="smag", floor=True, bctopm=3, randu=0.05, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                                                             bc="BCxm = 2\nBCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.04\nthls = 288.0\nqts = 0.0105\nBCtopq = 2\nqt_top = 0.0104\nBCbotq = 1\nwqsurf = 5.e-5",
                                                             oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
    # ... and with both entering: BCxT = 2, BCxq = 2 (xqi_profile mirrors the ghost about the profile; xqo_convective)
    "run_xopen_qt2_16x8x12s": ("run", 111, 16, 8, 12, dict(sgs="smag", floor=True, bctopm=3, randu=0.05, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                                                           bc="BCxm = 2\nBCxT = 2\nBCxq = 2\nBCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.04\nthls = 288.0\nqts = 0.0105\nBCtopq = 2\nqt_top = 0.0104\nBCbotq = 1\nwqsurf = 5.e-5",
                                                           oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
    # ... with passive scalars: BCxs = 2 (xsi_profile, xso_convective), one and two scalars, the second deck with obstacles and wall functions
    "k_xopen_sv_16x8x12": ("kernels", 102, 16, 8, 12, dict(sgs="vreman", nsv=2, floor=True, bctopm=3, randu=0.05, bc="BCxm = 2\nBCxs = 2", oracle="nspin = 4"), 1.04),
    "run_xopen_sv_16x8x12s": ("run", 103, 16, 8, 12, dict(sgs="smag", nsv=1, floor=True, bctopm=3, randu=0.05, bc="BCxm = 2\nBCxs = 2", oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
    "run_xopen_vr_24x8x10": ("run", 92, 24, 8, 10, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, bc="BCxm = 2", dx=0.4, oracle="nsub = 12\ndump_at = 6, 12"), 1.0),
    "k_ptop_12x8x6": ("kernels", 84, 12, 8, 6, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, oracle="nspin = 4"), 1.04),
    "run_ptop_16x8x12s": ("run", 85, 16, 8, 12, dict(sgs="smag", nsv=1, floor=True, bctopm=3, randu=0.05, oracle="nsub = 9\ndump_at = 3, 9"), 1.06),
    "run_ibmtall_16x12x10": ("run", 87, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, bctopm=1, randu=0.05, ibm=[(5, 8, 4, 7, 10), (11, 13, 8, 10, 2)],
                                                         oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
    # (with a kappa-advected scalar: advecc_kappa leaves cf w0(ke+1) dzfci(ke+1) in the out-of-domain plane svp(ke+1) -- src/modadvection.f90:400,
    #  zero under a closed lid -- and ibmnorm's `solid` averages it into a solid cell of level ke, src/modibm.f90:796-800: the device writes that
    #  plane too, lid_kappa_plane_kernel; and with the temperature, advected by advecc_2nd, which does not)
    "run_ptop_ibm_16x12x10": ("run", 86, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, bctopm=3, randu=0.05, ibm=IBM_BLOCKS["run_ptop_ibm_16x12x10"],
                                                          physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                          bc="BCtopT = 1\nwttop = 0.\nBCbotT = 1\nwtsurf = 0.02\nthls = 288.0",
                                                          oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
})
# immersed boundary with temperature (buoyant), a kappa-advected scalar, and moisture: ibmnorm's `solid`
# on thl with the volume-mean value, advecc2nd_corr (liberal, and conservative with lconservativeibm), diffc_corr on thl / qt,
# the masked slab averages of thermodynamics (IIw for thvh).  Adiabatic, impermeable walls (iwalltemp = iwallmoist = 1 with
# zero prescribed fluxes): wallfunheat, which cannot be built here, then adds exactly nothing.
IBM_BLOCKS["k_ibm_thl_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
IBM_BLOCKS["run_ibm_thl_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
IBM_BLOCKS["run_ibm_thlcons_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
IBM_BLOCKS["run_ibm_qt_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
_IBM_THL_BC = "BCtopT = 1\nwttop = 0.\nBCbotT = 1\nwtsurf = 0.02\nthls = 288.0"
CASES.update({
    "k_ibm_thl_16x12x10": ("kernels", 58, 16, 12, 10, dict(sgs="smag", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                           physics="ltempeq = .true.\nlbuoyancy = .true.", bc=_IBM_THL_BC,
                                                           oracle="nspin = 4"), 1.04),
    "run_ibm_thl_16x12x10": ("run", 59, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                         physics="ltempeq = .true.\nlbuoyancy = .true.", bc=_IBM_THL_BC,
                                                         oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
    "run_ibm_thlcons_16x12x10": ("run", 60, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                             physics="ltempeq = .true.\nlbuoyancy = .true.\nlconservativeibm = .true.",
                                                             bc=_IBM_THL_BC, oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
    "run_ibm_qt_16x12x10": ("run", 61, 16, 12, 10, dict(sgs="smag", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                        physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .false.",
                                                        bc=_IBM_THL_BC + "\nqts = 0.008\nBCtopq = 1\nBCbotq = 1\nwqsurf = 2.e-5",
                                                        oracle="nsub = 9\ndump_at = 3, 9"), 1.04),
})
# statistics (src/modstatsdump.f90 statsdump, its sampling half compiled from the reference: oracle/extract_statsdump.sh):
# the running time averages of tdump / xytdump after 8 steps.  First deck: a sample every second step, a dump (clock
# restart) after the sixth; second deck: obstacles (masked slab averages, -999 on levels without fluid), buoyancy, a sample
# every step (tsample <= dt, :802-803)
IBM_BLOCKS["run_stats_ibm_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
IBM_BLOCKS["run_ytstats_ibm_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
CASES.update({
    "run_stats_16x8x12s": ("run", 62, 16, 8, 12,
                           dict(sgs="smag", nsv=2, floor=True, randu=0.05, physics="ltempeq = .true.\nlbuoyancy = .false.",
                                bc="BCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.02",
                                output="ltdump = .true.\nlxytdump = .true.\ntsample = 0.5\ntstatsdump = 1.5",
                                oracle="nsub = 24\ndump_at = 24"), 1.06),
    # ytdump (y- and time-averaged x-z fields, :964-999, 1104-1132, 1466-1507) on the same obstacle deck
    "run_ytstats_ibm_16x12x10": ("run", 78, 16, 12, 10,
                                 dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                      physics="ltempeq = .true.\nlbuoyancy = .true.", bc=_IBM_THL_BC,
                                      output="lytdump = .true.\ntsample = 0.1\ntstatsdump = 1000.",
                                      oracle="nsub = 15\ndump_at = 15"), 1.04),
    "run_stats_ibm_16x12x10": ("run", 63, 16, 12, 10,
                               dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                    physics="ltempeq = .true.\nlbuoyancy = .true.", bc=_IBM_THL_BC,
                                    output="ltdump = .true.\nlxytdump = .true.\ntsample = 0.1\ntstatsdump = 1000.",
                                    oracle="nsub = 15\ndump_at = 15"), 1.04),
})
# a flat floor with BCbotm = 2 (wfuno) and no temperature equation -- thl0 stays at prof.inp's profile, thls at its default of -1.
# With the wall-function switches at their defaults checkinitvalues turns this floor into the neutral one (src/modstartup.f90:811-816:
# iwallmom = 2 without ltempeq -> iwallmom = 3 AND BCbotm = 3; that is what examples/999 of the reference runs); a deck that names
# iwallmom = 3 itself keeps its BCbotm = 2.  (The decks with the Uno floor and the temperature equation say iwalltemp = 2 for the
# same reason: with the default iwalltemp = 1 the rule fires too.)
CASES.update({
    "k_floor_uno_nothl_12x8x6": ("kernels", 64, 12, 8, 6, dict(sgs="vreman", floor=True, bcbotm=2, bc="z0h = 0.005", walls="iwallmom = 3", oracle="nspin = 3"), 1.04),
    "run_floor_uno_nothl_16x8x12s": ("run", 65, 16, 8, 12, dict(sgs="smag", nsv=1, floor=True, bcbotm=2, bc="z0h = 0.005", walls="iwallmom = 3",
                                                                oracle="nsub = 6\ndump_at = 3, 6"), 1.06),
})
# facet wall functions on the blocks (src/modibm.f90:1286 wallfunmom; the reference's default iwallmom = 2 with the
# stability functions on the facet temperatures, iwallmom = 3 neutral): uniform z (the reference's reconstruction assumes it)
for _n in ("k_ibm_wf3_16x12x10", "k_ibm_wf2_16x12x10", "run_ibm_wf2_16x12x10"):
    IBM_BLOCKS[_n] = IBM_BLOCKS["run_ibm_16x12x10"]
for _n in ("k_ibm_wh2_16x12x10", "run_ibm_wh2_16x12x10"):
    IBM_BLOCKS[_n] = IBM_BLOCKS["run_ibm_16x12x10"]
WF_CASES = {"k_ibm_wf3_16x12x10": 3, "k_ibm_wf2_16x12x10": 2, "run_ibm_wf2_16x12x10": 2, "k_ibm_wh2_16x12x10": 2, "run_ibm_wh2_16x12x10": 2,
            "k_ibm_wh1_16x12x10": 2, "run_ibm_wh1_16x12x10": 2, "run_ground_wf3_16x8x12": 3, "run_ground_wh2_16x8x12": 2}
# the floor as facets instead of lbottom (the reference's examples/001: a flat channel whose ground is the immersed boundary, wall
# functions on every first-level cell), neutral and with the stability functions + heat wall function
for _n in ("run_ground_wf3_16x8x12", "run_ground_wh2_16x8x12"):
    IBM_BLOCKS[_n] = "ground"
CASES.update({
    "run_ground_wf3_16x8x12": ("run", 73, 16, 8, 12, dict(sgs="vreman", nsv=1, randu=0.05, ibm="ground", iwallmom=3,
                                                          oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    "run_ground_wh2_16x8x12": ("run", 74, 16, 8, 12, dict(sgs="smag", nsv=1, randu=0.05, ibm="ground", iwallmom=2, walls="iwalltemp = 2",
                                                          physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                          bc="BCtopT = 1\nwttop = 0.\nthls = 288.0", oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
})
# + prescribed wall heat fluxes (wallfunheat with iwalltemp = 1 and non-zero bctf*: one flux per facet direction, :1510-1524;
#   grid-aligned facets only -- the reference leaves the flux of any other normal undefined)
for _n in ("k_ibm_wh1_16x12x10", "run_ibm_wh1_16x12x10"):
    IBM_BLOCKS[_n] = IBM_BLOCKS["run_ibm_16x12x10"]
    WF_NO_OBLIQUE.add(_n)
_WH1_BC = _IBM_THL_BC + "\nbctfxm = 0.012\nbctfxp = -0.008\nbctfym = 0.02\nbctfyp = 0.005\nbctfz = 0.015"
CASES.update({
    "k_ibm_wh1_16x12x10": ("kernels", 71, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                           iwallmom=2, walls="iwalltemp = 1", physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                           bc=_WH1_BC, oracle="nspin = 4"), 1.0),
    "run_ibm_wh1_16x12x10": ("run", 72, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                         iwallmom=2, walls="iwalltemp = 1", physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                         bc=_WH1_BC, oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
})
# + the heat wall function on the facet temperatures (wallfunheat with iwalltemp = 2, src/modibm.f90:1436)
CASES.update({
    "k_ibm_wh2_16x12x10": ("kernels", 69, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                           iwallmom=2, walls="iwalltemp = 2", physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                           bc=_IBM_THL_BC, oracle="nspin = 4"), 1.0),
    "run_ibm_wh2_16x12x10": ("run", 70, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                         iwallmom=2, walls="iwalltemp = 2", physics="ltempeq = .true.\nlbuoyancy = .true.",
                                                         bc=_IBM_THL_BC, oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
})
CASES.update({
    "k_ibm_wf3_16x12x10": ("kernels", 66, 16, 12, 10, dict(sgs="smag", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                           iwallmom=3, oracle="nspin = 4"), 1.0),
    "k_ibm_wf2_16x12x10": ("kernels", 67, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                           iwallmom=2, physics="ltempeq = .true.\nlbuoyancy = .true.", bc=_IBM_THL_BC,
                                                           oracle="nspin = 4"), 1.0),
    "run_ibm_wf2_16x12x10": ("run", 68, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                         iwallmom=2, physics="ltempeq = .true.\nlbuoyancy = .true.", bc=_IBM_THL_BC,
                                                         oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
})
# moisture next to obstacles: the moist thermodynamics' slab averages over the fluid cells only (diagfld's avexy_ibm on thl0, qt0, ql0
# with IIc / IIcs and thvh with IIw / IIws, src/modthermodynamics.f90:76,262-279), first with impermeable walls, then with the latent
# wall flux of vegetated facets (iwallmoist = 2: wallfunheat's moist_flux, src/modibm.f90:1556-1600,1989; the second facet type green)
GREEN_CASES = {"run_ibm_moistwq_16x12x10", "k_ibm_wq2_16x12x10"}
_IBM_MOIST = dict(sgs="vreman", nsv=1, floor=True, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"], iwallmom=2,
                  physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
                  bc=_IBM_THL_BC + "\nqts = 0.0105\nBCtopq = 1\nBCbotq = 1\nwqsurf = 2.e-5")
for _n in ("run_ibm_moist_16x12x10", "run_ibm_moistwq_16x12x10", "k_ibm_wq2_16x12x10"):
    IBM_BLOCKS[_n] = IBM_BLOCKS["run_ibm_16x12x10"]
    WF_CASES[_n] = 2
# inflow / outflow in x around obstacles (blocks away from the x ends): without wall functions, and with the neutral log law on the facets
for _n in ("run_xopen_ibm_16x12x10", "run_xopen_ibmwf3_16x12x10", "run_xopen_ibm_sv_16x12x10", "run_xopen_ibm_thl_16x12x10", "run_xopen_ibm_moist_16x12x10"):
    IBM_BLOCKS[_n] = IBM_BLOCKS["run_ibm_16x12x10"]
WF_CASES["run_xopen_ibm_moist_16x12x10"] = 2
GREEN_CASES.add("run_xopen_ibm_moist_16x12x10")
WF_CASES["run_xopen_ibm_thl_16x12x10"] = 2
WF_CASES["run_xopen_ibmwf3_16x12x10"] = 3
WF_CASES["run_xopen_ibm_sv_16x12x10"] = 3
CASES.update({
    "run_xopen_ibm_16x12x10": ("run", 93, 16, 12, 10, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, bc="BCxm = 2", ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                           oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    "run_xopen_ibm_sv_16x12x10": ("run", 104, 16, 12, 10, dict(sgs="smag", nsv=1, floor=True, bctopm=3, randu=0.05, bc="BCxm = 2\nBCxs = 2", ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                               iwallmom=3, oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    # heated obstacles in an inflow / outflow: the stability wall functions for momentum and heat on the facet temperatures, buoyancy, a scalar
    "run_xopen_ibm_thl_16x12x10": ("run", 105, 16, 12, 10, dict(sgs="vreman", nsv=1, floor=True, bctopm=3, randu=0.05, ibm=IBM_BLOCKS["run_ibm_16x12x10"], iwallmom=2,
                                                                physics="ltempeq = .true.\nlbuoyancy = .true.", walls="iwalltemp = 2",
                                                                bc="BCxm = 2\nBCxT = 2\nBCxs = 2\n" + _IBM_THL_BC, oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    # moisture next to heated, vegetated obstacles in an inflow / outflow, thl and qt periodic in x (the physics of the reference's tests/cases/525)
    "run_xopen_ibm_moist_16x12x10": ("run", 108, 16, 12, 10, dict(_IBM_MOIST, bctopm=3, bc="BCxm = 2\nBCxs = 2\n" + _IBM_MOIST["bc"], walls="iwalltemp = 2\niwallmoist = 2",
                                                                  extra="&ENERGYBALANCE\nwsoil = 300.\nwfc = 313.\n/", oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    "run_xopen_ibmwf3_16x12x10": ("run", 94, 16, 12, 10, dict(sgs="smag", floor=True, bctopm=3, randu=0.05, bc="BCxm = 2", ibm=IBM_BLOCKS["run_ibm_16x12x10"],
                                                              iwallmom=3, oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
})
CASES.update({
    "run_ibm_moist_16x12x10": ("run", 79, 16, 12, 10, dict(_IBM_MOIST, walls="iwalltemp = 2", oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    "run_ibm_moistwq_16x12x10": ("run", 80, 16, 12, 10, dict(_IBM_MOIST, walls="iwalltemp = 2\niwallmoist = 2",
                                                             extra="&ENERGYBALANCE\nwsoil = 300.\nwfc = 313.\n/", oracle="nsub = 9\ndump_at = 3, 9"), 1.0),
    "k_ibm_wq2_16x12x10": ("kernels", 81, 16, 12, 10, dict(_IBM_MOIST, walls="iwalltemp = 2\niwallmoist = 2",
                                                           extra="&ENERGYBALANCE\nwsoil = 300.\nwfc = 313.\n/", oracle="nspin = 4"), 1.0),
})
# scalars with an inflow / outflow in x while the flow stays periodic (BCxs = 2, the reference's dispersion examples 101 / 102):
# inlet ghosts mirrored about the inflow profile (xsi_profile, src/modboundary.f90:844), a convective outlet ghost (xso_convective
# :983, outflow speed ubulk under luvolflowr), no periodic refresh of the scalars' x ghosts
CASES.update({
    "k_bcxs_16x8x12": ("kernels", 75, 16, 8, 12, dict(sgs="smag", nsv=2, floor=True, randu=0.05, physics="luvolflowr = .true.\nuflowrate = 1.1",
                                                      bc="BCxs = 2", oracle="nspin = 5\nscal_b = 0.2"), 1.05),
    # ... and without a prescribed volume flow: the outlet convects with the mean of diagfld's slab averages (:143-156)
    "run_bcxs_avg_16x8x12s": ("run", 77, 16, 8, 12, dict(sgs="smag", nsv=1, floor=True, randu=0.05, bc="BCxs = 2",
                                                         oracle="nsub = 9\ndump_at = 3, 9\nscal_b = 0.2"), 1.06),
    "run_bcxs_16x8x12s": ("run", 76, 16, 8, 12, dict(sgs="vreman", nsv=1, floor=True, randu=0.05, physics="luvolflowr = .true.\nuflowrate = 1.1",
                                                     bc="BCxs = 2", oracle="nsub = 9\ndump_at = 3, 9\nscal_b = 0.2"), 1.06),
})
LSF_ONLY = ("k_lsf_12x8x24", "k_lsfq_12x8x20", "k_fix1_12x8x6")
THL_CASES = {"k_tke_moist_12x8x8": dict(dthl=0.25, qt=0.0118, dqt=-8e-5, tke=0.06),
             "run_moistnr_16x8x12s": dict(dthl=0.25, qt=0.0119, dqt=-6e-5),
             "k_thlk_12x8x6": dict(dthl=0.3), "run_thlk_16x8x12s": dict(dthl=0.25),
             "k_src_12x8x8": dict(psrc=[[(2.2, 1.3, 0.9, 0.5, 0.6), (4.9, 3.1, 2.2, 0.2, 0.4)], [(1.0, 2.0, 1.5, 1.0, 0.5), (5.5, 0.4, 0.3, 0.3, 0.7)]],
                                  lsrc=[[(0.5, 0.5, 0.6, 5.0, 3.5, 1.4, 0.4, 0.5)], [(3.0, 0.2, 2.0, 3.0, 3.8, 2.0, 0.6, 0.45)]]),
             "run_src_16x8x12s": dict(psrc=[[(3.1, 1.9, 1.2, 0.8, 0.7)]],
                                      lsrc=[[(1.0, 0.6, 0.5, 6.5, 0.6, 0.5, 0.5, 0.5), (6.0, 3.5, 2.5, 2.0, 1.0, 3.5, 0.3, 0.6)]]),
             "k_uno_12x8x6": dict(dthl=0.2), "run_uno_16x8x12s": dict(dthl=0.25),
             "k_moist_12x8x8": dict(dthl=0.25, qt=0.0118, dqt=-8e-5), "run_moist_16x8x12s": dict(dthl=0.25, qt=0.0119, dqt=-6e-5),
             "k_lsfq_12x8x20": dict(dthl=0.3, ug=1.0, wtop=0.025, qt=0.008, dqt=-3e-4, dqtdx=2e-7, dqtdy=-1e-7, dqtdt=3e-8),
             "k_qt_12x8x6": dict(dthl=0.3, qt=0.008, dqt=-4e-4), "run_qt_16x8x12s": dict(dthl=0.25, qt=0.007, dqt=-2e-4),
             "k_lsf_12x8x24": dict(dthl=0.3, ug=1.05, wtop=0.02), "run_lsf_16x8x24s": dict(dthl=0.25, ug=0.95, wtop=-0.03), "k_tke_12x8x6": dict(tke=0.05), "k_tke_thl_12x8x6": dict(tke=0.08, dthl=0.3), "k_coriol_12x8x6": dict(ug=1.1), "run_profforc_16x16x8": dict(ug=1.3), "k_thl_12x8x6": dict(dthl=0.4, dthlrad=2e-3), "run_thl_16x8x12s": dict(dthl=0.25, dthlrad=-1e-3),
             "k_buoy_12x8x6": dict(dthl=0.3), "run_buoy_16x8x12s": dict(dthl=0.2),
             "k_xopen_moist_16x8x12": dict(v=0.1, dthl=0.25, qt=0.0118, dqt=-8e-5), "run_xopen_moist_16x8x12s": dict(v=0.1, dthl=0.25, qt=0.0119, dqt=-6e-5), "run_xopen_qt2_16x8x12s": dict(v=0.1, dthl=0.25, qt=0.0119, dqt=-6e-5),
             "run_xopen_ibm_moist_16x12x10": dict(u=0.9, v=0.15, dthl=0.25, qt=0.0119, dqt=-6e-5),
             "k_xopen_16x8x12": dict(v=0.1), "run_xopen_16x8x12s": dict(v=0.1), "k_xopen_sv_16x8x12": dict(v=0.1), "run_xopen_sv_16x8x12s": dict(v=0.1), "run_xopen_ibm_sv_16x12x10": dict(u=0.9, v=0.15), "run_xopen_ibm_thl_16x12x10": dict(u=0.9, v=0.15, dthl=0.25), "k_xopen_thl_16x8x12": dict(v=0.1, dthl=0.3), "run_xopen_thl_16x8x12s": dict(v=0.1, dthl=0.25), "run_xopen_volflow_16x8x12s": dict(v=0.1), "run_xopen_vr_24x8x10": dict(u=0.8, v=-0.05), "run_xopen_ibm_16x12x10": dict(v=0.1), "run_xopen_ibmwf3_16x12x10": dict(u=0.9, v=0.15),
             "k_ibm_thl_16x12x10": dict(dthl=0.3), "run_ibm_thl_16x12x10": dict(dthl=0.25), "run_ptop_ibm_16x12x10": dict(dthl=0.25), "run_ibm_thlcons_16x12x10": dict(dthl=0.25),
             "run_ibm_qt_16x12x10": dict(dthl=0.25, qt=0.007, dqt=-2e-4),
             "run_stats_16x8x12s": dict(dthl=0.25), "run_stats_ibm_16x12x10": dict(dthl=0.25), "run_ytstats_ibm_16x12x10": dict(dthl=0.25),
             "k_ibm_wf2_16x12x10": dict(dthl=0.3), "run_ibm_wf2_16x12x10": dict(dthl=0.25),
             "k_ibm_wh2_16x12x10": dict(dthl=0.3), "run_ibm_wh2_16x12x10": dict(dthl=0.25),
             "k_ibm_wh1_16x12x10": dict(dthl=0.3), "run_ibm_wh1_16x12x10": dict(dthl=0.25), "run_ground_wh2_16x8x12": dict(dthl=0.25),
             "run_ibm_moist_16x12x10": dict(dthl=0.25, qt=0.0119, dqt=-6e-5), "run_ibm_moistwq_16x12x10": dict(dthl=0.25, qt=0.0119, dqt=-6e-5),
             "k_ibm_wq2_16x12x10": dict(dthl=0.3, qt=0.0118, dqt=-8e-5),
             "k_vreman_buoycorr_12x8x10": dict(dthl=0.004), "run_vreman_buoycorr_16x8x12s": dict(dthl=0.004)}


# restart files written by the reference's own writerestartfiles (src/modsave.f90:37-128): the files are the
# fixture (data), plus the state dumped next to them
RESTART_CASES = {
    "restart_8x8x8": (31, 8, 8, 8, dict(sgs="smag", nsv=2, floor=True, oracle="nsub = 6"), 1.05),
}


def make_restart_cases():
    for name, (iexp, nx, ny, nz, kw, stretch) in RESTART_CASES.items():
        cdir = os.path.join(HERE, "cases", name)
        os.makedirs(cdir, exist_ok=True)
        kw, sc = split_scal(kw)
        write_case(cdir, iexp, deck(iexp, nx, ny, nz, **kw), zlevels(nz, 0.5, stretch), nsv=kw.get("nsv", 0), **sc)
        odir = os.path.join(HERE, name)
        os.makedirs(odir, exist_ok=True)
        with tempfile.TemporaryDirectory() as tmp:
            for fn in os.listdir(cdir):
                shutil.copy(os.path.join(cdir, fn), tmp)
            out = os.path.join(tmp, "out.bin")
            subprocess.check_call([REF, f"namoptions.{iexp:03d}", "restart", out], cwd=tmp, stdout=subprocess.DEVNULL)
            d = read_dump(out)
            keep = {k: v for k, v in d.items() if k.count(".") == 0 or k.startswith("rst.")}
            # the same deck run on without a restart: where a warm start from the files above has to arrive
            with open(os.path.join(tmp, f"namoptions.{iexp:03d}")) as f:
                txt = f.read().replace("nsub = 6", "nsub = 9\ndump_at = 9")
            with open(os.path.join(tmp, f"namoptions.{iexp:03d}"), "w") as f:
                f.write(txt)
            out2 = os.path.join(tmp, "out2.bin")
            subprocess.check_call([REF, f"namoptions.{iexp:03d}", "run", out2], cwd=tmp, stdout=subprocess.DEVNULL)
            d2 = read_dump(out2)
            keep.update({k: v for k, v in d2.items()
                         if k.startswith("s009.") and k.split(".")[1] in ("u0", "v0", "w0", "pres0", "sv0_01", "sv0_02")})
            tmpf = os.path.join(odir, "state.bin")
            write_dump(tmpf, keep)
            with open(tmpf, "rb") as f, gzip.GzipFile(tmpf + ".gz", "wb", mtime=0) as g:
                g.write(f.read())
            os.remove(tmpf)
            for fn in sorted(os.listdir(tmp)):
                if fn.startswith("initd") or fn.startswith("inits"):
                    with open(os.path.join(tmp, fn), "rb") as f, \
                            gzip.GzipFile(os.path.join(odir, fn + ".gz"), "wb", mtime=0) as g:
                        g.write(f.read())
        print(f"{name}: {sorted(os.listdir(odir))}")


# A deck of the reference's own examples, run as it is (data files copied from the reference's examples directory: deck,
# prof.inp, lscale.inp): examples/999, the flat neutral channel at 128^3 with the floor wall function at its defaults
# (BCbotm = 2 without temperature equation), the adaptive time step and tdump / xytdump / fielddump output.  The golden holds
# xytdump's table and the clock after `nsub` substeps of the reference binary (the 3-D fields would be 17 MB each).
EXAMPLES = {"example_999": ("999", 999, 75), "example_001": ("001", 1, 75), "example_002": ("002", 2, 75), "example_101": ("101", 101, 75)}
# examples/001: the same channel with the ground as an immersed boundary (128 facets, 16384 boundary points per grid, ~17800
# facet sections on u and v: the output of the reference's pre-processing).  The shipped deck leaves iwallmom at its default
# of 2, which needs a Tfacinit.inp the example does not ship (the reference stops in readfacetfiles) and reads mask_c
# unallocated; the fixture's deck selects the neutral wall function (one line added to &WALLS), everything else is the example's.
EXAMPLE_FILES = {"example_001": ["facets.inp.001", "factypes.inp.001", "facet_sections_u.txt", "facet_sections_v.txt", "facet_sections_w.txt",
                                 "facet_sections_c.txt", "fluid_boundary_u.txt", "fluid_boundary_v.txt", "fluid_boundary_w.txt",
                                 "fluid_boundary_c.txt", "solid_u.txt", "solid_v.txt", "solid_w.txt", "solid_c.txt"]}
EXAMPLE_FILES["example_002"] = [f.replace(".001", ".002") for f in EXAMPLE_FILES["example_001"]]      # 64^3, an array of cubes: 1024 facets
# examples/101: street canyons with heated walls -- temperature with buoyancy, wall functions with the stability functions
# (iwallmom = 2, the default) and the heat wall function on the facet temperatures (iwalltemp = 2), a prescribed volume flow, a
# scalar line source whose plume enters clean (BCxs = 2: inflow profile, convective outflow).  The deck as shipped.
EXAMPLE_FILES["example_101"] = [f.replace(".001", ".101") for f in EXAMPLE_FILES["example_001"]] + ["Tfacinit.inp.101", "scalar.inp.101",
                                                                                                    "scalarsourcel.inp.1.101"]
# (examples/102 is a warm start from the restart files of a 2 x 2 rank run: not runnable on the one rank of this build)
EXAMPLE_PATCH = {}      # (the decks as shipped: checkinitvalues picks the neutral wall function for 001 / 002 itself)


def make_example_cases(only):
    for name, (exdir, iexp, nsub) in EXAMPLES.items():
        if only and name not in only:
            continue
        src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(REF))), "..", "reference", "examples", exdir)
        src = os.path.join("/root/reference/examples", exdir)
        cdir = os.path.join(HERE, "cases", name)
        os.makedirs(cdir, exist_ok=True)
        # the example's input data, kept compressed: the deck, the profiles, the pre-processing's lists (1.4 MB of text for 001)
        for fn in [f"namoptions.{iexp:03d}", f"prof.inp.{iexp:03d}", f"lscale.inp.{iexp:03d}"] + EXAMPLE_FILES.get(name, []):
            with open(os.path.join(src, fn), "rb") as f:
                data = f.read()
            if fn.startswith("namoptions") and name in EXAMPLE_PATCH:
                old, new = EXAMPLE_PATCH[name]
                assert data.decode().count(old) == 1
                data = data.decode().replace(old, new).encode()
            with gzip.GzipFile(os.path.join(cdir, fn + ".gz"), "wb", mtime=0) as gz:
                gz.write(data)
        with tempfile.TemporaryDirectory() as tmp:
            for fn in os.listdir(cdir):
                if fn.endswith(".gz"):
                    with gzip.open(os.path.join(cdir, fn), "rb") as f, open(os.path.join(tmp, fn[:-3]), "wb") as o:
                        o.write(f.read())
                else:
                    shutil.copy(os.path.join(cdir, fn), tmp)
            import re
            with open(os.path.join(tmp, f"namoptions.{iexp:03d}")) as f:
                txt = f.read()
            # one rank here (the stand-in decomposition of this build); + the driver's own group; the deck is otherwise untouched
            txt = re.sub(r"nprocx\s*=\s*\d+", "nprocx = 1", re.sub(r"nprocy\s*=\s*\d+", "nprocy = 1", txt))
            with open(os.path.join(tmp, f"namoptions.{iexp:03d}"), "w") as f:
                f.write(txt + f"\n&ORACLE\nnsub = {nsub}\n/\n")
            out = os.path.join(tmp, "out.bin")
            # (128^3: the reference's array-valued expressions in statsdump need more than the default 8 MB of stack)
            subprocess.check_call(["bash", "-c", f"ulimit -s unlimited; exec {REF} namoptions.{iexp:03d} run {out}"], cwd=tmp)
            d = read_dump(out)
        keep = {k: v for k, v in d.items() if k.count(".") == 0 or k.startswith(("xyt.", "end."))}
        tmpf = os.path.join(HERE, name + ".bin")
        write_dump(tmpf, keep)
        with open(tmpf, "rb") as f, gzip.GzipFile(tmpf + ".gz", "wb", mtime=0) as g:
            g.write(f.read())
        os.remove(tmpf)
        print(f"{name}: {len(keep)} records, {os.path.getsize(tmpf + '.gz') / 1024:.0f} kB")


FULL = os.path.join(ROOT, "oracle", "_ref", "udales_full")


FULL_EXAMPLES = {"999": 128, "002": 64, "101": 64, "001": 128}      # example -> grid size (cubic)


def make_full_example(ex="999"):
    """(examples/002 -- an array of cubes -- and examples/101 -- heated street canyons with wall functions for momentum and heat, a
    prescribed volume flow and a scalar line source with BCxs = 2 -- the same way, 64^3 each.)
    examples/999 through the reference's own EXECUTABLE (oracle/_ref/udales_full: program.f90 and all), command line and files
    as a user has them -- one change to the deck: one rank (the stand-in decomposition of this build), and a shorter run: 11 s, past
    the first xytdump, with a restart file at the end.  Kept: the clock after every step (monitor file), xytdump's first record as
    handed to NetCDF, and of the restart file the clock, the slab means and rms of u0, v0, w0, pres0 and every 8th point of them."""
    import re
    import numpy as np
    from refdump import Field, read_ncrec
    sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
    from udcore import restart
    n = FULL_EXAMPLES[ex]
    cdir = os.path.join(HERE, "cases", f"example_{ex}")
    with tempfile.TemporaryDirectory() as tmp:
        for fn in os.listdir(cdir):
            with gzip.open(os.path.join(cdir, fn), "rb") as f, open(os.path.join(tmp, fn[:-3]), "wb") as o:
                o.write(f.read())
        with open(os.path.join(tmp, f"namoptions.{ex}")) as f:
            txt = f.read()
        txt = re.sub(r"nprocx\s*=\s*\d+", "nprocx = 1", re.sub(r"nprocy\s*=\s*\d+", "nprocy = 1", txt))
        txt = re.sub(r"runtime\s*=\s*[0-9.]+", "runtime = 11.", re.sub(r"trestart\s*=\s*[0-9.]+", "trestart = 10.9", txt))
        with open(os.path.join(tmp, f"namoptions.{ex}"), "w") as f:
            f.write(txt)
        subprocess.check_call(["bash", "-c", f"ulimit -s unlimited; exec {FULL} namoptions.{ex}"], cwd=tmp, stdout=subprocess.DEVNULL)
        keep = {"monitor": Field(np.loadtxt(os.path.join(tmp, "monitor000.txt")), (1,))}
        for k, v in read_ncrec(os.path.join(tmp, f"xytdump.{ex}.nc")).items():
            if k.endswith(("xyt", "txyc", "txy")) or k == "time":
                keep["xyt." + k[:11]] = Field(np.atleast_1d(v[0][1]).astype(float), (1,))
        tfile = os.path.join(tmp, f"tdump.000.000.{ex}.nc")      # tdump's first record: every 16th point and the level means of each variable
        if os.path.exists(tfile):
            for k, v in read_ncrec(tfile).items():
                if k == "time":
                    continue
                a = np.squeeze(np.asarray(v[0][1], dtype=float))
                if a.ndim != 3:      # (the coordinate variables)
                    continue
                keep["tp." + k[:12]] = Field(np.ascontiguousarray(a[::16, ::16, ::16]), (1, 1, 1))
                keep["tm." + k[:12]] = Field(a.mean(axis=(1, 2)), (1,))
        rst = [f for f in os.listdir(tmp) if f.startswith("initd")]
        assert len(rst) == 1
        r = restart.read_initd(os.path.join(tmp, rst[0]), n, n, n)
        keep["rst.time"] = Field(np.array([r["timee"], r["dt"], float(rst[0][5:13])]), (1,))
        for k in ("u0", "v0", "w0", "pres0", "thl0"):
            if k == "thl0" and "ltempeq" not in txt:
                continue
            a = r[k][1:n + 1, 1:n + 1, 1:n + 1]
            keep[f"rst.{k}.mean"] = Field(a.mean(axis=(1, 2)), (1,))
            keep[f"rst.{k}.rms"] = Field(np.sqrt((a ** 2).mean(axis=(1, 2))), (1,))
            keep[f"rst.{k}.pts"] = Field(np.ascontiguousarray(a[::8, ::8, ::8]), (1, 1, 1))
    tmpf = os.path.join(HERE, f"full_example_{ex}.bin")
    write_dump(tmpf, keep)
    with open(tmpf, "rb") as f, gzip.GzipFile(tmpf + ".gz", "wb", mtime=0) as g:
        g.write(f.read())
    os.remove(tmpf)
    print(f"full_example_{ex}: {len(keep)} records, {os.path.getsize(tmpf + '.gz') / 1024:.0f} kB")


# examples/950 of the reference: inflow from a precursor run's driver files (BCxm = 3), 256 x 128 x 128, obstacles with wall functions on
# 6612 facets, a prescribed volume flow, the adaptive time step, tdump / xytdump / fielddump switched on.  The example ships without the
# precursor's driver data (its driver_files/ holds the time stamps only), so a precursor deck derived from the example's own -- the same
# y-z grid, profiles and flow rate, a periodic channel 32 cells long without obstacles, one plane every 0.5 s -- is run through the
# reference first and its four records are kept with the case as data.  Deck changes of the driven run: one rank, three steps, and
# driverstore = 4 (the records there are).  Golden: the clock after every step and the restart file (slab means, rms, every 8th point).
EX950_FILES = ["namoptions.950", "prof.inp.950", "lscale.inp.950", "facets.inp.950", "factypes.inp.950"] + \
              [f"{a}_{g}.txt" for a in ("facet_sections", "fluid_boundary", "solid") for g in "uvwc"]


def ex950_decks(txt):
    """(driven deck, precursor deck) from the example's own"""
    import re
    one = re.sub(r"nprocx\s*=\s*\d+", "nprocx       = 1", re.sub(r"nprocy\s*=\s*\d+", "nprocy       = 1", txt))
    drv = re.sub(r"runtime\s*=\s*[0-9.]+", "runtime      = 1.5\ntrestart     = 1.4", one)
    drv = re.sub(r"driverstore\s*=\s*\d+", "driverstore  = 4", drv)
    pre = re.sub(r"iexpnr\s*=\s*950", "iexpnr       = 949", one)
    pre = re.sub(r"libm\s*=\s*\.true\.", "libm         = .false.", pre)
    pre = re.sub(r"itot\s*=\s*256", "itot         = 32", re.sub(r"xlen\s*=\s*256", "xlen         = 32", pre))
    pre = re.sub(r"BCxm\s*=\s*3", "BCxm         = 1", re.sub(r"BCtopm\s*=\s*3", "BCtopm       = 1", pre))
    pre = re.sub(r"dtmax\s*=\s*[0-9.]+", "dtmax        = 0.5", pre).replace("ladaptive    = .true.", "ladaptive    = .false.")
    pre = re.sub(r"runtime\s*=\s*[0-9.]+", "runtime      = 2.2", pre)
    for grp in ("WALLS", "OUTPUT", "INPS"):
        pre = re.sub(r"&" + grp + r"\b.*?\n/\n", "", pre, flags=re.S)
    pre = re.sub(r"&DRIVER\b.*?\n/\n", "&DRIVER\nidriver      = 1\ntdriverstart = 0.\ndtdriver     = 0.5\ndriverstore  = 4\niplane       = 16\n/\n", pre, flags=re.S)
    assert "idriver      = 1" in pre and "&WALLS" not in pre and "driverstore  = 4" in drv
    return drv, pre


def make_example_950():
    import numpy as np
    from refdump import Field
    sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
    from udcore import restart
    src = "/root/reference/examples/950"
    cdir = os.path.join(HERE, "cases", "example_950")
    os.makedirs(cdir, exist_ok=True)
    for fn in EX950_FILES:
        with open(os.path.join(src, fn), "rb") as f, gzip.GzipFile(os.path.join(cdir, fn + ".gz"), "wb", mtime=0) as gz:
            gz.write(f.read())
    with open(os.path.join(src, "namoptions.950")) as f:
        drv, pre = ex950_decks(f.read())
    with tempfile.TemporaryDirectory() as tmp:
        for fn in EX950_FILES:
            shutil.copy(os.path.join(src, fn), tmp)
        for a in ("prof", "lscale"):
            shutil.copy(os.path.join(src, f"{a}.inp.950"), os.path.join(tmp, f"{a}.inp.949"))
        with open(os.path.join(tmp, "namoptions.949"), "w") as f:
            f.write(pre)
        subprocess.check_call(["bash", "-c", f"ulimit -s unlimited; exec {FULL} namoptions.949"], cwd=tmp, stdout=subprocess.DEVNULL)
        for q in "uvwt":      # the precursor's planes: data of the case
            with open(os.path.join(tmp, f"{q}driver_000.949"), "rb") as f, gzip.GzipFile(os.path.join(cdir, f"{q}driver_000.949.gz"), "wb", mtime=0) as gz:
                gz.write(f.read())
        for fn in os.listdir(tmp):
            if fn.startswith(("initd", "monitor")):
                os.remove(os.path.join(tmp, fn))
        with open(os.path.join(tmp, "namoptions.950"), "w") as f:
            f.write(drv)
        subprocess.check_call(["bash", "-c", f"ulimit -s unlimited; exec {FULL} namoptions.950"], cwd=tmp, stdout=subprocess.DEVNULL)
        keep = {"monitor": Field(np.atleast_1d(np.loadtxt(os.path.join(tmp, "monitor000.txt"))), (1,))}
        rst = [f for f in os.listdir(tmp) if f.startswith("initd") and f.endswith(".950")]
        assert len(rst) == 1, rst
        nx, ny, nz = 256, 128, 128
        r = restart.read_initd(os.path.join(tmp, rst[0]), nx, ny, nz)
        keep["rst.time"] = Field(np.array([r["timee"], r["dt"], float(rst[0][5:13])]), (1,))
        for k in ("u0", "v0", "w0", "pres0"):
            a = r[k][1:nz + 1, 1:ny + 1, :]      # (x ghost columns kept: the inlet's and the outlet's state)
            keep[f"rst.{k}.mean"] = Field(a.mean(axis=(1, 2)), (1,))
            keep[f"rst.{k}.rms"] = Field(np.sqrt((a ** 2).mean(axis=(1, 2))), (1,))
            keep[f"rst.{k}.pts"] = Field(np.ascontiguousarray(a[::8, ::8, ::8]), (1, 1, 1))
            keep[f"rst.{k}.out"] = Field(np.ascontiguousarray(a[:, :, -2:]), (1, 1, 1))      # the outlet: ie, ie+1
    tmpf = os.path.join(HERE, "full_example_950.bin")
    write_dump(tmpf, keep)
    with open(tmpf, "rb") as f, gzip.GzipFile(tmpf + ".gz", "wb", mtime=0) as g:
        g.write(f.read())
    os.remove(tmpf)
    print(f"full_example_950: {len(keep)} records, {os.path.getsize(tmpf + '.gz') / 1024:.0f} kB; case {sum(os.path.getsize(os.path.join(cdir, f)) for f in os.listdir(cdir)) / 1e6:.1f} MB")


# tests/cases/525 of the reference -- trees over a flat floor with BCxm = 3, the temperature and the moisture periodic -- cannot run through
# the solver as shipped, the reference's own included: it is a pre-processing case (tests/integration/directshortwave), its deck says
# runmode = 1005 and facet_sections_u / _v / _c.txt are not in the tree (initibmwallfun stops at src/modibm.f90:352).  Its sibling 526 is
# complete and is the same set-up at a quarter of the resolution with periodic momentum: so the inflow / outflow of 525 is put on 526 --
# BCxm = 3 from the planes of a precursor derived from 526's own deck (a periodic channel 16 cells long on the same y-z grid, no trees, no
# obstacles, fixed step; iexpnr 301 = the driverjobnr both decks name), kept with the case as data.
def c526_open_decks(txt):
    """(driven deck, precursor deck) from namoptions.526.serial"""
    import re
    drv = re.sub(r"BCxm\s*=\s*1", "BCxm         = 3", txt)
    drv = re.sub(r"idriver\s*=\s*0", "idriver      = 2", re.sub(r"driverstore\s*=\s*\d+", "driverstore  = 8", drv))
    pre = re.sub(r"iexpnr\s*=\s*526", "iexpnr       = 301", txt)
    pre = re.sub(r"libm\s*=\s*\.true\.", "libm         = .false.", pre)
    pre = re.sub(r"itot\s*=\s*128", "itot         = 16", re.sub(r"xlen\s*=\s*256", "xlen         = 32", pre))
    pre = re.sub(r"BCtopm\s*=\s*3", "BCtopm       = 1", pre).replace("ladaptive    = .true.", "ladaptive    = .false.")
    pre = re.sub(r"runtime\s*=\s*[0-9.]+", "runtime      = 0.085", pre)
    for grp in ("WALLS", "OUTPUT", "INPS", "TREES"):
        pre = re.sub(r"&" + grp + r"\b.*?\n/\n", "", pre, flags=re.S)
    pre = re.sub(r"&DRIVER\b.*?\n/\n", "&DRIVER\nidriver      = 1\ntdriverstart = 0.\ndtdriver     = 0.01\ndriverstore  = 8\niplane       = 8\n/\n", pre, flags=re.S)
    assert "idriver      = 1" in pre and "&WALLS" not in pre and "&TREES" not in pre and "driverstore  = 8" in drv and "BCxm         = 3" in drv
    return drv, pre


def make_case_526_open():
    src = "/root/reference/tests"
    cdir = os.path.join(HERE, "cases", "case_526_open")
    os.makedirs(cdir, exist_ok=True)
    with open(os.path.join(src, "integration", "processor_boundaries", "namoptions.526.serial")) as f:
        _, pre = c526_open_decks(f.read())
    with tempfile.TemporaryDirectory() as tmp:
        for a in ("prof", "lscale"):
            shutil.copy(os.path.join(src, "cases", "526", f"{a}.inp.526"), os.path.join(tmp, f"{a}.inp.301"))
        with open(os.path.join(tmp, "namoptions.301"), "w") as f:
            f.write(pre)
        subprocess.check_call(["bash", "-c", f"ulimit -s unlimited; exec {FULL} namoptions.301"], cwd=tmp, stdout=subprocess.DEVNULL)
        got = sorted(fn for fn in os.listdir(tmp) if "driver_" in fn)
        for fn in got:
            with open(os.path.join(tmp, fn), "rb") as f, gzip.GzipFile(os.path.join(cdir, fn + ".gz"), "wb", mtime=0) as gz:
                gz.write(f.read())
    print(f"case_526_open: the precursor's planes {got}, {sum(os.path.getsize(os.path.join(cdir, f)) for f in os.listdir(cdir)) / 1e3:.0f} kB")


def split_scal(kw):
    """scal_a / scal_b in a case's `oracle` text describe scalar.inp (they were a group of the driver once): taken out of the deck."""
    import re
    kw = dict(kw)
    sc = {}
    for key in ("scal_a", "scal_b"):
        m = re.search(key + r"\s*=\s*([0-9.eE+-]+)", kw.get("oracle", ""))
        if m:
            sc[key] = float(m.group(1))
            kw["oracle"] = re.sub(r"\n?" + key + r"\s*=\s*[0-9.eE+-]+", "", kw["oracle"])
    return kw, sc


def make_reference_test_cases():
    """The inputs of the reference's OWN solver-facing tests, staged as data (tests/test_reference_own_tests.py):
    tests/cases/100 of the reference (the Xie / Castro cube array, 128^3, immersed boundary: prof / lscale / facet / point-list
    files; the STL the solver never reads is left out), and the decks its test drivers put on top of it --
    tests/integration/processor_boundaries/namoptions.100.serial (one step, tdump) and
    tests/integration/mpi_operators/namoptions.1005.{serial,ysplit} (runmode 1005: src/tests.f90's operator test).  No outputs are
    kept: the all-reference executable (oracle/_ref/udales_full, which travels with the snapshot) produces them on the test box."""
    src = "/root/reference/tests"
    cdir = os.path.join(HERE, "cases", "case_100")
    os.makedirs(cdir, exist_ok=True)
    files = [os.path.join(src, "cases", "100", fn) for fn in sorted(os.listdir(os.path.join(src, "cases", "100"))) if not fn.endswith(".stl")]
    files += [os.path.join(src, "integration", "processor_boundaries", "namoptions.100.serial"),
              os.path.join(src, "integration", "mpi_operators", "namoptions.1005.serial"),
              os.path.join(src, "integration", "mpi_operators", "namoptions.1005.ysplit")]
    for fn in files:
        with open(fn, "rb") as f, gzip.GzipFile(os.path.join(cdir, os.path.basename(fn) + ".gz"), "wb", mtime=0) as g:
            g.write(f.read())
    print(f"case_100: {len(files)} input files staged")
    # tests/cases/526: the other half of the reference's processor_boundaries test -- trees (vegetation.f90) over a flat floor as an
    # immersed boundary, temperature + moisture + buoyancy, the open lid (BCtopm = 3), treedump -- with the test driver's deck
    # namoptions.526.serial.  (The STLs are pre-processing inputs the solver never opens.)
    cdir = os.path.join(HERE, "cases", "case_526")
    os.makedirs(cdir, exist_ok=True)
    files = [os.path.join(src, "cases", "526", fn) for fn in sorted(os.listdir(os.path.join(src, "cases", "526"))) if not fn.endswith(".stl")]
    files += [os.path.join(src, "integration", "processor_boundaries", "namoptions.526.serial")]
    for fn in files:
        with open(fn, "rb") as f, gzip.GzipFile(os.path.join(cdir, os.path.basename(fn) + ".gz"), "wb", mtime=0) as g:
            g.write(f.read())
    print(f"case_526: {len(files)} input files staged")


# Inflow from a precursor run's planes (&BC BCxm = 3, idriver = 2; the class of the reference's examples/950 and tests/cases/525, whose
# own driver files are not shipped): a periodic precursor deck (idriver = 1) writes u/v/w/tdriver_000.<nr> -- kept in the case directory as
# data -- and the driven deck reads them.  Second case: obstacles with the neutral wall functions and a prescribed volume flow, as examples/950.
DRIVER_CASES = {
    # name: (iexpnr, precursor's iexpnr, nx, ny, nz, kwargs of the driven deck, stretch, prof kwargs)
    "run_xdriver_16x8x12s": (96, 95, 16, 8, 12, dict(sgs="smag", floor=True, bctopm=3, randu=0.05, bc="BCxm = 3"), 1.06, dict(v=0.1)),
    "run_xdriver_ibm_16x12x10": (98, 97, 16, 12, 10, dict(sgs="vreman", floor=True, bctopm=3, randu=0.05, bc="BCxm = 3", iwallmom=3,
                                                            physics="luvolflowr = .true.\nuflowrate = 0.95"), 1.0, dict(u=0.9, v=0.15)),
    # ... and the physics of tests/cases/525: temperature and moisture (periodic in x: the deck's defaults BCxT = BCxq = 1), buoyancy, heated
    # and vegetated obstacles with the stability wall functions, the latent flux of the green facets, Vreman
    "run_xdriver_moist_16x12x10": (110, 109, 16, 12, 10, dict(_IBM_MOIST, nsv=0, bctopm=3, bc="BCxm = 3\n" + _IBM_MOIST["bc"], walls="iwalltemp = 2\niwallmoist = 2",
                                                              ENERGYBALANCE="wsoil = 300.\nwfc = 313."), 1.0, dict(u=0.9, v=0.15, dthl=0.25, qt=0.0119, dqt=-6e-5)),
}
# ... and the scalars entering from the precursor's planes too: BCxT = BCxq = BCxs = 3 (xTi_driver, xqi_driver, xsi_driver, the convective
# outlets xTo / xqo / xso_convective), temperature + moisture + buoyancy + a passive scalar
DRIVER_CASES["run_xdriver_scal_16x8x12s"] = (113, 112, 16, 8, 12, dict(sgs="smag", nsv=1, floor=True, bctopm=3, randu=0.05, physics="lmoist = .true.\nltempeq = .true.\nlbuoyancy = .true.",
    bc="BCxm = 3\nBCxT = 3\nBCxq = 3\nBCxs = 3\nBCtopT = 2\nthl_top = 290.5\nBCbotT = 1\nwtsurf = 0.04\nthls = 288.0\nqts = 0.0105\nBCtopq = 2\nqt_top = 0.0104\nBCbotq = 1\nwqsurf = 5.e-5"),
    1.06, dict(v=0.1, dthl=0.25, qt=0.0119, dqt=-6e-5))
IBM_BLOCKS["run_xdriver_ibm_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
WF_CASES["run_xdriver_ibm_16x12x10"] = 3
IBM_BLOCKS["run_xdriver_moist_16x12x10"] = IBM_BLOCKS["run_ibm_16x12x10"]
WF_CASES["run_xdriver_moist_16x12x10"] = 2
GREEN_CASES.add("run_xdriver_moist_16x12x10")


def make_driver_cases(only):
    nstore, nsub = 8, 9
    for name, (iexp, ipre, nx, ny, nz, kw, stretch, prof) in DRIVER_CASES.items():
        if only and name not in only:
            continue
        cdir = os.path.join(HERE, "cases", name)
        os.makedirs(cdir, exist_ok=True)
        zf = zlevels(nz, 0.5, stretch)
        with tempfile.TemporaryDirectory() as tmp:
            # the precursor: periodic channel on the same y-z grid, closed lid, one plane per time step from the start
            pre = dict(sgs=kw["sgs"], floor=True, bctopm=1, randu=0.05,
                       extra=f"&DRIVER\nidriver = 1\ntdriverstart = 0.\ndtdriver = 0.25\ndriverstore = {nstore}\niplane = {nx // 2 + 1}\n/",
                       oracle=f"nsub = {3 * nstore + 3}")
            if "physics" in kw and "lmoist" in kw["physics"]:      # (the precursor carries the same equations)
                pre.update(physics=kw["physics"], bc="\n".join(ln for ln in kw["bc"].split("\n") if not ln.startswith("BCx")), nsv=kw.get("nsv", 0))
            write_case(tmp, ipre, deck(ipre, nx, ny, nz, **pre), zf, nsv=pre.get("nsv", 0), **prof)
            subprocess.check_call([REF, f"namoptions.{ipre:03d}", "run", os.path.join(tmp, "pre.bin")], cwd=tmp, stdout=subprocess.DEVNULL)
            for q in "uvwt" + ("hqs" if "BCxT = 3" in kw.get("bc", "") else ""):
                shutil.copy(os.path.join(tmp, f"{q}driver_000.{ipre:03d}"), cdir)
        ibm = IBM_BLOCKS.get(name)
        kw = dict(kw)
        eb = kw.pop("ENERGYBALANCE", None)
        dk = dict(kw, ibm=ibm, extra=f"&DRIVER\nidriver = 2\ndriverjobnr = {ipre}\ndriverstore = {nstore}\n/" + (f"\n&ENERGYBALANCE\n{eb}\n/" if eb else ""),
                  oracle=f"nsub = {nsub}\ndump_at = 3, {nsub}")
        write_case(cdir, iexp, deck(iexp, nx, ny, nz, **dk), zf, nsv=kw.get("nsv", 0), **prof)
        if ibm:
            write_ibm_files(cdir, ibm, nx, ny, nz)
            write_facet_files(cdir, iexp, ibm, nx, ny, nz, 0.5, 0.5, 0.5, True, name in GREEN_CASES)
        with tempfile.TemporaryDirectory() as tmp:
            for fn in os.listdir(cdir):
                shutil.copy(os.path.join(cdir, fn), tmp)
            out = os.path.join(tmp, "out.bin")
            subprocess.check_call([REF, f"namoptions.{iexp:03d}", "run", out], cwd=tmp, stdout=subprocess.DEVNULL)
            d = read_dump(out)
        keep = {k: v for k, v in d.items() if k.count(".") == 0 or k.split(".")[1] in ("u0", "v0", "w0", "pres0", "um", "vm", "wm", "uouttot", "thl0", "thlm", "qt0", "qtm") or ".sv0" in k or ".svm" in k}
        tmpf = os.path.join(HERE, name + ".bin")
        write_dump(tmpf, keep)
        with open(tmpf, "rb") as f, gzip.GzipFile(tmpf + ".gz", "wb", mtime=0) as g:
            g.write(f.read())
        os.remove(tmpf)
        print(f"{name}: {len(keep)} records, {os.path.getsize(tmpf + '.gz') / 1024:.0f} kB, case files {sorted(os.listdir(cdir))[:4]} ...")


def main():
    if not os.path.exists(REF):
        sys.exit(f"{REF} missing: run `make -C oracle ref` in the build container first")
    only = set(sys.argv[1:])          # optional: regenerate only the named cases
    for name, (mode, iexp, nx, ny, nz, kw, stretch) in CASES.items():
        if only and name not in only:
            continue
        cdir = os.path.join(HERE, "cases", name)
        os.makedirs(cdir, exist_ok=True)
        kw, sc = split_scal(kw)
        write_case(cdir, iexp, deck(iexp, nx, ny, nz, **kw), zlevels(nz, 0.5, stretch), nsv=kw.get("nsv", 0), **sc, **THL_CASES.get(name, {}))
        if name in IBM_BLOCKS:
            write_ibm_files(cdir, IBM_BLOCKS[name], nx, ny, nz)
        if name in WF_CASES:
            write_facet_files(cdir, iexp, IBM_BLOCKS[name], nx, ny, nz, kw.get("dx", 0.5), kw.get("dy", 0.5), 0.5, name not in WF_NO_OBLIQUE, name in GREEN_CASES)
        with tempfile.TemporaryDirectory() as tmp:
            for fn in os.listdir(cdir):
                shutil.copy(os.path.join(cdir, fn), tmp)
            out = os.path.join(tmp, "out.bin")
            subprocess.check_call([REF, f"namoptions.{iexp:03d}", mode, out], cwd=tmp)
            d = read_dump(out)
        if mode == "kernels":
            keep = {k: v for k, v in d.items()
                    if k in KEEP_KERNELS or ".sv" in k}
            if name in LSF_ONLY:      # only what tests/test_level_forcings.py reads
                keep = {k: v for k, v in keep.items() if k.count(".") == 0 or k.startswith(("frc0.", "lsf.") if "lsf" in name else ("fix0.", "pre.u", "pre.v"))
                        or k in ("in.v0", "in.w0", "in.um", "in.vm", "in.wm", "in.pres0", "sub.u0", "sub.thl0", "in.thlm",
                                 "sub.qt0", "in.qtm", "in.sv0_01")}
        else:
            keep = {k: v for k, v in d.items()
                    if k.count(".") == 0 or k.split(".")[1] in ("u0", "v0", "w0", "pres0", "um", "thl0", "thlm", "e120", "e12m", "qt0", "qtm", "dpdxl", "time")
                    or (k.startswith("s000.") and not k.startswith("s000.ek")) or ".sv0" in k or k.startswith(("st.", "xyt.", "yt."))
                    or ("xopen" in name and k.split(".")[1] in ("vm", "wm", "uouttot"))}      # (thl0, thlm are kept for every case)
            # (s000.ekm/ekh are dumped before the first closure call: uninitialised memory, not data)
        tmpf = os.path.join(HERE, name + ".bin")
        write_dump(tmpf, keep)
        with open(tmpf, "rb") as f, gzip.GzipFile(tmpf + ".gz", "wb", mtime=0) as g:
            g.write(f.read())
        os.remove(tmpf)
        print(f"{name}: {len(keep)} records, {os.path.getsize(tmpf + '.gz') / 1024:.0f} kB")
    if only == {"full_example_950"}:
        make_example_950()
        return
    if only == {"case_526_open"}:
        make_case_526_open()
        return
    if not only or any(n in DRIVER_CASES for n in only):
        make_driver_cases(only)
    if only and all(n in DRIVER_CASES for n in only):
        return
    make_restart_cases()
    make_example_cases(only)
    if not only or "case_100" in only or "case_526" in only:
        make_reference_test_cases()
    if not only or "case_526_open" in only:
        make_case_526_open()
    for ex in FULL_EXAMPLES:
        if not only or f"full_example_{ex}" in only:
            make_full_example(ex)
    if not only or "full_example_950" in only:      # (~6 minutes of the reference's program)
        make_example_950()


if __name__ == "__main__":
    main()
