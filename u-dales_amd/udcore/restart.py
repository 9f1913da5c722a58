"""Restart files in the reference's own layout: lets a device run warm-start from, and hand
back to, CPU uDALES.

Layout (writerestartfiles, src/modsave.f90:80-121; read back by readrestartfiles, src/modstartup.f90:2194-2217):
one Fortran *sequential unformatted* file per rank, `initd<ntrun:8>_<myidx:3>_<myidy:3>.<expnr:3>`, records

    mindist(ib:ie, jb:je, kb:ke)                 real(8)
    wall(ib:ie, jb:je, kb:ke, 1:5)               integer(4)
    u0, v0, w0, pres0, thl0, e120, ekm, qt0, ql0, ql0h   each (ib-ih:ie+ih, jb-jh:je+jh, kb:ke+kh), real(8)
    timee, dt                                    real(8)

and, when nsv > 0, `inits...` with sv0(ib-ih:ie+ih, jb-jh:je+jh, kb:ke+kh, 1:nsv) and timee.  Every record is
framed by 4-byte length markers (gfortran/flang convention).  With the y-slab decomposition (nprocx = 1) rank r
writes myidx = 0, myidy = r.

Arrays here are numpy [k, j, i] "m-arrays" with one ghost cell on every side (shape (nz+2, ny+2, nx+2)), the
convention of udcore.grid / DynCore.download; the file holds their k = 1..nz+1 planes.
"""
from __future__ import annotations

import os
import struct

import numpy as np

M_FIELDS = ("u0", "v0", "w0", "pres0", "thl0", "e120", "ekm", "qt0", "ql0", "ql0h")


def restart_name(ntrun: int, myidy: int, expnr: int, kind: str = "d", myidx: int = 0) -> str:
    return f"init{kind}{ntrun:08d}_{myidx:03d}_{myidy:03d}.{expnr:03d}"


def _put(f, payload: bytes):
    n = struct.pack("<i", len(payload))
    f.write(n); f.write(payload); f.write(n)


def _get(f) -> bytes:
    head = f.read(4)
    if len(head) != 4:
        raise EOFError("restart file: truncated record header")
    (n,) = struct.unpack("<i", head)
    payload = f.read(n)
    (m,) = struct.unpack("<i", f.read(4))
    if m != n or len(payload) != n:
        raise ValueError("restart file: record markers do not match (not a sequential unformatted file?)")
    return payload


def write_initd(path, nx, ny, nz, fields: dict, timee: float, dt: float, mindist=None, wall=None, fill=None):
    """fields: m-arrays for any of M_FIELDS; missing ones are filled with fill[name] (default 0; the neutral
    cold start has thl0 = the prof.inp value and e120 = e12min)."""
    fill = dict(fill or {})
    shp = (nz + 2, ny + 2, nx + 2)
    with open(path, "wb") as f:
        md = np.zeros((nz, ny, nx)) if mindist is None else np.ascontiguousarray(mindist, dtype="<f8")
        assert md.shape == (nz, ny, nx)
        _put(f, md.tobytes())
        wl = np.zeros((5, nz, ny, nx), dtype="<i4") if wall is None else np.ascontiguousarray(wall, dtype="<i4")
        assert wl.shape == (5, nz, ny, nx)
        _put(f, wl.tobytes())
        for name in M_FIELDS:
            a = fields.get(name)
            if a is None:
                a = np.full(shp, float(fill.get(name, 0.0)))
            a = np.asarray(a, dtype="<f8")
            assert a.shape == shp, (name, a.shape, shp)
            _put(f, np.ascontiguousarray(a[1:]).tobytes())          # k = kb .. ke+kh
        _put(f, struct.pack("<2d", timee, dt))


def read_initd(path, nx, ny, nz) -> dict:
    """Returns the ten m-arrays (ghost plane k = kb-1 zero-filled: the file does not hold it), mindist, wall,
    timee, dt."""
    out = {}
    with open(path, "rb") as f:
        out["mindist"] = np.frombuffer(_get(f), dtype="<f8").reshape(nz, ny, nx).copy()
        out["wall"] = np.frombuffer(_get(f), dtype="<i4").reshape(5, nz, ny, nx).copy()
        for name in M_FIELDS:
            a = np.zeros((nz + 2, ny + 2, nx + 2))
            a[1:] = np.frombuffer(_get(f), dtype="<f8").reshape(nz + 1, ny + 2, nx + 2)
            out[name] = a
        out["timee"], out["dt"] = struct.unpack("<2d", _get(f))
    return out


def write_inits(path, nx, ny, nz, sv0: list, timee: float):
    """sv0: list of nsv arrays, either m-arrays (halo 1) or c-arrays (halo 2, shape (nz+4, ny+4, nx+4))."""
    planes = []
    for a in sv0:
        a = np.asarray(a, dtype="<f8")
        if a.shape == (nz + 4, ny + 4, nx + 4):
            a = a[1:-1, 1:-1, 1:-1]
        assert a.shape == (nz + 2, ny + 2, nx + 2), a.shape
        planes.append(np.ascontiguousarray(a[1:]))
    with open(path, "wb") as f:
        _put(f, np.stack(planes).tobytes())
        _put(f, struct.pack("<d", timee))


def read_inits(path, nx, ny, nz, nsv) -> dict:
    with open(path, "rb") as f:
        raw = np.frombuffer(_get(f), dtype="<f8").reshape(nsv, nz + 1, ny + 2, nx + 2)
        (timee,) = struct.unpack("<d", _get(f))
    sv = []
    for n in range(nsv):
        a = np.zeros((nz + 2, ny + 2, nx + 2))
        a[1:] = raw[n]
        sv.append(a)
    return {"sv0": sv, "timee": timee}


# ---- device <-> restart files ---------------------------------------------------------------------------
def save_restart(core, directory, expnr, ntrun, timee, dt, rank=0, fill=None):
    """Download u0, v0, w0, pres0, ekm (+ scalars) of this rank's slab and write the reference's restart files.
    Call after RK stage 3 (the reference only writes then, src/modsave.f90:61)."""
    from . import lib as L
    g = core.g
    ny = core.nyl
    fields = {k: core.download(k) for k in _restart_fields(core)}
    path = os.path.join(directory, restart_name(ntrun, rank, expnr, "d"))
    write_initd(path, g.nx, ny, g.nz, fields, timee, dt, fill=fill)
    paths = [path]
    if core.nsv:
        sv = [core.download(L.scalar_field(L.SV0, n), halo=2) for n in range(core.nsv)]
        ps = os.path.join(directory, restart_name(ntrun, rank, expnr, "s"))
        write_inits(ps, g.nx, ny, g.nz, sv, timee)
        paths.append(ps)
    return paths


def _restart_fields(core):
    """The initd records this configuration transports (the rest are written as `fill` / ignored on read)."""
    names = ["u0", "v0", "w0", "pres0", "ekm"]
    if getattr(core, "ltempeq", False):
        names.append("thl0")
    if getattr(core, "loneeqn", False):
        names.append("e120")
    if getattr(core, "lmoist", False):
        names.append("qt0")
    return names


def find_layout(directory, expnr, ntrun, kind="d"):
    """(nprocx, nprocy) of the restart set init<kind><ntrun>_<myidx>_<myidy>.<expnr> in a directory, from its file names."""
    import glob
    import re
    pat = re.compile(rf"init{kind}{ntrun:08d}_(\d{{3}})_(\d{{3}})\.{expnr:03d}$")
    ids = [tuple(int(v) for v in pat.search(os.path.basename(f)).groups())
           for f in glob.glob(os.path.join(directory, f"init{kind}{ntrun:08d}_*_*.{expnr:03d}")) if pat.search(os.path.basename(f))]
    if not ids:
        raise FileNotFoundError(f"no init{kind}{ntrun:08d}_*_*.{expnr:03d} in {directory}")
    px, py = max(i for i, _ in ids) + 1, max(j for _, j in ids) + 1
    if len(set(ids)) != px * py:
        raise ValueError(f"restart set {ntrun} in {directory}: files of a {px} x {py} layout expected, {len(set(ids))} found")
    return px, py


def read_global(directory, expnr, ntrun, nx, ny, nz, nsv=0, rows=None, fields=None):
    """The whole domain from a restart set written on any pencil layout of the reference (nprocx x nprocy files, each holding
    its block with one ghost cell around it): -> (fields {name: m-array [nz+2, ny+2, nx+2]}, sv0 list of such arrays, timee, dt).
    Interior cells come from the rank that owns them; the outer ghost ring from the blocks on the domain's edge.
    rows = (r0, r1): only rows r0 <= j < r1 of the ghosted global array (a y-slab with its ghost rows) -- the arrays come back
    [nz+2, r1-r0, nx+2] and only the files whose rows overlap are opened; fields: only these names."""
    px, py = find_layout(directory, expnr, ntrun)
    if nx % px or ny % py:
        raise ValueError(f"restart set on {px} x {py} ranks does not divide a {nx} x {ny} domain")
    lx, ly = nx // px, ny // py
    r0, r1 = rows if rows is not None else (0, ny + 2)
    shape = (nz + 2, r1 - r0, nx + 2)
    out, sv, timee, dt = {}, [np.zeros(shape) for _ in range(nsv)], 0., 0.
    have_s = nsv > 0 and os.path.exists(os.path.join(directory, restart_name(ntrun, 0, expnr, "s")))
    for iy in range(py):
        # destination rows in the global array (with its ghost ring) and the matching rows of the block
        gj0, gj1 = iy * ly + (0 if iy == 0 else 1), (iy + 1) * ly + (2 if iy == py - 1 else 1)
        lo, hi = max(gj0, r0), min(gj1, r1)
        if lo >= hi:
            if rows is None or (timee or dt):
                continue
        for ix in range(px):
            d = read_initd(os.path.join(directory, restart_name(ntrun, iy, expnr, "d", myidx=ix)), lx, ly, nz)
            timee, dt = d["timee"], d["dt"]
            if lo >= hi:      # (only the clock was wanted from this file)
                break
            gi0, gi1 = ix * lx + (0 if ix == 0 else 1), (ix + 1) * lx + (2 if ix == px - 1 else 1)
            bi0, bj0 = gi0 - ix * lx, lo - iy * ly
            blocks = [(k, v) for k, v in d.items() if isinstance(v, np.ndarray) and v.ndim == 3 and v.shape == (nz + 2, ly + 2, lx + 2)
                      and (fields is None or k in fields)]
            for k, v in blocks:
                out.setdefault(k, np.zeros(shape))[:, lo - r0:hi - r0, gi0:gi1] = v[:, bj0:bj0 + hi - lo, bi0:bi0 + gi1 - gi0]
            del d
            if have_s:
                s = read_inits(os.path.join(directory, restart_name(ntrun, iy, expnr, "s", myidx=ix)), lx, ly, nz, nsv)
                for n, a in enumerate(s["sv0"]):
                    sv[n][:, lo - r0:hi - r0, gi0:gi1] = a[:, bj0:bj0 + hi - lo, bi0:bi0 + gi1 - gi0]
    return out, (sv if have_s else None), timee, dt


def load_restart_global(core, directory, expnr, ntrun, rank=0, nranks=1, read_scalars=True):
    """Warm start from a restart set of any CPU layout: this rank's y-slab of the assembled fields (see load_restart)."""
    from . import lib as L
    g = core.g
    nyl = core.nyl
    ny = nyl * nranks
    j0 = rank * nyl
    names = _restart_fields(core)
    # (only the files whose rows overlap this slab and its two ghost rows are read, and of those only the fields that are uploaded)
    d, sv, timee, dt = read_global(directory, expnr, ntrun, g.nx, ny, g.nz, core.nsv if read_scalars else 0, rows=(j0, j0 + nyl + 2), fields=set(names))
    cut = np.ascontiguousarray
    for k in names:
        core.upload(k, cut(d[k]))
    for k0 in names:
        if k0 not in ("pres0", "ekm"):
            core.upload(k0[:-1] + "m", cut(d[k0]))
    if core.nsv and sv is not None and read_scalars:
        for n, a in enumerate(sv):
            c = np.zeros((g.nz + 4, nyl + 4, g.nx + 4))
            c[1:-1, 1:-1, 1:-1] = cut(a)
            core.upload(L.scalar_field(L.SV0, n), c)
            core.upload(L.scalar_field(L.SVM, n), c)
    core.halos()
    if getattr(core, "open_x", False):      # inflow / outflow in x: the start-up's `boundary` with the restart's dt (DynCore.open_x_startup)
        core.open_x_startup(dt)
    core.boundary()
    return timee, dt


def load_restart(core, directory, expnr, ntrun, rank=0):
    """Warm start as readrestartfiles + readinitfiles do (src/modstartup.f90:1292-1340): u0.. from the file,
    um = u0 (the file is written after stage 3, when the reference itself has um = u0), ghosts re-derived by
    halos/boundary.  Returns (timee, dt)."""
    from . import lib as L
    g = core.g
    ny = core.nyl
    d = read_initd(os.path.join(directory, restart_name(ntrun, rank, expnr, "d")), g.nx, ny, g.nz)
    names = _restart_fields(core)
    for k in names:
        core.upload(k, d[k])
    for k0 in names:
        if k0 not in ("pres0", "ekm"):      # um = u0, thlm = thl0, ... (src/modstartup.f90:1675-1684)
            core.upload(k0[:-1] + "m", d[k0])
    if core.nsv:
        s = read_inits(os.path.join(directory, restart_name(ntrun, rank, expnr, "s")), g.nx, ny, g.nz, core.nsv)
        for n, a in enumerate(s["sv0"]):
            c = np.zeros((g.nz + 4, ny + 4, g.nx + 4))
            c[1:-1, 1:-1, 1:-1] = a
            core.upload(L.scalar_field(L.SV0, n), c)
            core.upload(L.scalar_field(L.SVM, n), c)
    core.halos()
    if getattr(core, "open_x", False):
        core.open_x_startup(d["dt"])
    core.boundary()
    return d["timee"], d["dt"]
