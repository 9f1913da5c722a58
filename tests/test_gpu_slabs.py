"""Multi-slab (multi-GPU) code path, tested on ONE GPU.

(1) UDC_FORCE_SLAB=1 drives a single rank through the distributed layout (1-D x transform, packed
    all-to-all blocks, transposed spectral array, packed ghost-row exchange) -> must match the golden
    reference dumps like the fast single-GPU path does.
(2) P = 2 and 4 virtual ranks in one process (one host thread per rank, device-to-device copies
    instead of RCCL, everything else identical) -> decomposition invariance against the P = 1 result,
    the same property the reference pins with its processor_boundaries test
    (tests/integration/processor_boundaries/test_processor_boundaries.py:28-34: 1e-9 / 2e-8).
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from common import RUN_CASES, deck_path, load_fixture, marr, nocorner, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switches", [
    {"UDC_FORCE_SLAB": "1"},                                # one rank through the slab layout: own line transforms, exchanges onto itself
    {"UDC_FORCE_SLAB": "1", "UDC_FFT_FUSED": "0"},          # ... with rocFFT + transpose kernels (rung 3 of bench.py's ladder)
    {"UDC_FORCE_SLAB": "1", "UDC_THOMAS_MIRROR_MIN": "16"},  # ... mirrored runs of a line solved together wherever a line has 16 rows
    {"UDC_FORCE_SLAB": "1", "UDC_DIV_IN_FFT": "0"},         # ... with the separate divergence kernel
    {"UDC_FORCE_SLAB": "1", "UDC_P_TRANSPOSE": "0"},        # ... p's ghost rows in an exchange of their own (default: inside the backward transpose)
    {"UDC_FORCE_SLAB": "1", "UDC_HALO_OVERLAP": "0", "UDC_MOM_PIPE": "0", "UDC_A2A_CHUNKS": "1"},      # rungs 1 and 2 of the ladder
    {"UDC_NO_FOLD": "1"},                                   # single slab: separate ghost-row kernels
    {"UDC_PTOTAL": "0"},                                    # ... pres0 and p kept apart as in the reference (default: the pressure-total form)
    {"UDC_SV_INLINE": "0"},                                 # ... every scalar integrated by tstep_integrate's kernel (default: a plain passive kappa scalar in its own sweep)
    {"UDC_NO_ALIAS": "1"},                                  # ... um always a real copy
    {"UDC_SCALAR_PAIR": "0"},                               # ... thl and qt swept one by one
    {"UDC_THOMAS": "0"},                                    # ... the streaming tridiagonal kernel
    {"UDC_OWN_FWD": "1", "UDC_NAT_REG": "0"},               # ... own forward half through the Stockham y pass
    {"UDC_FORCE_SLAB": "1", "UDC_SLAB_YREG": "0"},          # ... slab transforms through the Stockham y pass
], ids=lambda d: " ".join(f"{k[4:]}={v}" for k, v in d.items()))
def test_every_switch_setting_matches_reference(switches, monkeypatch):
    """Every run fixture (the reference's real program's restart files) through the fused substep under each of the library's order /
    variant switches (udc_create reads them): none may change a result beyond round-off."""
    from common import RUN_CASES, carr, deck_path, interior, load_fixture, marr, nocorner, relerr
    import udcore
    from udcore import read_deck, cold_start
    from udcore import lib as L
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    for name, iexp in RUN_CASES.items():
        fix = load_fixture(name)
        d = read_deck(deck_path(name, iexp))
        core = udcore.from_deck(d)
        core.load_state(cold_start(core.g, d, nsv=core.nsv, pre_boundary=True))
        core.start_up()
        dt = float(d.get("RUN", "dtmax"))
        dumps = sorted(int(k[1:4]) for k in fix if k.endswith(".u0") and k != "s000.u0")
        for isub in range(1, max(dumps) + 1):
            core.substep((isub - 1) % 3 + 1, dt, True)
            if isub in dumps:
                for k in ("u0", "v0", "w0", "pres0"):
                    ref = marr(fix, f"s{isub:03d}.{k}", core.g.nz)
                    e = relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]))
                    assert e <= 1e-9, (name, isub, k, e)
                for n in range(core.nsv):      # (the scalars' inflow / outflow of the BCxs deck lives here)
                    e = relerr(interior(core.download(L.scalar_field(L.SV0, n), halo=2), 2),
                               interior(carr(fix, f"s{isub:03d}.sv0_{n + 1:02d}", core.g.nz), 2))
                    assert e <= 1e-9, (name, isub, "sv0", n, e)
        core.close()


def test_rccl_transport_single_rank():
    """UDC_FORCE_COMM=1: a real RCCL communicator (one rank) carries the ghost rows, the two all-to-alls of the
    Poisson solve and the all-reduces of the forced slab path; torch is imported first, as in bench.py, so the
    library has to coexist with torch's own RCCL user.  Results must match the golden dumps."""
    code = r'''
import sys, ctypes, numpy as np
import torch
torch.cuda.set_device(0)
sys.path[:0] = ["%s/tests", "%s/u-dales_amd"]
from common import RUN_CASES, deck_path, load_fixture, marr, nocorner, relerr
import udcore
from udcore import read_deck, cold_start
name, iexp = sorted(RUN_CASES.items())[0]
fix = load_fixture(name)
d = read_deck(deck_path(name, iexp))
core = udcore.from_deck(d)
buf = (ctypes.c_ubyte * 128)()
assert core.lib.udc_comm_unique_id(buf) == 0
core.comm_init(bytes(buf))
core.load_state(cold_start(core.g, d, nsv=core.nsv))
dt = float(d.get("RUN", "dtmax"))
dumps = sorted(int(k[1:4]) for k in fix if k.endswith(".u0") and k != "s000.u0")
for isub in range(1, max(dumps) + 1):
    core.substep((isub - 1) %% 3 + 1, dt, True)
    if isub in dumps:
        for k in ("u0", "v0", "w0", "pres0"):
            ref = marr(fix, f"s{isub:03d}.{k}", core.g.nz)
            e = relerr(nocorner(core.download(k)[1:-1]), nocorner(ref[1:-1]))
            assert e <= 1e-9, (name, isub, k, e)
core.dt, core.rk3step = dt, 3
m = core.tstep_update(dt, ladaptive=True)
dv = core.divergence()
# what the communicator says about itself, and the exchange account with RCCL moving the buffers (bench.py's N > 1 fields)
info = core.comm_info()
assert info["transport"] == "rccl" and info["nranks"] == 1 and info["rank"] == 0 and info["version"].count(".") == 2, info
core.comm_stats(1)
for isub in range(3):
    core.substep(isub + 1, dt, True)
cs = core.comm_stats(2)
nch = info["transpose_k_chunks"]
assert cs["alltoall_ops"] == 3 * 2 * nch and cs["alltoall_ms"] > 0 and cs["ghost_row_exchanges"] >= 3 * 3 and cs["ghost_row_ms"] > 0, cs
plan = core.last_plan()
assert plan["slab_layout"] and plan["transpose_k_chunks"] == nch, plan
core.comm_dry_run(True)
core.substep(1, dt, True)      # (exchanges skipped: timing only, the state is discarded)
core.comm_dry_run(False)
core.sync()
core.close()
print("RCCL_OK", m, dv)
''' % (ROOT, ROOT)
    env = dict(os.environ, UDC_FORCE_SLAB="1", UDC_FORCE_COMM="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "RCCL_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_rccl_carries_the_overlapped_exchanges(tmp_path):
    """The ghost rows that travel beside the sweeps (closure edge / interior, project + integrate edge / interior: k_halo_y_begin /
    _join on the communication stream; the momentum sweep pipelined with the solve's k-chunks) through a real one-rank RCCL communicator, on a grid with enough tile rows for the splits
    (64 x 48 x 24: 6 and 12 tile rows), Vreman + floor wall function, nine substeps: the forced slab path with RCCL, with the overlap
    on and off, against the single-slab path (2-D rocFFT, folded ghost rows)."""
    code = r'''
import sys, ctypes, os, numpy as np
import torch
torch.cuda.set_device(0)
sys.path[:0] = ["%s/tests", "%s/u-dales_amd"]
from udcore.core import DynCore
from udcore.grid import Grid
nx, ny, nz = 64, 48, 24
g = Grid.uniform(nx, ny, nz)
core = DynCore(g, sgs=2, nsv=0, lbottom=True, z0=0.05)
if os.environ.get("UDC_FORCE_COMM") == "1":
    buf = (ctypes.c_ubyte * 128)()
    assert core.lib.udc_comm_unique_id(buf) == 0
    core.comm_init(bytes(buf))
core.set_forcing(np.full(nz, -1e-4), np.zeros(nz))
rng = np.random.default_rng(11)
for k, base in (("u0", 1.0), ("v0", 0.0), ("w0", 0.0)):
    a = np.zeros(g.mshape())
    a[1:-1, 1:-1, 1:-1] = base + 0.04 * (rng.random((nz, ny, nx)) - 0.5)
    if k == "w0":
        a[1] = 0.
    core.upload(k, a); core.upload(k.replace("0", "m"), a)
core.halos(); core.boundary()
core.run(9, 0.25)
out = {k: core.download(k) for k in ("u0", "v0", "w0", "pres0")}
out["div"] = core.divergence()[0]
core.close()
np.save(sys.argv[1], out, allow_pickle=True)
print("RUN_OK")
''' % (ROOT, ROOT)
    import numpy as np
    res = {}
    for tag, env in (("single", {}), ("rccl", {"UDC_FORCE_SLAB": "1", "UDC_FORCE_COMM": "1", "UDC_A2A_CHUNKS": "4"}),
                     ("rccl_inline", {"UDC_FORCE_SLAB": "1", "UDC_FORCE_COMM": "1", "UDC_A2A_CHUNKS": "4", "UDC_HALO_OVERLAP": "0", "UDC_MOM_PIPE": "0"})):
        out = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert "RUN_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        res[tag] = np.load(out, allow_pickle=True).item()
    # the overlapped order (exchanges beside the sweeps, the momentum sweep cut along the solve's four k-chunks) changes no arithmetic
    for k in ("u0", "v0", "w0", "pres0"):
        assert np.array_equal(res["rccl"][k], res["rccl_inline"][k]), k
    for tag in ("rccl", "rccl_inline"):
        assert res[tag]["div"] < 1e-10
        for k in ("u0", "v0", "w0", "pres0"):
            a, b = res[tag][k][1:-1], res["single"][k][1:-1]          # ghost rows included
            assert np.abs(a - b).max() <= 1e-10 * max(np.abs(b).max(), 1e-3), (tag, k, np.abs(a - b).max())


def ibm_block_lists(nx, ny, nz):
    """Solid / fluid-boundary point lists (u, v, w, c) of two blocks on the floor, the rule of tests/golden/make_golden.py."""
    c = np.zeros((nz + 2, ny, nx), dtype=bool)
    c[1:5, ny // 2 - 3:ny // 2 + 3, 4:10] = True          # straddles the middle of the domain in y (a slab boundary for even P)
    c[1:3, 2:5, nx - 9:nx - 4] = True
    u = c | np.roll(c, 1, axis=2)
    v = c | np.roll(c, 1, axis=1)
    w = c.copy(); w[1:] |= c[:-1]
    out = []
    for name, sol in (("u", u), ("v", v), ("w", w), ("c", c)):
        nb = np.zeros_like(sol)
        for ax, sh in ((2, 1), (2, -1), (1, 1), (1, -1)):
            nb |= np.roll(sol, sh, axis=ax)
        nb[1:] |= sol[:-1]; nb[:-1] |= sol[1:]
        bnd = nb & ~sol

        def pts(m, lo):
            m = m.copy(); m[:lo] = False; m[nz + 1:] = False
            kji = np.argwhere(m)
            return np.ascontiguousarray(np.stack([kji[:, 2] + 1, kji[:, 1] + 1, kji[:, 0]], axis=1), dtype=np.int32)
        out.append((pts(sol, 1), pts(bnd, 2 if name == "w" else 1)))
    return out


def synthetic_sections(g, grid, bnd):
    """Facet sections for the wall functions on every fluid-boundary point of a grid: one oblique facet; two thirds of the
    sections take the velocity at the cell, the others at a reconstruction point inside the neighbouring cells (udcore.facets'
    table layout)."""
    from udcore.facets import _findloc
    nx, ny, nz, dx, dy = g.nx, g.ny, g.nz, g.dx, g.dy
    xh, xf = np.arange(nx + 1) * dx, (np.arange(nx + 1) + 0.5) * dx
    yh, yf = np.arange(ny + 1) * dy, (np.arange(ny + 1) + 0.5) * dy
    zf, zh = g.zf[1:nz + 2], g.zh[1:nz + 2]
    xg, yg, zg = {"u": (xh, yf, zf), "v": (xf, yh, zf), "w": (xf, yf, zh)}[grid]
    n = len(bnd)
    S = {"n": n, "cell": np.asarray(bnd, dtype=np.int32).reshape(n, 3), "area": np.full(n, 0.1), "dist": np.full(n, 0.25),
         "fac": np.ones(n, dtype=np.int32), "comprec": np.ones(n, dtype=np.int32), "recpt": np.zeros((n, 3)),
         "recids": np.ones((n, 4, 3), dtype=np.int32)}
    for q, (i, j, k) in enumerate(S["cell"]):
        if q % 3 or k >= nz - 1:
            continue
        p = np.array([xg[i - 1] + 0.3 * dx, yg[j - 1] + 0.2 * dy, zg[k - 1] + 0.3 * (zf[1] - zf[0])])
        ids = np.array([[_findloc(p[0], a), _findloc(p[1], b), _findloc(p[2], c)] for a, b, c in ((xh, yf, zf), (xf, yh, zf), (xf, yf, zh), (xf, yf, zf))])
        if ids.min() < 1 or ids[:, 0].max() > nx or ids[:, 1].max() > ny or ids[:, 2].max() > nz:
            continue
        S["comprec"][q], S["recpt"][q], S["recids"][q] = 0, p, ids
    return S


def run_virtual(*args, **kw):
    """Virtual ranks need the in-process `local group` transport, which only libudcore_test.so has (the product library is built
    without it): the whole run, handles included, lives inside lib.test_transport()."""
    from udcore import lib as L
    with L.test_transport():
        return _run_virtual(*args, **kw)


def _run_virtual(P, g, deck_like, st_global, nsub, dt, sgs, nsv=0, extras=False):
    """Run nsub substeps on P virtual ranks; returns the stitched global u0, v0, w0, pres0."""
    from udcore.core import DynCore
    from udcore import lib as L
    lib = L.load()
    group = lib.udc_local_group_create(P) if P > 1 else 0
    nyl = g.ny // P
    cores, errs = [None] * P, []
    out = {}

    def worker(r):
        try:
            core = DynCore(g, sgs=sgs, nsv=nsv, rank=r, nranks=P, lbottom=extras, z0=0.04)
            cores[r] = core
            if extras:      # floor wall function + prescribed volume flow + buoyant temperature (all-reduced sums)
                core.set_masscorr(True, 1.03, True, 0.02)
                core.set_tempeq(bctopt=2, thl_top=291., wtsurf=0.03)
                if extras == 2:      # moist thermodynamics: three more all-reduced slab sums per call + thvh
                    core.set_moisture(bctopq=2, qt_top=0.0105, wqsurf=4e-5)
                    core.set_moist_thermo(288., 0.0105)
                core.set_buoyancy(True)
            if extras == 3:      # immersed boundary: two blocks, one across the slab boundaries, with temperature (adiabatic walls):
                # per-slab point lists, level sums with the solid points taken out all-reduced over the slabs
                for q, (sol, bnd) in enumerate(ibm_block_lists(g.nx, g.ny, g.nz)):
                    if q < 3:
                        core.set_ibm_points(q, sol, bnd)
                core.set_ibm_points(3, *ibm_block_lists(g.nx, g.ny, g.nz)[3])
                core.ibm_commit()
                # facet wall functions (iwallmom = 2) on every boundary point: every slab is handed all sections and keeps its
                # rows'; reconstruction cells reach into the neighbouring slab's ghost rows
                facets = {"norm": np.array([[0.6, 0., 0.8]]), "z0": np.array([0.01]), "z0h": np.array([0.001]), "tsurf": np.array([289.])}
                core.set_ibm_wallfun(2, 0.71, g.zf[1:g.nz + 2], g.zh[1:g.nz + 2])
                for q, gr in enumerate("uvw"):
                    S = synthetic_sections(g, gr, ibm_block_lists(g.nx, g.ny, g.nz)[q][1])
                    assert (S["comprec"] == 0).sum() > 10
                    core.set_ibm_sections(q, S, facets, np.ones((S["n"], 2)))
            if P > 1:
                core.comm_init_local(group)
            local = {}
            for k, a in st_global.items():
                h = (a.shape[1] - g.ny) // 2
                local[k] = np.ascontiguousarray(a[:, r * nyl:r * nyl + nyl + 2 * h, :])
            core.load_state(local)
            core.halos()
            core.boundary()
            td = None
            if extras == 3:      # xytdump over the slabs: per-slab mask bits, level sums all-reduced, one sample per step
                from udcore.stats import TDump
                core.ltempeq = True
                td = TDump(core, tsample=dt, tstatsdump=1e9, xyt=True, yt=True, xy=True, y=True, ibm_lists=dict(zip("uvwc", ibm_block_lists(g.nx, g.ny, g.nz))),
                           jtot=g.ny, j0=r * nyl, nyl=nyl)
            for isub in range(nsub):
                core.substep(isub % 3 + 1, dt, bool(extras))
                if td is not None:
                    td.step(isub % 3 + 1, dt, dt * (isub // 3 + 1))
            if td is not None:
                out.setdefault(r, {})["xyt"] = td.xyt()
                out[r]["xyt"].update({"yt." + k: v for k, v in td.yt().items()})     # ytdump's y-averages: same table on every rank
                out[r]["xyt"].update({"xy." + k: v for k, v in td.xy().items()})     # xydump / ydump: the last sample's own tables
                out[r]["xyt"].update({"y." + k: v for k, v in td.y().items()})
            out.setdefault(r, {}).update({k: core.download(k) for k in ("u0", "v0", "w0", "pres0") + (("thl0",) if extras else ()) + (("qt0",) if extras == 2 else ())})
            out[r]["div"] = core.divergence()
        except Exception as e:   # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    res = {}
    for k in ("u0", "v0", "w0", "pres0") + (("thl0",) if extras else ()) + (("qt0",) if extras == 2 else ()):
        res[k] = np.concatenate([out[r][k][:, 1:-1, :] for r in range(P)], axis=1)
    res["div"] = out[0]["div"]
    if "xyt" in out[0]:
        for r in range(1, P):      # the table is the same on every rank
            for k, v in out[0]["xyt"].items():
                assert np.array_equal(v, out[r]["xyt"][k]), (r, k)
        res["xyt"] = out[0]["xyt"]
    for c in cores:
        c.close()
    return res


@pytest.mark.parametrize("shape,sgs,chunks,extras", [((32, 16, 12), 2, 1, False), ((24, 32, 10), 1, 2, False),
                                                     ((32, 16, 12), 2, 3, False), ((32, 16, 12), 2, 1, 1),
                                                     ((32, 16, 12), 2, 2, 2),
                                                     ((128, 64, 16), 2, 2, False), ((64, 32, 8), 1, 4, 1),
                                                     # slabs of 64 / 32 rows on 2 / 4 ranks: the momentum sweep pipelined with four k-chunks
                                                     # (vp's ghost row handed on chunk by chunk), p's ghost rows inside the backward transpose
                                                     ((64, 128, 16), 2, 4, False),
                                                     # rows of 512 cells carry a line of padding (Geo.sy = nx + 16): halo packs, own line
                                                     # FFTs and the exchange buffers across slabs; 272 levels: the wave-specialised Thomas
                                                     # kernel (two-workgroup occupancy) on the slabs' share of the modes
                                                     ((512, 32, 8), 2, 2, False), ((32, 32, 272), 2, 1, False),
                                                     ((32, 32, 12), 2, 1, 3)])      # immersed boundary + temperature
def test_decomposition_invariance(shape, sgs, chunks, extras, monkeypatch):
    # chunks > 1: the k-chunked all-to-all pipeline (exchange on a second stream, overlapped with rocFFT)
    # power-of-two nx and ny: the slab ranks run the own line FFTs with fused packing (udc_fft.hip) against the single
    # rank's 2-D rocFFT; (24, 32, 10) keeps the rocFFT + transpose kernels of the slab path covered
    monkeypatch.setenv("UDC_A2A_CHUNKS", str(chunks))
    from test_gpu_parity import random_state
    from udcore.grid import Grid
    nx, ny, nz = shape
    g = Grid.uniform(nx, ny, nz)
    st = random_state(g, seed=42)
    if extras:
        rng = np.random.default_rng(5)
        t = np.zeros(g.mshape())
        t[1:-1, 1:-1, 1:-1] = 288. + 0.2 * g.zf[1:nz + 1, None, None] + 0.05 * rng.standard_normal((nz, ny, nx))
        t[:, 0, :] = t[:, ny, :]; t[:, ny + 1, :] = t[:, 1, :]
        t[:, :, 0] = t[:, :, nx]; t[:, :, nx + 1] = t[:, :, 1]
        t[0] = t[1]; t[nz + 1] = 2 * 291. - t[nz]
        st["thl0"], st["thlm"] = t, t.copy()
    if extras == 2:
        q = np.zeros(g.mshape())
        q[1:-1, 1:-1, 1:-1] = 0.0118 - 8e-5 * g.zf[1:nz + 1, None, None] + 2e-4 * rng.standard_normal((nz, ny, nx))
        q[:, 0, :] = q[:, ny, :]; q[:, ny + 1, :] = q[:, 1, :]
        q[:, :, 0] = q[:, :, nx]; q[:, :, nx + 1] = q[:, :, 1]
        q[nz + 1] = 2 * 0.0105 - q[nz]
        st["qt0"], st["qtm"] = q, q.copy()
    ref = run_virtual(1, g, None, st, 6, 0.05, sgs, extras=extras)
    assert ref["div"][0] < 1e-11
    for P in (2, 4) + ((8,) if ny % 16 == 0 and ny >= 32 else ()):      # 8 slabs = the driver's full node
        got = run_virtual(P, g, None, st, 6, 0.05, sgs, extras=extras)
        for k in ("u0", "v0", "w0", "pres0") + (("thl0",) if extras else ()) + (("qt0",) if extras == 2 else ()):
            e = relerr(got[k][1:-1], ref[k][1:-1], 1.0 if k == "thl0" else None)
            assert e <= 1e-10, (P, k, e)
        assert abs(got["div"][0] - ref["div"][0]) < 1e-12      # all-reduced max agrees on every rank
        if "xyt" in ref:
            for k, v in ref["xyt"].items():
                fl = v != -999.                 # ytdump's marker of columns without fluid points: same places, exactly
                assert np.array_equal(fl, got["xyt"][k] != -999.), (P, k)
                sc = max(np.abs(v[fl]).max(), 1e-1 if "thlpthlp" in k or "wpthlp" in k else 1e-3)
                assert np.abs(got["xyt"][k] - v)[fl].max() <= 1e-10 * sc, (P, k)
            assert np.abs(ref["xyt"]["yt.upupyt"]).max() > 1e-4
            assert np.abs(ref["xyt"]["uxyt"]).max() > 0.1 and ref["xyt"]["tketxyc"].min() > 0.
