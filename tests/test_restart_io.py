"""Restart files in the reference's layout (src/modsave.f90:80-121): the fixture under
tests/golden/restart_8x8x8/ was written by the reference's own writerestartfiles (oracle/_ref/udales_ref in
`restart` mode, see tests/golden/make_golden.py); the same run's state was dumped next to it."""
import gzip
import os

import numpy as np
import pytest

from common import GOLDEN, carr, marr
from refdump import read_dump
from udcore import restart as R

NAME, IEXP, N = "restart_8x8x8", 31, 8
DIR = os.path.join(GOLDEN, NAME)


def _unzip(fn, tmp_path):
    out = os.path.join(tmp_path, fn)
    with gzip.open(os.path.join(DIR, fn + ".gz"), "rb") as f, open(out, "wb") as g:
        g.write(f.read())
    return out


def _state():
    with gzip.open(os.path.join(DIR, "state.bin.gz"), "rb") as f:
        return read_dump(f.read())


def test_file_names():
    assert R.restart_name(2, 0, 31) == "initd00000002_000_000.031"
    assert R.restart_name(1234, 5, 7, "s") == "inits00001234_000_005.007"


def test_read_reference_restart(tmp_path):
    st = _state()
    timee, dt, ntrun = st["rsttime"].data
    path = _unzip(R.restart_name(int(ntrun), 0, IEXP), tmp_path)
    d = R.read_initd(path, N, N, N)
    assert d["timee"] == timee and d["dt"] == dt
    for k in ("u0", "v0", "w0", "pres0", "ekm"):
        np.testing.assert_array_equal(d[k][1:], marr(st, "rst." + k, N)[1:], err_msg=k)
    assert np.all(d["thl0"][1:] == 288.0) and np.all(d["qt0"] == 0.0)
    s = R.read_inits(_unzip(R.restart_name(int(ntrun), 0, IEXP, "s"), tmp_path), N, N, N, 2)
    assert s["timee"] == timee
    for n in range(2):
        ref = carr(st, f"rst.sv0_{n + 1:02d}", N)[1:-1, 1:-1, 1:-1]
        np.testing.assert_array_equal(s["sv0"][n][1:], ref[1:])


@pytest.mark.parametrize("px,py", [(2, 2), (1, 4), (4, 1)])
def test_restart_set_of_any_pencil_layout_is_assembled(px, py, tmp_path):
    """A run of the reference on nprocx x nprocy ranks leaves one file per pencil, each with its block and a ghost cell around it
    (src/modsave.f90:80-121: ib-ih:ie+ih, jb-jh:je+jh).  The reference-written single-rank file, cut into such blocks (ghost cells
    from the neighbouring block, as the reference's halo exchange leaves them), comes back as the whole domain."""
    st = _state()
    timee, dt, ntrun = st["rsttime"].data
    ntrun = int(ntrun)
    d = R.read_initd(_unzip(R.restart_name(ntrun, 0, IEXP), tmp_path), N, N, N)
    s = R.read_inits(_unzip(R.restart_name(ntrun, 0, IEXP, "s"), tmp_path), N, N, N, 2)
    out = tmp_path / "split"
    out.mkdir()
    lx, ly = N // px, N // py
    for ix in range(px):
        for iy in range(py):
            cut = lambda a: np.ascontiguousarray(a[:, iy * ly:iy * ly + ly + 2, ix * lx:ix * lx + lx + 2])      # noqa: E731
            R.write_initd(str(out / R.restart_name(ntrun, iy, IEXP, "d", myidx=ix)), lx, ly, N, {k: cut(d[k]) for k in R.M_FIELDS}, timee, dt)
            R.write_inits(str(out / R.restart_name(ntrun, iy, IEXP, "s", myidx=ix)), lx, ly, N, [cut(a) for a in s["sv0"]], timee)
    assert R.find_layout(str(out), IEXP, ntrun) == (px, py)
    f, sv, t2, dt2 = R.read_global(str(out), IEXP, ntrun, N, N, N, nsv=2)
    assert (t2, dt2) == (timee, dt)
    for k in R.M_FIELDS:
        np.testing.assert_array_equal(f[k][1:], d[k][1:], err_msg=k)
    for n in range(2):
        np.testing.assert_array_equal(sv[n][1:], s["sv0"][n][1:])
    # a y-slab of it (what one rank of a multi-GPU run reads: its rows + two ghost rows, only the files that hold them, only the
    # fields it uploads)
    for j0, nyl in ((0, N // 2), (N // 2, N // 2), (2, 3)):
        fs, svs, t3, dt3 = R.read_global(str(out), IEXP, ntrun, N, N, N, nsv=2, rows=(j0, j0 + nyl + 2), fields={"u0", "pres0"})
        assert (t3, dt3) == (timee, dt) and set(fs) == {"u0", "pres0"}
        for k in fs:
            np.testing.assert_array_equal(fs[k], f[k][:, j0:j0 + nyl + 2, :], err_msg=k)
        for n in range(2):
            np.testing.assert_array_equal(svs[n], sv[n][:, j0:j0 + nyl + 2, :])


def test_write_is_byte_identical(tmp_path):
    """What read_initd returns, written back, is the reference's file byte for byte."""
    st = _state()
    ntrun = int(st["rsttime"].data[2])
    src = _unzip(R.restart_name(ntrun, 0, IEXP), tmp_path)
    d = R.read_initd(src, N, N, N)
    out = os.path.join(tmp_path, "copy.bin")
    R.write_initd(out, N, N, N, {k: d[k] for k in R.M_FIELDS}, d["timee"], d["dt"], d["mindist"], d["wall"])
    assert open(out, "rb").read() == open(src, "rb").read()
    srcs = _unzip(R.restart_name(ntrun, 0, IEXP, "s"), tmp_path)
    s = R.read_inits(srcs, N, N, N, 2)
    outs = os.path.join(tmp_path, "copys.bin")
    R.write_inits(outs, N, N, N, s["sv0"], s["timee"])
    assert open(outs, "rb").read() == open(srcs, "rb").read()
    # from the dumped state alone (thl0 / e120 via fill values): identical too
    fields = {k: marr(st, "rst." + k, N) for k in ("u0", "v0", "w0", "pres0", "ekm")}
    out2 = os.path.join(tmp_path, "fromstate.bin")
    R.write_initd(out2, N, N, N, fields, d["timee"], d["dt"], fill={"thl0": 288.0, "e120": float(d["e120"][1, 1, 1])})
    a, b = R.read_initd(out2, N, N, N), d
    for k in R.M_FIELDS:
        np.testing.assert_array_equal(a[k][1:], b[k][1:], err_msg=k)


def test_truncated_file_is_rejected(tmp_path):
    src = _unzip(R.restart_name(2, 0, IEXP), tmp_path)
    bad = os.path.join(tmp_path, "bad.bin")
    with open(src, "rb") as f, open(bad, "wb") as g:
        g.write(f.read()[:5000])
    with pytest.raises((EOFError, ValueError, Exception)):
        R.read_initd(bad, N, N, N)


@pytest.mark.gpu
def test_runner_warm_starts_from_the_deck(tmp_path):
    """&RUN lwarmstart / startfile as the reference reads them (src/modstartup.f90:784, 2194-2229), the restart set cut into the
    files of a 2 x 2 CPU layout: the runner picks ntrun from the file name, assembles the set, reads the scalars under lreadscal,
    and continues from the file's clock with the file's fields."""
    import shutil
    from common import nocorner, relerr
    from udcore import lib as L
    from udcore import run
    st = _state()
    timee, dt, ntrun = st["rsttime"].data
    ntrun = int(ntrun)
    d = R.read_initd(_unzip(R.restart_name(ntrun, 0, IEXP), tmp_path), N, N, N)
    s = R.read_inits(_unzip(R.restart_name(ntrun, 0, IEXP, "s"), tmp_path), N, N, N, 2)
    work = tmp_path / "case"
    work.mkdir()
    for fn in os.listdir(os.path.join(GOLDEN, "cases", NAME)):
        shutil.copy(os.path.join(GOLDEN, "cases", NAME, fn), work)
    for ix in range(2):
        for iy in range(2):
            cut = lambda a: np.ascontiguousarray(a[:, iy * 4:iy * 4 + 6, ix * 4:ix * 4 + 6])      # noqa: E731
            R.write_initd(str(work / R.restart_name(ntrun, iy, IEXP, "d", myidx=ix)), 4, 4, N, {k: cut(d[k]) for k in R.M_FIELDS}, timee, dt)
            R.write_inits(str(work / R.restart_name(ntrun, iy, IEXP, "s", myidx=ix)), 4, 4, N, [cut(a) for a in s["sv0"]], timee)
    deck = work / f"namoptions.{IEXP:03d}"
    text = deck.read_text()
    text = text.replace("&RUN\n", f"&RUN\nlwarmstart = .true.\nstartfile = 'initd{ntrun:08d}_xxx_xxx.{IEXP:03d}'\n", 1)
    text = text.replace("&SCALARS\n", "&SCALARS\nlreadscal = .true.\n", 1)
    deck.write_text(text)
    got = {}

    def at_end(core, tdump):
        got["time"] = core.timee
        got["u0"] = core.download("u0")
        got["sv"] = core.download(L.scalar_field(L.SV0, 0), halo=2)

    # zero steps: the state the time loop would start from
    assert run.main([str(deck), "--steps", "1", "--quiet"], at_end=at_end) == 0
    assert abs(got["time"] - (timee + dt)) < 1e-12      # one step on from the file's clock
    # the same through the single-file reader, one step
    import udcore
    from udcore import read_deck
    core = udcore.from_deck(read_deck(os.path.join(GOLDEN, "cases", NAME, f"namoptions.{IEXP:03d}")))
    for kind in "ds":
        _unzip(R.restart_name(ntrun, 0, IEXP, kind), tmp_path)
    t0, dt0 = R.load_restart(core, str(tmp_path), IEXP, ntrun)
    for rk in (1, 2, 3):
        core.substep(rk, dt0, True)
    assert relerr(nocorner(got["u0"][1:-1]), nocorner(core.download("u0")[1:-1])) <= 1e-12
    assert relerr(got["sv"][2:-2, 2:-2, 2:-2], core.download(L.scalar_field(L.SV0, 0), halo=2)[2:-2, 2:-2, 2:-2]) <= 1e-12
    core.close()


@pytest.mark.gpu
def test_device_warm_start_and_hand_back(tmp_path):
    """Device run warm-started from the reference's restart file continues like the reference does, and the
    restart file the device writes after those substeps matches the reference's own fields."""
    import udcore
    from common import deck_path, nocorner, relerr
    from udcore import read_deck
    st = _state()
    ntrun = int(st["rsttime"].data[2])
    for kind in "ds":
        _unzip(R.restart_name(ntrun, 0, IEXP, kind), tmp_path)
    d = read_deck(deck_path(NAME, IEXP))
    core = udcore.from_deck(d)
    timee, dt = R.load_restart(core, str(tmp_path), IEXP, ntrun)
    assert dt == float(d.get("RUN", "dtmax")) and timee == st["rsttime"].data[0]
    # the uploaded state equals the dumped one (ghosts re-derived by halos/boundary)
    for k in ("u0", "v0", "w0", "pres0"):
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(marr(st, "rst." + k, N)[1:-1])) <= 1e-14, k
    # hand back: what the device writes is the reference's file content for the dynamic fields
    os.makedirs(tmp_path / "out", exist_ok=True)
    paths = R.save_restart(core, str(tmp_path / "out"), IEXP, ntrun, timee, dt, fill={"thl0": 288.0})
    a = R.read_initd(paths[0], N, N, N)
    b = R.read_initd(os.path.join(tmp_path, R.restart_name(ntrun, 0, IEXP)), N, N, N)
    for k in ("u0", "v0", "w0", "pres0", "ekm", "thl0"):
        assert relerr(nocorner(a[k][1:-1]), nocorner(b[k][1:-1])) <= 1e-14, k
    # and the run continues where the reference's uninterrupted run goes (substeps 7-9 of the same deck)
    for rk in (1, 2, 3):
        core.substep(rk, dt, True)
    for k in ("u0", "v0", "w0", "pres0"):
        assert relerr(nocorner(core.download(k)[1:-1]), nocorner(marr(st, "s009." + k, N)[1:-1])) <= 1e-9, k
    # scalars: the restart format does not hold the ghost planes below the floor that the kappa limiter reads
    # at k = kb (the reference's own warm start leaves them zero, src/modstartup.f90:1609-1682, while its
    # uninterrupted run keeps the cold-start values), so only the levels the floor cannot have reached yet match
    from udcore import lib as L
    for n in range(2):
        got = core.download(L.scalar_field(L.SV0, n), halo=2)
        ref = carr(st, f"s009.sv0_{n + 1:02d}", N)
        assert relerr(got[2 + 5:-2, 2:-2, 2:-2], ref[2 + 5:-2, 2:-2, 2:-2]) <= 1e-9
        assert relerr(got[2:-2, 2:-2, 2:-2], ref[2:-2, 2:-2, 2:-2]) <= 1e-3
    divmax, _ = core.divergence()
    assert divmax < 1e-11
    core.close()


@pytest.mark.gpu
def test_restart_round_trip_with_temperature_and_moisture(tmp_path):
    """thl0 and qt0 travel through the initd file: a device run stopped after RK stage 3, written out, read into a fresh
    handle and continued arrives where the uninterrupted device run does (moist thermodynamics on)."""
    import udcore
    from common import RUN_CASES, deck_path, nocorner, relerr
    from udcore import read_deck, cold_start
    name = "run_moist_16x8x12s"
    d = read_deck(deck_path(name, RUN_CASES[name]))
    dt = float(d.get("RUN", "dtmax"))
    a = udcore.from_deck(d)
    a.load_state(cold_start(a.g, d))
    for isub in range(9):
        a.substep(isub % 3 + 1, dt, True)
        if isub == 5:
            paths = R.save_restart(a, str(tmp_path), 36, 2, 2 * dt, dt)
            th = a.thermo_state()
    rec = R.read_initd(paths[0], a.g.nx, a.g.ny, a.g.nz)
    assert rec["thl0"][1:-1].min() > 280. and 0.005 < rec["qt0"][1:-1].max() < 0.02
    b = udcore.from_deck(d)
    R.load_restart(b, str(tmp_path), 36, 2)
    b.thermo_state(th)          # presf/exnf of the previous call are not in the reference's restart format
    for isub in range(6, 9):
        b.substep(isub % 3 + 1, dt, True)
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0"):
        sc = 1.0 if k == "thl0" else None
        assert relerr(nocorner(b.download(k)[1:-1]), nocorner(a.download(k)[1:-1]), sc) <= 1e-9, k
    a.close(); b.close()
