"""A deck of the reference's own examples, unmodified, through the device runner (`python -m udcore.run`): examples/999, the
flat neutral channel at 128^3 -- floor wall function at the reference's defaults (BCbotm = 2 and iwallmom = 2 with the temperature
equation off, which checkinitvalues turns into the neutral function), adaptive time step, &OUTPUT tdump + xytdump + fielddump, a CPU layout of 4 x 2 ranks in
&RUN, a pre-processing group (&INP) the Fortran never reads.  Golden: xytdump's table and the clock after 25 steps of the
reference binary on the same files (tests/golden/make_golden.py, EXAMPLES)."""
import gzip
import os
import shutil
import sys

import numpy as np
import pytest

from common import GOLDEN, load_fixture

pytestmark = pytest.mark.gpu
XYT_FIX = {"uwtxyik": "uwxyt", "vwtxyjk": "vwxyt", "wwtxyk": "wwxyt", "uvtxyij": "uvxyt", "upwptxyik": "upwpxyt", "vpwptxyjk": "vpwpxyt",
           "upvptxyij": "upvpxyt", "wthltxyk": "wthlxyt", "wpthlptxyk": "wpthlpxyt"}


def _unpack(case, tmp_path):
    src = os.path.join(GOLDEN, "cases", case)
    for fn in os.listdir(src):
        if fn.endswith(".gz"):
            with gzip.open(os.path.join(src, fn), "rb") as f, open(tmp_path / fn[:-3], "wb") as o:
                o.write(f.read())
        else:
            shutil.copy(os.path.join(src, fn), tmp_path)


def test_reference_example_001_with_the_ground_as_facets(tmp_path):
    """examples/001 of the reference: the same 128^3 channel with the ground as an immersed boundary -- the lists and the ~53000
    facet sections are the output of the reference's own pre-processing (multiple sections per boundary point, facets from an
    STL), read by udcore/facets.py.  The deck as shipped: its iwallmom stays at the default of 2, which checkinitvalues turns into
    the neutral wall function because the temperature equation is off.  25 adaptive steps against the reference on the same files."""
    from udcore import run
    fix = load_fixture("example_001")
    _unpack("example_001", tmp_path)
    got = {}

    def at_end(core, tdump):
        got["xyt"], got["time"], got["div"] = tdump.xyt(), (core.timee, core.dt), core.divergence()[0]

    assert run.main([str(tmp_path / "namoptions.001"), "--steps", "25", "--quiet"], at_end=at_end) == 0
    tref, dtref = fix["end.time"].data
    assert abs(got["time"][0] - tref) <= 1e-9 * tref and abs(got["time"][1] - dtref) <= 1e-8 * dtref
    assert got["div"] < 1e-10
    for k, rec in fix.items():
        if k.startswith("xyt."):
            ref = rec.data[:128]
            g = got["xyt"][XYT_FIX.get(k[4:], k[4:])]
            assert np.abs(g - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3), (k, np.abs(g - ref).max())
    assert 0.2 < got["xyt"]["uxyt"][0] < 0.5 and got["xyt"]["uxyt"][5] > 0.9      # the wall functions slowed the first level down


def test_reference_example_002_cube_array_with_facets(tmp_path):
    """examples/002: 64^3, a regular array of cubes from an STL -- 1024 facets, ~8500 boundary points per grid and ~10800 facet
    sections per velocity grid from the reference's pre-processing, the deck's CPU layout of 2 x 2 ranks, xytdump with masked
    slab averages; the deck as shipped (neutral wall function by checkinitvalues' rule, as for 001).  25 adaptive steps against the
    reference."""
    from udcore import run
    fix = load_fixture("example_002")
    _unpack("example_002", tmp_path)
    got = {}

    def at_end(core, tdump):
        got["xyt"], got["time"], got["div"] = tdump.xyt(), (core.timee, core.dt), core.divergence()[0]

    assert run.main([str(tmp_path / "namoptions.002"), "--steps", "25", "--quiet"], at_end=at_end) == 0
    tref, dtref = fix["end.time"].data
    assert abs(got["time"][0] - tref) <= 1e-9 * tref and abs(got["time"][1] - dtref) <= 1e-8 * dtref
    assert got["div"] < 1e-10
    for k, rec in fix.items():
        if k.startswith("xyt."):
            ref = rec.data[:64]
            g = got["xyt"][XYT_FIX.get(k[4:], k[4:])]
            assert np.abs(g - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3 * 2.25), (k, np.abs(g - ref).max())


def test_reference_example_101_runs_unmodified(tmp_path):
    """examples/101, the deck as shipped: street canyons with heated walls (320 facets) at 64^3 -- temperature with buoyancy, wall
    functions with the stability functions on the facet temperatures (iwallmom = 2), the heat wall function (iwalltemp = 2), a
    prescribed volume flow, a scalar line source, the scalar entering clean and leaving through a convective outlet (BCxs = 2),
    xytdump over the fluid cells.  25 adaptive steps against the reference binary on the same files: the statistics table,
    the clock, and the scalar's mean per level and along x."""
    from udcore import lib as L
    from udcore import run
    fix = load_fixture("example_101")
    _unpack("example_101", tmp_path)
    got = {}

    def at_end(core, tdump):
        got["xyt"], got["time"], got["div"] = tdump.xyt(), (core.timee, core.dt), core.divergence()[0]
        got["sv"] = core.download(L.scalar_field(L.SV0, 0), halo=2)[2:-2, 2:-2, 2:-2]

    # (--ibm-mask-wrap none: the golden is the reference's single-rank run; with the deck's 2 x 2 layout the obstacle cells on the
    # edge of the domain would average over other neighbours -- nothing the fluid feels, but xytdump's first-level thlsgs sums
    # over every cell, solid ones included, src/modmpi.f90:646-649)
    assert run.main([str(tmp_path / "namoptions.101"), "--steps", "25", "--quiet", "--ibm-mask-wrap", "none"], at_end=at_end) == 0
    tref, dtref = fix["end.time"].data
    assert abs(got["time"][0] - tref) <= 1e-9 * tref and abs(got["time"][1] - dtref) <= 1e-8 * dtref
    assert got["div"] < 1e-10
    for k, rec in fix.items():
        if k.startswith("xyt."):
            ref = rec.data[:64]
            g = got["xyt"][XYT_FIX.get(k[4:], k[4:])]
            sc = max(np.abs(ref).max(), 1e-3 * 2.25) if "thl" not in k else max(np.abs(ref).max(), 1e-3)
            lo = 0
            # (xyt.thlsgsxyt: level kb of a w-masked average has no fluid point and becomes the sum over EVERY cell (avexy_ibm's rule),
            #  so the values `solid` parks inside the obstacles enter.  For an obstacle cell on the domain's edge the reference averages
            #  ghost cells: the m-field's ghost row as the last `halos` left it, the tendency's, which nobody fills -- reproduced since
            #  round 3 (src/modibm.f90:748-826), so this number needs no allowance any more)
            assert np.abs(g[lo:] - ref[lo:]).max() <= 2e-9 * sc, (k, np.abs(g - ref).max())
    alongx = got["sv"].mean(axis=(0, 1))
    assert np.abs(alongx - fix["end.sv1x"].data).max() <= 1e-9 * np.abs(fix["end.sv1x"].data).max()
    # mean over the fluid cells of a level: the plume sits in the first levels, between the canyon walls
    ref = fix["end.sv1xy"].data
    assert ref.max() > 0.1 and got["sv"][0].max() > ref[0]


def test_reference_example_999_runs_unmodified(tmp_path):
    from udcore import run
    fix = load_fixture("example_999")
    _unpack("example_999", tmp_path)
    got = {}

    def at_end(core, tdump):
        got["xyt"] = tdump.xyt()
        got["time"] = (core.timee, core.dt)
        got["div"] = core.divergence()[0]
        got["nsamples"] = tdump.nsamples

    assert run.main([str(tmp_path / "namoptions.999"), "--steps", "25", "--quiet"], at_end=at_end) == 0
    # the clock: the adaptive time step followed the reference's through 75 substeps
    tref, dtref = fix["end.time"].data
    assert abs(got["time"][0] - tref) <= 1e-9 * tref and abs(got["time"][1] - dtref) <= 1e-8 * dtref
    assert got["div"] < 1e-10 and got["nsamples"] >= 10
    nz = 128
    u2 = 1.0
    for k, rec in fix.items():
        if not k.startswith("xyt."):
            continue
        ref = rec.data[:nz]
        g = got["xyt"][XYT_FIX.get(k[4:], k[4:])]
        # profiles of O(1e-7 .. 1) quantities of a flow with |u| = 1: absolute agreement on the scale of u^2
        assert np.abs(g - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3 * u2), (k, np.abs(g - ref).max())
    # the neutral floor (what checkinitvalues makes of the deck's defaults, src/modstartup.f90:811-816) slowed the first level down
    assert 0.2 < got["xyt"]["uxyt"][0] < 0.5 and got["xyt"]["uxyt"][5] > 0.9
    # output files of the deck's &OUTPUT
    files = os.listdir(tmp_path)
    assert "tdump.999.npz" in files and "xytdump.999.npz" in files and "fielddump.000.999.npz" in files
    fd = np.load(tmp_path / "fielddump.000.999.npz")
    assert fd["u0"].shape[1:] == (128, 128, 128) and len(fd["time"]) >= 3 and fd["u0"].dtype == np.float32
    xd = np.load(tmp_path / "xytdump.999.npz")
    assert xd["uxyt"].shape[1] == 128 and len(xd["time"]) >= 1
