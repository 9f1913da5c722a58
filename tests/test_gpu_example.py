"""A deck of the reference's own examples, unmodified, through the device runner (`python -m udcore.run`): examples/999, the
flat neutral channel at 128^3 -- floor wall function at the reference's defaults (BCbotm = 2 with the temperature equation
off and thls at its default of -1), adaptive time step, &OUTPUT tdump + xytdump + fielddump, a CPU layout of 4 x 2 ranks in
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
           "upvptxyij": "upvpxyt"}


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
    STL), read by udcore/facets.py.  One line is added to the shipped deck (iwallmom = 3: its default of 2 needs a facet
    temperature file the example does not ship).  25 adaptive steps against the reference binary on the same files."""
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
    slab averages; as for 001 the deck gets `iwallmom = 3`.  25 adaptive steps against the reference binary."""
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
    # the floor at the reference's defaults is a strong sink: the first level has lost nearly all its momentum
    assert got["xyt"]["uxyt"][0] < 0.05 and got["xyt"]["uxyt"][5] > 0.9
    # output files of the deck's &OUTPUT
    files = os.listdir(tmp_path)
    assert "tdump.999.npz" in files and "xytdump.999.npz" in files and "fielddump.000.999.npz" in files
    fd = np.load(tmp_path / "fielddump.000.999.npz")
    assert fd["u0"].shape[1:] == (128, 128, 128) and len(fd["time"]) >= 3 and fd["u0"].dtype == np.float32
    xd = np.load(tmp_path / "xytdump.999.npz")
    assert xd["uxyt"].shape[1] == 128 and len(xd["time"]) >= 1
