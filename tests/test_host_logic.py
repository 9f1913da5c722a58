"""CPU tests of the host-side logic and of the C-ABI library's symbol table (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from common import RUN_CASES, KERNEL_CASES, carr, deck_path, load_fixture, marr
from udcore import read_deck, sgs_from_deck, cold_start
from udcore.grid import Grid, lcg_noise
from udcore.namoptions import parse_namelists
from udcore import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_namelist_parser_reference_style_deck():
    text = """&RUN
iexpnr       = 999
runtime      = 21.
ladaptive    = .true.
nprocx       = 4   ! comment
/

&DOMAIN
itot = 128
xlen = 64
/
&DYNAMICS
iadv_sv = 7, 7
/
"""
    n = parse_namelists(text)
    assert n["RUN"]["iexpnr"] == 999 and n["RUN"]["ladaptive"] is True and n["RUN"]["nprocx"] == 4
    assert n["DOMAIN"]["itot"] == 128 and n["DOMAIN"]["xlen"] == 64
    assert n["DYNAMICS"]["iadv_sv"] == [7, 7]


@pytest.mark.parametrize("name,iexp", sorted({**KERNEL_CASES, **RUN_CASES}.items()))
def test_grid_metrics_match_reference(name, iexp):
    """Grid.from_deck == initglobal (src/modglobal.f90:747-762) bit for bit."""
    fix = load_fixture(name)
    g = Grid.from_deck(read_deck(deck_path(name, iexp)))
    np.testing.assert_array_equal(g.dzf, fix["dzf"].data)
    np.testing.assert_array_equal(g.dzh[1:], fix["dzh"].data)
    meta = fix["meta"].data
    assert (g.nx, g.ny, g.nz) == tuple(int(x) for x in meta[:3])
    assert g.dx == meta[3] and g.dy == meta[4]


@pytest.mark.parametrize("name,iexp", sorted(KERNEL_CASES.items()))
def test_sgs_constants_match_reference(name, iexp):
    fix = load_fixture(name)
    meta = fix["meta"].data
    d = read_deck(deck_path(name, iexp))
    sgs, csz, c_vreman, prandtli = sgs_from_deck(d)
    assert prandtli == meta[15] and c_vreman == meta[16]
    assert abs(csz - meta[17]) <= 1e-15
    # closure precedence of src/modsubgrid.f90:208,269,363: Smagorinsky, Vreman, one-equation, none
    oneeqn = 3 if d.get("NAMSUBGRID", "loneeqn") else 0
    assert sgs == (1 if meta[18] else (2 if meta[19] else oneeqn)) * int(meta[21])


@pytest.mark.parametrize("name,iexp", sorted(RUN_CASES.items()))
def test_cold_start_matches_reference(name, iexp):
    """cold_start == readinitfiles cold start + randomize_field + halos + boundary (s000 dump)."""
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    g = Grid.from_deck(d)
    nsv = int(d.get("SCALARS", "nsv"))
    st = cold_start(g, d, nsv=nsv)
    for k in ("u0", "v0", "w0", "um", "vm", "wm"):
        np.testing.assert_array_equal(st[k], marr(fix, f"s000.{k}", g.nz), err_msg=k)
    for k in ("thl0", "thlm", "qt0", "qtm"):
        if f"s000.{k}" in fix:
            lo = 1 if k.endswith("m") else 0      # the floor ghost of thlm / qtm is never read
            np.testing.assert_array_equal(st[k][lo:, 1:-1, 1:-1], marr(fix, f"s000.{k}", g.nz)[lo:, 1:-1, 1:-1], err_msg=k)
    for n in range(nsv):
        ref = carr(fix, f"s000.sv0_{n + 1:02d}", g.nz)
        np.testing.assert_array_equal(st[f"sv0_{n}"][:, 1:-1, 1:-1], ref[:, 1:-1, 1:-1])


def test_cold_start_slab_equals_global_rows():
    """The LCG noise is keyed on the global index, so a y-slab gets the same values."""
    full = lcg_noise(16, 12, 0, 12, 3)
    part = lcg_noise(16, 12, 4, 4, 3)
    np.testing.assert_array_equal(full[4:8], part)


def test_forcing_matches_reference():
    import udcore
    for name, iexp in RUN_CASES.items():
        fix = load_fixture(name)
        d = read_deck(deck_path(name, iexp))
        nz = d.get("DOMAIN", "ktot")
        dpdx = float(d.get("PHYSICS", "dpdx"))
        dpdxl = [0.0 * vg - pg - dpdx for vg, pg in zip(d.vg, d.pgx)]
        np.testing.assert_allclose(dpdxl, fix["dpdxl"].data[:nz], rtol=0, atol=0)


def test_header_symbols_are_exported():
    """libudcore.so loads and exports every entry point include/udcore.h declares."""
    if not os.path.exists(L.LIBPATH):
        import __graft_entry__ as ge
        ge.build()
    with open(os.path.join(ROOT, "include", "udcore.h")) as f:
        hdr = f.read()
    # (the block under UDC_TEST_TRANSPORT declares the virtual-rank tests' transport, which only libudcore_test.so contains)
    test_part = re.search(r"#ifdef UDC_TEST_TRANSPORT(.*?)#endif", hdr, re.S).group(1)
    test_only = set(re.findall(r"\b(udc_\w+)\s*\(", test_part))
    declared = set(re.findall(r"\b(udc_\w+)\s*\(", hdr)) - test_only
    assert test_only == set(L.TEST_EXPORTS)
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = ctypes.CDLL(L.LIBPATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    for sym in test_only:      # the shipped library does not carry test scaffolding
        assert not hasattr(lib, sym), sym
    assert lib.udc_version() >= 100
    if os.path.exists(L.TESTLIBPATH):
        tl = ctypes.CDLL(L.TESTLIBPATH, mode=ctypes.RTLD_LOCAL)
        for sym in sorted(declared | test_only):
            assert hasattr(tl, sym), sym


def test_no_device_fails_loudly():
    """Without a GPU the product path must refuse to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from udcore.core import DynCore
    with pytest.raises(L.UdcError):
        DynCore(Grid.uniform(8, 8, 8))


def test_scalar_sources_slab_equals_global_rows():
    """udcore/sources.py evaluates the sources per y-slab from the global row offset: stitched slabs == the global field;
    the box registered with the device is the tight bounding box."""
    from udcore.sources import source_field
    g = Grid.uniform(16, 12, 8)
    pts = [(3.1, 2.2, 1.4, 0.8, 0.6)]
    lns = [(1.0, 0.6, 0.5, 6.5, 4.6, 2.5, 0.5, 0.5)]
    full = source_field(g, pts, lns)
    assert full.shape == (8, 12, 16) and full.max() > 0 and (full == 0).any()
    for P in (2, 3):
        nyl = g.ny // P
        parts = [source_field(g, pts, lns, j0=r * nyl, nyl=nyl) for r in range(P)]
        np.testing.assert_array_equal(np.concatenate(parts, axis=1), full)
    # a source outside a slab leaves that slab empty
    far = source_field(g, [(3.0, 0.4, 1.0, 1.0, 0.1)], [], j0=8, nyl=4)
    assert not far.any()


def test_bench_kernel_table_folds_split_launches():
    """bench.py: a kernel launched in pieces on the y-slab path (`<name>_edge` launches ahead of `<name>`) is one row of the kernel
    table with the time of all pieces and the launch count of the last; algorithmic bytes are looked up under the folded name; a
    surveyed launch shorter than the markers' cost carries no rate (no division by zero)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    tab = {"closure_edge": (0.3, 6), "closure": (0.9, 6), "mom_truetruetruetrue_edge": (2.0, 24), "mom_truetruetruetrue": (0.5, 6),
           "project_integrate_edge": (0.4, 12), "project_integrate": (0.8, 6), "halo_pack": (0.01, 6), "only_edge": (0.2, 3)}
    out = b.fold_edges(tab)
    assert out["closure"] == (pytest.approx(1.2), 6) and out["mom_truetruetruetrue"] == (pytest.approx(2.5), 6)
    assert out["project_integrate"] == (pytest.approx(1.2), 6) and out["halo_pack"] == (0.01, 6) and out["only"] == (0.2, 3)
    assert not any(k.endswith("_edge") for k in out)
    assert b.algo_bytes("mom_truetruetruetrue", 0, 1.0) == 64 and b.algo_bytes("mom_truetruetruetrue", 0, 0.0) == 88
    assert b.algo_bytes("project_integrate", 2) == 72 + 48 and b.algo_bytes("halo_pack") is None
