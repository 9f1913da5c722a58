"""The order of a fused substep as a table, enumerated.

substep_fused (u-dales_amd/csrc/udc_api.hip) no longer decides anything itself: it asks plan_substep (udc_plan.h, a pure function of the
switches, of what the handle is and of the call) and executes the answer.  Here that function -- compiled by g++ into
u-dales_amd/lib/libudcplan.so, no GPU involved -- is run over EVERY combination of its inputs (all 2^7 switch settings x a lattice of
configurations and calls: ~19 million rows) and compared with the table as DESIGN.md states it, written down a second time below in
numpy; and the invariants that make an order safe are checked on every row."""
import ctypes
import itertools
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "u-dales_amd", "lib", "libudcplan.so")

IN = ["no_fold", "no_alias", "ek_always", "halo_overlap", "mom_pipe", "div_in_fft", "ptotal",
      "slab", "comm_stream", "sgs", "lbuoycorr", "nslots", "ibm_on", "stats_any", "fft_fused", "own_fwd", "tend_plane", "between",
      "closure_tile_rows", "mom_tile_rows", "int_tile_rows", "x_row_groups", "levels_per_chunk", "p_transpose", "open_lid", "lid_masked", "rk3step", "um_alias", "ibm_edits_now"]
OUT = ["lds", "pup", "fold", "alias_ok", "materialise_um", "rotate", "skip_um", "closure", "need_ekh", "mom_pipe", "div_in_fft",
       "vp_row", "p_row", "integrate", "ptotal"]
FOLDED, OVERLAPPED, PLAIN = 0, 1, 2
ROW_FOLDED, ROW_BESIDE, ROW_INLINE, ROW_PIPED, ROW_TRANSPOSED = 0, 1, 2, 3, 4
INT_ONE, INT_EDGES_FIRST = 0, 1


def lib():
    if not os.path.exists(LIB):
        pytest.skip("u-dales_amd/lib/libudcplan.so not built")
    L = ctypes.CDLL(LIB)
    assert L.udc_plan_fields() == len(IN) and L.udc_plan_outputs() == len(OUT)
    L.udc_plan_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    return L


def run(L, rows):
    a = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.zeros((a.shape[0], len(OUT)), dtype=np.int32)
    L.udc_plan_batch(a.ctypes.data, a.shape[0], out.ctypes.data)
    return {k: out[:, q] for q, k in enumerate(OUT)}


def table(i):
    """DESIGN.md section 7, restated: i = dict of input columns -> dict of expected output columns."""
    b = lambda x: x.astype(bool)      # noqa: E731
    lds = np.ones_like(i["slab"], dtype=bool)
    pup = lds
    fold = lds & ~b(i["slab"]) & ~b(i["no_fold"])
    alias_ok = pup & ~b(i["no_alias"]) & ~b(i["ibm_on"]) & ~b(i["open_lid"])
    mat = b(i["um_alias"]) & ~(alias_ok & (i["rk3step"] == 1))
    rotate = b(i["um_alias"]) & ~mat
    skip = alias_ok & (i["rk3step"] == 3)
    beside = lambda rows: b(i["slab"]) & b(i["comm_stream"]) & b(i["halo_overlap"]) & (rows >= 3)      # noqa: E731
    sv = (i["sgs"] == 1) | (i["sgs"] == 2)
    c_folded = fold & sv & ~b(i["lbuoycorr"])
    c_over = ~c_folded & lds & sv & ~b(i["lbuoycorr"]) & beside(i["closure_tile_rows"])
    closure = np.where(c_folded, FOLDED, np.where(c_over, OVERLAPPED, PLAIN))
    need_ekh = np.where(c_folded | c_over, b(i["ek_always"]) | (i["rk3step"] == 3) | (i["nslots"] > 0) | b(i["stats_any"]), True)
    pipe = (b(i["slab"]) & lds & pup & b(i["mom_pipe"]) & b(i["fft_fused"]) & b(i["div_in_fft"]) & beside(i["mom_tile_rows"])
            & (i["nslots"] == 0) & (i["sgs"] != 3) & ~b(i["between"]) & (i["x_row_groups"] >= 2) & (i["levels_per_chunk"] >= 4) & ~b(i["open_lid"]))
    div = pup & ((b(i["slab"]) & b(i["fft_fused"]) & b(i["div_in_fft"])) | (~b(i["slab"]) & b(i["own_fwd"])))
    needs_row = ~fold | (b(i["ibm_on"]) & b(i["ibm_edits_now"]))
    vp = np.where(pipe, ROW_PIPED, np.where(needs_row, np.where(div & beside(np.full_like(i["sgs"], 3)) & (i["x_row_groups"] >= 2), ROW_BESIDE, ROW_INLINE),
                                            ROW_FOLDED))
    prow = np.where(fold, ROW_FOLDED, np.where(b(i["slab"]) & b(i["fft_fused"]) & b(i["p_transpose"]), ROW_TRANSPOSED,
                                               np.where(beside(i["int_tile_rows"]) & (i["int_tile_rows"] >= 4), ROW_BESIDE, ROW_INLINE)))
    integ = np.where(~fold & beside(i["int_tile_rows"]), INT_EDGES_FIRST, INT_ONE)
    ptot = b(i["ptotal"]) & pup & ~b(i["tend_plane"]) & ~(b(i["open_lid"]) & b(i["lid_masked"]))
    return dict(lds=lds, pup=pup, fold=fold, alias_ok=alias_ok, materialise_um=mat, rotate=rotate, skip_um=skip, closure=closure,
                need_ekh=need_ekh, mom_pipe=pipe, div_in_fft=div, vp_row=vp, p_row=prow, integrate=integ, ptotal=ptot)


def lattice(p_transpose=1, open_lid=0):
    """every configuration / call for one setting of the switches (and of the two inputs that are fixed per handle)"""
    axes = dict(slab=[0, 1], sgs=[0, 1, 2, 3], lbuoycorr=[0, 1], nslots=[0, 2], ibm_on=[0, 1], stats_any=[0, 1], fft_fused=[0, 1],
                own_fwd=[0, 1], tend_plane=[0, 1], between=[0, 1], rows=[2, 3, 8], x_row_groups=[1, 4], levels_per_chunk=[2, 16], rk3step=[1, 2, 3],
                um_alias=[0, 1], ibm_edits_now=[0, 1])
    grids = np.meshgrid(*[np.array(v, dtype=np.int32) for v in axes.values()], indexing="ij")
    cols = {k: g.ravel() for k, g in zip(axes, grids)}
    cols["p_transpose"] = np.full_like(cols["slab"], p_transpose)
    cols["open_lid"] = np.full_like(cols["slab"], open_lid)
    cols["lid_masked"] = cols["open_lid"] * cols["ibm_on"] * (cols["sgs"] % 2)      # (obstacles that reach the lid: some of the decks with obstacles)
    cols["comm_stream"] = cols["slab"].copy()      # (the communication stream exists exactly where the slab layout was set up)
    for k in ("closure_tile_rows", "mom_tile_rows", "int_tile_rows"):
        cols[k] = cols["rows"]
    del cols["rows"]
    return cols


def test_every_combination_matches_the_table_and_is_safe():
    L = lib()
    total = 0
    # every setting of the seven order switches for the usual handle (closed lid, p's rows inside the transpose); every 8th setting
    # for the other three kinds of handle
    runs = [(1, 0, sw) for sw in itertools.product([0, 1], repeat=7)]
    runs += [(pt, lid, sw) for pt, lid in ((0, 0), (1, 1), (0, 1)) for q, sw in enumerate(itertools.product([0, 1], repeat=7)) if q % 8 == 3]
    bases = {}
    for pt, lid, sw in runs:
        if (pt, lid) not in bases:
            bases[(pt, lid)] = lattice(pt, lid)
        base = bases[(pt, lid)]
        n = len(base["slab"])
        i = dict(base)
        for name, v in zip(IN[:7], sw):
            i[name] = np.full(n, v, dtype=np.int32)
        rows = np.stack([i[k] for k in IN], axis=1)
        got = run(L, rows)
        exp = table(i)
        for k in OUT:
            assert np.array_equal(got[k], np.asarray(exp[k]).astype(np.int32)), (k, sw)
        g = {k: v.astype(np.int64) for k, v in got.items()}
        slab, over = i["slab"].astype(bool), i["halo_overlap"].astype(bool)
        anything_beside = (g["closure"] == OVERLAPPED) | (g["vp_row"] == ROW_BESIDE) | (g["vp_row"] == ROW_PIPED) | (g["p_row"] == ROW_BESIDE) | \
                          (g["integrate"] == INT_EDGES_FIRST) | (g["mom_pipe"] == 1)
        # nothing travels beside a sweep on a single slab, or with the overlap switched off
        assert not (anything_beside & (~slab | ~over)).any()
        # a single slab never exchanges at all: every ghost row is folded into the owning kernel unless folding is off
        assert not ((g["fold"] == 1) & slab).any()
        assert not ((g["vp_row"] == ROW_FOLDED) & (g["fold"] == 0)).any() and not ((g["p_row"] == ROW_FOLDED) & (g["fold"] == 0)).any()
        # the pipelined sweep implies its row already travelling, the divergence inside the transform, no scalar, no other term
        m = g["mom_pipe"] == 1
        assert ((g["vp_row"] == ROW_PIPED) == m).all() and (g["div_in_fft"][m] == 1).all() and (i["nslots"][m] == 0).all() and (i["between"][m] == 0).all()
        # p's rows inside the backward transpose: only on slabs with the own line transforms (the blocks' layout is theirs)
        assert not ((g["p_row"] == ROW_TRANSPOSED) & ((i["slab"] == 0) | (i["fft_fused"] == 0) | (i["p_transpose"] == 0))).any()
        # p's row beside the first interior rows only with the edges-first integration
        assert not ((g["p_row"] == ROW_BESIDE) & (g["integrate"] != INT_EDGES_FIRST)).any()
        # um aliasing: rotation only on stage 1 of an aliased state, the skip only on stage 3, never with obstacles; and an aliased state
        # is either rotated or materialised
        assert not ((g["rotate"] == 1) & (i["rk3step"] != 1)).any() and not ((g["skip_um"] == 1) & (i["rk3step"] != 3)).any()
        assert not (((g["rotate"] == 1) | (g["skip_um"] == 1)) & (i["ibm_on"] == 1)).any()
        assert ((g["rotate"] + g["materialise_um"]) == i["um_alias"]).all()
        # the divergence inside a transform needs the predicted-velocity form of the tendencies
        assert not ((g["div_in_fft"] == 1) & (g["pup"] == 0)).any()
        # the pressure-total form: never where the tendencies are summed over less than the periodic volume (outflow-rate mass
        # correction, volume flow over the fluid cells of an immersed boundary), only over the predicted-velocity form
        pt = g["ptotal"] == 1
        assert not (pt & ((i["tend_plane"] == 1) | (g["pup"] == 0))).any()
        # the open lid (BCtopm = 3): its rows of bcpup / tderive / tstep_integrate are plane kernels that read and write wm(ke+1), wp(ke+1) under
        # their own names: no aliasing, no pipelined sweep; the pressure-total form only where the lid's slab mean is the zero mode's
        # (no obstacle reaches level ke)
        lid = i["open_lid"] == 1
        assert not (lid & ((g["mom_pipe"] == 1) | (g["skip_um"] == 1) | (g["rotate"] == 1))).any()
        assert not (pt & (i["lid_masked"] == 1)).any() and (pt & lid).any() == bool(lid.any() and sw[6])
        total += n
    assert total == len(runs) * n and len(runs) == 128 + 3 * 16 and n > 100000


def test_named_configurations():
    """The rows of DESIGN.md section 7's table for the BASELINE configurations, with the library's defaults."""
    L = lib()
    dflt = dict(no_fold=0, no_alias=0, ek_always=0, halo_overlap=1, mom_pipe=1, div_in_fft=1, ptotal=1, tend_plane=0, lbuoycorr=0, stats_any=0,
                ibm_edits_now=0, um_alias=0, open_lid=0, lid_masked=0, p_transpose=1)

    def one(**kw):
        i = dict(dflt, **kw)
        g = run(L, np.array([[i[k] for k in IN]], dtype=np.int32))
        return {k: int(v[0]) for k, v in g.items()}
    # configs[1]: 256^3 on one GPU, Vreman, no scalars
    single = dict(slab=0, comm_stream=0, sgs=2, nslots=0, ibm_on=0, fft_fused=0, own_fwd=1, between=0, closure_tile_rows=32, mom_tile_rows=32,
                  int_tile_rows=64, x_row_groups=0, levels_per_chunk=256)
    p = one(rk3step=1, **single)
    assert (p["fold"], p["closure"], p["need_ekh"], p["mom_pipe"], p["div_in_fft"], p["vp_row"], p["p_row"], p["integrate"]) == (1, FOLDED, 0, 0, 1, ROW_FOLDED, ROW_FOLDED, INT_ONE)
    p = one(rk3step=3, **single)
    assert p["need_ekh"] == 1 and p["skip_um"] == 1
    p = one(rk3step=1, **dict(single, um_alias=1))
    assert p["rotate"] == 1 and p["materialise_um"] == 0
    # configs[2]: one GPU, Smagorinsky + a kappa scalar: ekh on every stage
    p = one(rk3step=2, **dict(single, sgs=1, nslots=1))
    assert p["need_ekh"] == 1 and p["closure"] == FOLDED
    # configs[3]: 1024 x 512 x 512 on eight GPUs (slab 1024 x 64 x 512): everything beside a sweep, the sweep under the forward transposes
    slab = dict(slab=1, comm_stream=1, sgs=2, nslots=0, ibm_on=0, fft_fused=1, own_fwd=0, between=0, closure_tile_rows=8, mom_tile_rows=8,
                int_tile_rows=16, x_row_groups=16, levels_per_chunk=128)
    p = one(rk3step=2, **slab)
    assert (p["fold"], p["closure"], p["mom_pipe"], p["div_in_fft"], p["vp_row"], p["p_row"], p["integrate"]) == (0, OVERLAPPED, 1, 1, ROW_PIPED, ROW_TRANSPOSED, INT_EDGES_FIRST)
    # ... the first rung below the defaults of bench.py's ladder (UDC_HALO_OVERLAP=0 UDC_MOM_PIPE=0): every exchange in line
    p = one(rk3step=2, **dict(slab, halo_overlap=0, mom_pipe=0))
    assert (p["closure"], p["mom_pipe"], p["vp_row"], p["p_row"], p["integrate"]) == (PLAIN, 0, ROW_INLINE, ROW_TRANSPOSED, INT_ONE)      # (p's rows still ride in the transpose: no exchange to put in line)
    # configs[4]: the cube array on eight GPUs: obstacles act between the sweep and the solve, so the sweep is not pipelined, and um is
    # never aliased; the rest still travels beside the sweeps
    p = one(rk3step=3, **dict(slab, ibm_on=1, between=1, ibm_edits_now=1, nslots=0))
    assert (p["mom_pipe"], p["vp_row"], p["alias_ok"], p["skip_um"], p["closure"], p["integrate"]) == (0, ROW_BESIDE, 0, 0, OVERLAPPED, INT_EDGES_FIRST)
