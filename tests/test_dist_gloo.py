"""world_size-2 (gloo, CPU) tests of the N>1 path's host logic and exchange patterns:
slab ownership, decomposition-independent cold start, periodic ghost-row ring, and the
all-to-all spectral transpose of the slab Poisson solve (index maps mirrored by the HIP kernels),
checked against the single-rank CPU oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path[:0] = [os.path.join(ROOT, "u-dales_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import slab_model as slab
    from udcore.grid import Grid, lcg_noise
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    nx, ny, nz = 16, 12, 6
    g = Grid.from_levels(nx, ny, nz, 8.0, 6.0, np.cumsum(0.5 * 1.07 ** np.arange(nz)) - 0.25 * 1.07 ** np.arange(nz))
    j0, j1 = slab.slab_rows(ny, world, rank)
    nyl = j1 - j0
    rng = np.random.default_rng(123)
    full = rng.standard_normal((nz, ny, nx))
    full -= full.mean()                       # compatible RHS (zero mean), like a divergence
    # --- ghost rows: ring exchange reproduces the periodic global neighbours
    a = np.zeros((nz, nyl + 4, nx))
    a[:, 2:-2, :] = full[:, j0:j1, :]
    slab.halo_exchange(a, 2, dist, rank, world)
    np.testing.assert_array_equal(a[:, :2, :], full[:, [(j0 - 2) % ny, (j0 - 1) % ny], :])
    np.testing.assert_array_equal(a[:, -2:, :], full[:, [j1 % ny, (j1 + 1) % ny], :])
    # --- a producer launched edge rows first, its ghost rows travelling while the interior rows are swept (closure / integrate split)
    nyw = 16 * world
    fullw = rng.standard_normal((nz, nyw, nx))
    jw0 = rank * 16
    aw = np.zeros((nz, 16 + 2, nx))
    aw[:, 1:-1, :] = fullw[:, jw0:jw0 + 16, :]
    slab.halo_exchange(aw, 1, dist, rank, world)
    bw = slab.produce_and_exchange_split(aw, 1, 4, dist, rank, world)
    bfull = np.roll(fullw, 1, axis=1) - 2. * fullw + np.roll(fullw, -1, axis=1)
    np.testing.assert_array_equal(bw[:, 1:-1, :], bfull[:, jw0:jw0 + 16, :])
    np.testing.assert_array_equal(bw[:, 0, :], bfull[:, (jw0 - 1) % nyw, :])
    np.testing.assert_array_equal(bw[:, -1, :], bfull[:, (jw0 + 16) % nyw, :])
    # --- cold-start noise is a function of the global index only
    np.testing.assert_array_equal(lcg_noise(nx, ny, j0, nyl, 3), lcg_noise(nx, ny, 0, ny, 3)[j0:j1])
    # --- distributed Poisson solve == single-rank oracle
    p_loc = slab.poisson_slab_model(full[:, j0:j1, :].copy(), g.dx, g.dy, g.dzf, g.dzh, dist, rank, world)
    np.save(os.path.join(tmp, f"p_{rank}.npy"), p_loc)
    if rank == 0:
        np.save(os.path.join(tmp, "rhs.npy"), full)
        np.save(os.path.join(tmp, "grid.npy"), np.array([g.dx, g.dy]))
        np.save(os.path.join(tmp, "dzf.npy"), g.dzf)
        np.save(os.path.join(tmp, "dzh.npy"), g.dzh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_slab_exchange_patterns_gloo(world, tmp_path):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle_lib as ol
    rhs = np.load(tmp_path / "rhs.npy")
    dx, dy = np.load(tmp_path / "grid.npy")
    dzf, dzh = np.load(tmp_path / "dzf.npy"), np.load(tmp_path / "dzh.npy")
    nz, ny, nx = rhs.shape
    o = ol.Oracle(nx, ny, nz, float(dx), float(dy), dzf, dzh)
    p = np.zeros(o.mshape())
    p[1:-1, 1:-1, 1:-1] = rhs
    o.call("orc_poisson_solve", p)
    got = np.concatenate([np.load(tmp_path / f"p_{r}.npy") for r in range(world)], axis=1)
    ref = p[1:-1, 1:-1, 1:-1]
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_split_launch_rows_cover_the_slab_once():
    """tile_rows / tile_row (udc_internal.h, mirrored in slab_model.py): the edge launch and the interior launch together visit
    every tile row of the slab exactly once, for every split the library makes (e = 1)."""
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import slab_model as slab
    for gy in range(3, 40):
        edge, inner = slab.tile_rows(gy, 1, True), slab.tile_rows(gy, 1, False)
        rows = [slab.tile_row(edge, b) for b in range(edge[0])] + [slab.tile_row(inner, b) for b in range(inner[0])]
        assert sorted(rows) == list(range(gy)), gy
        assert [slab.tile_row(edge, b) for b in range(edge[0])] == [0, gy - 1]


def test_momentum_pipe_ranges_feed_their_chunk():
    """k_momentum_pipe_stage's level ranges (mirrored in slab_model.py): together they cover every level once, in order, and when
    stage c has run every level the x forward transform of k-chunk c reads is there -- its own levels and, for the divergence's
    pwp(k + 1), the first level of the next chunk."""
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import slab_model as slab
    for nz in (16, 24, 64, 256, 512):
        for nch in (1, 2, 4, 8):
            if nz % nch or nz // nch < 4:
                continue
            r = slab.momentum_pipe_ranges(nz, nch)
            assert r[0][0] == 0 and r[-1][1] == nz and all(r[c][1] == r[c + 1][0] for c in range(nch - 1)), (nz, nch, r)
            nzc = nz // nch
            for c in range(nch):
                need_top = min((c + 1) * nzc, nz - 1)          # highest level read: k + 1 for k = (c + 1) nzc - 1, none above the lid
                assert r[c][1] > need_top and r[c][0] <= c * nzc + (0 if c == 0 else 1), (nz, nch, c, r)


def test_slab_index_maps_roundtrip():
    sys.path[:0] = [os.path.join(ROOT, "u-dales_amd")]
    import slab_model as slab
    rng = np.random.default_rng(1)
    for P, nx, nyl, nz in ((1, 8, 4, 3), (2, 16, 6, 2), (4, 10, 2, 3), (8, 256, 4, 2)):
        nkx, cx = slab.kx_chunk(nx, P)
        specA = rng.standard_normal((nz, nyl, nkx)) + 1j * rng.standard_normal((nz, nyl, nkx))
        send = slab.pack_fwd(specA, P)
        assert send.shape == (P, nz, cx, nyl)
        # all ranks identical here: "receiving" my own blocks from everyone = send itself
        back = slab.unpack_bwd(slab.pack_bwd(slab.unpack_fwd(send), P), nkx)
        # with every rank holding the same rows, block s of the return trip carries kx chunk s
        np.testing.assert_array_equal(back[:, :, :cx], specA[:, :, :cx])
    with pytest.raises(ValueError):
        slab.slab_rows(10, 4, 0)
