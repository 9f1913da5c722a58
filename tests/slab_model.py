"""TEST SUPPORT (not part of the product package).  Index maps of the y-slab decomposition as the HIP kernels in
u-dales_amd/csrc/udc_pois.hip (slab_pack_*/slab_unpack_*), udc_fft.hip and udc_halo.hip (halo_pack/unpack) use them.

Also a small numpy + torch.distributed *model* of the two collective patterns (neighbour ghost rows,
all-to-all spectral transpose).  The model is what the CPU (gloo, world_size 2) tests run: the
production transport is RCCL inside libudcore, which needs GPUs.
"""
from __future__ import annotations

import numpy as np


def slab_rows(jtot: int, nranks: int, rank: int):
    """Global 0-based row range [j0, j1) owned by `rank` (jtot % nranks == 0, src/modstartup.f90:730-760)."""
    if jtot % nranks:
        raise ValueError("jtot must be divisible by the number of slabs")
    nyl = jtot // nranks
    return rank * nyl, (rank + 1) * nyl


def kx_chunk(nx: int, nranks: int):
    """(nkx, cx): number of r2c modes in x and the per-rank chunk ceil(nkx / nranks)."""
    nkx = nx // 2 + 1
    return nkx, (nkx + nranks - 1) // nranks


def pack_fwd(specA: np.ndarray, nranks: int) -> np.ndarray:
    """specA[k, j, kx] (local rows) -> send[d, k, kxl, j], zero padded for kx >= nkx."""
    nz, nyl, nkx = specA.shape
    cx = (nkx + nranks - 1) // nranks
    pad = np.zeros((nz, nyl, cx * nranks), dtype=specA.dtype)
    pad[:, :, :nkx] = specA
    return np.ascontiguousarray(pad.reshape(nz, nyl, nranks, cx).transpose(2, 0, 3, 1))


def unpack_fwd(recv: np.ndarray) -> np.ndarray:
    """recv[s, k, kxl, j] -> specB[k, kxl, y] with y = s*nyl + j."""
    P, nz, cx, nyl = recv.shape
    return np.ascontiguousarray(recv.transpose(1, 2, 0, 3).reshape(nz, cx, P * nyl))


def pack_bwd(specB: np.ndarray, nranks: int) -> np.ndarray:
    nz, cx, ny = specB.shape
    nyl = ny // nranks
    return np.ascontiguousarray(specB.reshape(nz, cx, nranks, nyl).transpose(2, 0, 1, 3))


def unpack_bwd(recv: np.ndarray, nkx: int) -> np.ndarray:
    """recv[s, k, kxl, j] -> specA[k, j, kx], kx = s*cx + kxl < nkx."""
    P, nz, cx, nyl = recv.shape
    return np.ascontiguousarray(recv.transpose(1, 3, 0, 2).reshape(nz, nyl, P * cx)[:, :, :nkx])


# ---------------------------------------------------------------- torch.distributed model
def halo_exchange(a: np.ndarray, width: int, dist, rank: int, nranks: int) -> None:
    """In place: a[k, j, i] with `width` ghost rows on each side of the local rows (periodic ring)."""
    import torch
    prev, nxt = (rank - 1) % nranks, (rank + 1) % nranks
    to_prev = torch.from_numpy(np.ascontiguousarray(a[:, width:2 * width, :]))
    to_next = torch.from_numpy(np.ascontiguousarray(a[:, -2 * width:-width, :]))
    from_prev, from_next = torch.empty_like(to_next), torch.empty_like(to_prev)
    if nranks == 1:
        from_next.copy_(to_prev)
        from_prev.copy_(to_next)
    else:
        reqs = [dist.isend(to_prev, prev, tag=1), dist.isend(to_next, nxt, tag=2),
                dist.irecv(from_next, nxt, tag=1), dist.irecv(from_prev, prev, tag=2)]
        for r in reqs:
            r.wait()
    a[:, :width, :] = from_prev.numpy()
    a[:, -width:, :] = from_next.numpy()


def alltoall(send: np.ndarray, dist, nranks: int) -> np.ndarray:
    import torch
    if nranks == 1:
        return send.copy()
    s = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(send))).contiguous()
    r = torch.empty_like(s)
    dist.all_to_all_single(r, s)
    return torch.view_as_complex(r).numpy()


def poisson_slab_model(rhs_local: np.ndarray, dx, dy, dzf, dzh, dist, rank: int, nranks: int) -> np.ndarray:
    """Distributed solve of the reference's Poisson problem (src/modpois.f90:440-712) with the slab
    layout: rfft in x -> all-to-all -> fft in y + Thomas in z -> ifft in y -> all-to-all -> irfft in x.
    rhs_local[k, j, i] are this rank's rows; returns p for the same rows."""
    nz, nyl, nx = rhs_local.shape
    ny = nyl * nranks
    nkx, cx = kx_chunk(nx, nranks)
    specA = np.fft.rfft(rhs_local, axis=2)
    specB = unpack_fwd(alltoall(pack_fwd(specA, nranks), dist, nranks))
    specB = np.fft.fft(specB, axis=2) / (nx * ny)
    kx = rank * cx + np.arange(cx)
    xrt = np.where(kx < nkx, -4. / dx ** 2 * np.sin(np.pi * np.minimum(kx, nkx - 1) / nx) ** 2, -1.0)
    ky = np.arange(ny)
    yrt = -4. / dy ** 2 * np.sin(np.pi * np.minimum(ky, ny - ky) / ny) ** 2
    ev = np.where((kx < nkx)[:, None], xrt[:, None] + yrt[None, :], -1.0)
    a = np.zeros(nz + 2); b = np.zeros(nz + 2); c = np.zeros(nz + 2)
    for k in range(1, nz + 1):
        a[k] = 1. / (dzf[k] * dzh[k]); c[k] = 1. / (dzf[k] * dzh[k + 1]); b[k] = -(a[k] + c[k])
    b[1] += a[1]
    btopD = b[nz] - c[nz]
    b[nz] = b[nz] + c[nz]
    a[1] = 0.; c[nz] = 0.
    x = specB
    d = np.zeros((nz + 1,) + ev.shape)
    z = 1. / (b[1] + ev)
    d[1] = c[1] * z
    x[0] = x[0] * z
    for k in range(2, nz):
        z = 1. / (b[k] + ev - a[k] * d[k - 1])
        d[k] = c[k] * z
        x[k - 1] = (x[k - 1] - a[k] * x[k - 2]) * z
    bbk = np.where(ev == 0., btopD, b[nz] + ev)
    z = bbk - a[nz] * d[nz - 1]
    x[nz - 1] = (x[nz - 1] - a[nz] * x[nz - 2]) / z
    for k in range(nz - 1, 0, -1):
        x[k - 1] = x[k - 1] - d[k] * x[k]
    specB = np.fft.ifft(x, axis=2) * ny
    specA = unpack_bwd(alltoall(pack_bwd(specB, nranks), dist, nranks), nkx)
    return np.fft.irfft(specA, n=nx, axis=2) * nx


# ---------------------------------------------------------------- split launches (udc_internal.h: TileGrid row subsets, udc_halo.hip)
def tile_rows(gy_full: int, e: int, edge: bool):
    """TileGrid of a split launch as (gy, y0, ysplit, yjump): edge = the first and last `e` tile rows, else the rows in between."""
    if edge:
        return 2 * e, 0, e, gy_full - 2 * e
    return gy_full - 2 * e, e, 1 << 30, 0


def tile_row(t, b: int) -> int:
    """local tile row b of a launch -> tile row of the slab (tile_row in udc_internal.h)"""
    gy, y0, ysplit, yjump = t
    return y0 + b + (yjump if b >= ysplit else 0)


def produce_and_exchange_split(a: np.ndarray, width: int, tile_h: int, dist, rank: int, nranks: int) -> np.ndarray:
    """What k_closure_lds(rows = 1) -> k_halo_y_begin -> k_closure_lds(rows = 2) -> k_halo_y_join do, on a [k, j, i] array `a` with one
    valid ghost row on each side: a 3-point y stencil b[j] = a[j-1] - 2 a[j] + a[j+1] evaluated over the edge tile rows first, the
    ghost rows of b posted (not waited for), the interior tile rows evaluated, then the wait and the unpack.  -> b with `width` ghost
    rows.  The interior evaluation must not touch a row the exchange reads or writes (asserted)."""
    import torch
    nz, nyg, nx = a.shape
    nyl = nyg - 2
    gy = (nyl + tile_h - 1) // tile_h
    assert gy >= 3
    b = np.full((nz, nyl + 2 * width, nx), np.nan)

    def sweep(t):
        rows = []
        for bl in range(t[0]):
            r = tile_row(t, bl)
            for j in range(r * tile_h, min((r + 1) * tile_h, nyl)):
                b[:, width + j, :] = a[:, j, :] - 2. * a[:, j + 1, :] + a[:, j + 2, :]
                rows.append(j)
        return rows
    edge_rows = sweep(tile_rows(gy, 1, True))
    prev, nxt = (rank - 1) % nranks, (rank + 1) % nranks
    to_prev = torch.from_numpy(np.ascontiguousarray(b[:, width:2 * width, :]))
    to_next = torch.from_numpy(np.ascontiguousarray(b[:, -2 * width:-width, :]))
    assert not np.isnan(to_prev.numpy()).any() and not np.isnan(to_next.numpy()).any()      # the edge launch wrote what travels
    from_prev, from_next = torch.empty_like(to_next), torch.empty_like(to_prev)
    reqs = [dist.isend(to_prev, prev, tag=11), dist.isend(to_next, nxt, tag=12),
            dist.irecv(from_next, nxt, tag=11), dist.irecv(from_prev, prev, tag=12)]
    inner_rows = sweep(tile_rows(gy, 1, False))
    assert sorted(edge_rows + inner_rows) == list(range(nyl))                                # every row exactly once
    assert min(inner_rows) >= width and max(inner_rows) < nyl - width                        # the interior launch stays off the travelling rows
    for r in reqs:
        r.wait()
    b[:, :width, :] = from_prev.numpy()
    b[:, -width:, :] = from_next.numpy()
    return b


# ---------------------------------------------------------------- the momentum sweep pipelined with the solve (udc_api.hip)
def momentum_pipe_ranges(nz: int, nch: int):
    """Level ranges [kbeg, kend) of k_momentum_pipe_stage(c), c = 0 .. nch - 1 (tile rows 1 ..; tile row 0 is swept over all levels
    before the first stage)."""
    nzc = nz // nch
    return [(0 if c == 0 else c * nzc + 1, nz if c == nch - 1 else (c + 1) * nzc + 1) for c in range(nch)]
