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
