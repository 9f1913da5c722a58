"""Index algebra of the register line transforms of udc_fft.hip (ffty_natreg_kernel / ffty_slabreg_kernel), restated in numpy and held
against numpy.fft: fft16 as two radix-4 layers with X[q + 4 r] left in slot 4 q + r, fft8 / fft32 as one radix-2 layer over two halves,
and a line of N = 16 x N2 points as n = N2 n1 + n2, k = k1 + 16 k2 with the twiddle W_N^(n2 k1) between the two steps; the inverse as
conj(FFT(conj(x))).  (CPU test: the device kernels themselves are compared with rocFFT's path and the oracle in test_gpu_own_forward.py.)"""
import numpy as np
import pytest


def regpos16(k):
    return 4 * (k & 3) + (k >> 2)


def regpos(n, k):
    if n == 16:
        return regpos16(k)
    if n == 8:
        return 4 + (k >> 1) if k & 1 else k >> 1
    return 16 + regpos16(k >> 1) if k & 1 else regpos16(k >> 1)


def r4(a, b, c, d):
    t0, t1, t2, e = a + c, a - c, b + d, b - d
    t3 = -1j * e
    return t0 + t2, t1 + t3, t0 - t2, t1 - t3


def fft4(x):
    return list(r4(*x))


def fft16(x):
    x = list(x)
    for j in range(4):
        x[j], x[j + 4], x[j + 8], x[j + 12] = r4(x[j], x[j + 4], x[j + 8], x[j + 12])
        for q in range(1, 4):
            x[j + 4 * q] *= np.exp(-2j * np.pi * j * q / 16)
    for q in range(4):
        x[4 * q:4 * q + 4] = r4(*x[4 * q:4 * q + 4])
    return x


def fft_reg(x):
    n = len(x)
    if n == 16:
        return fft16(x)
    h = n // 2
    x = list(x)
    for j in range(h):
        a, b = x[j] + x[j + h], x[j] - x[j + h]
        x[j], x[j + h] = a, b * np.exp(-2j * np.pi * j / n)
    sub = fft4 if h == 4 else fft16
    return sub(x[:h]) + sub(x[h:])


@pytest.mark.parametrize("n", [8, 16, 32])
def test_register_transforms(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y = fft_reg(x)
    ref = np.fft.fft(x)
    for k in range(n):
        assert abs(y[regpos(n, k)] - ref[k]) < 1e-13


@pytest.mark.parametrize("n2", [8, 16, 32])
@pytest.mark.parametrize("inverse", [False, True])
def test_line_of_16_times_n2(n2, inverse):
    n = 16 * n2
    rng = np.random.default_rng(n + inverse)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    src = np.conj(x) if inverse else x
    lds = np.zeros((16, n2), complex)
    for t in range(n2):                                   # thread (column, n2 = t)
        y = fft16([src[n2 * n1 + t] for n1 in range(16)])
        for k1 in range(16):
            lds[k1, t] = y[regpos16(k1)] * np.exp(-2j * np.pi * t * k1 / n)
    out = np.zeros(n, complex)
    for t in range(16):                                   # thread (column, k1 = t)
        z = fft_reg(list(lds[t, :]))
        for k2 in range(n2):
            out[t + 16 * k2] = z[regpos(n2, k2)]
    if inverse:
        out = np.conj(out)
    ref = np.fft.ifft(x) * n if inverse else np.fft.fft(x)
    assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
