#!/usr/bin/env python3
"""Sub-sampled golden fixtures at BASELINE configs[1] and configs[2]'s FULL sizes, from the reference's own Fortran.

    python tests/golden/make_full_size.py [c1] [c2]        # needs oracle/_ref/udales_ref (make -C oracle ref); ~40 + ~15 min of one core

  c1  256 x 256 x 256 neutral channel (bench.py's deck: Vreman, floor wall function, fixed dt = 0.25), 100 steps = 300 RK3 substeps
      (SURVEY.md section 8(d) "Parity run")
  c2  512 x 512 x 256, Smagorinsky + one kappa-advected scalar (linear profile), 3 steps = 9 substeps
  c3s 1024 x 64 x 512 neutral channel, 3 steps = 9 substeps: one rank's slab of eight of configs[3] as a whole domain

oracle/_ref/udales_ref (the reference's unmodified src/ under oracle/ref_driver.f90, see make_golden.py) runs the deck that
bench.write_deck writes and dumps its state after the last substep; what is kept per field (u0, v0, w0, pres0[, sv0_01], interior cells):
  sample       every 8th cell (c1) / every 16th in x and y and 8th in z (c2), starting at (3, 5, 1) -- 32^3 values, 262 kB
  level_sum    the sum over each level (all cells)
  level_amax   max |.| over each level (all cells)
(one file per field: tests/golden/full_size_<case>_<field>.npz)
so that a full-size device run is compared value by value on the sample and through two reductions over every cell.  Data only.
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "udales_ref")

CASES = {      # name: (iexpnr, nx, ny, nz, nsub, write_deck keywords, sample strides (x, y, z), fields)
    "c1": (77, 256, 256, 256, 300, {}, (8, 8, 8), ("u0", "v0", "w0", "pres0")),
    "c2": (79, 512, 512, 256, 9, dict(nsv=1, sgs="smag"), (16, 16, 8), ("u0", "v0", "w0", "pres0", "sv0_01")),
    # one rank's slab of eight of configs[3] (1024 x 512 x 512 on 8 GPUs) as a whole domain: the shapes the multi-GPU kernels work on
    # (x lines of 1024, columns of 512 levels); configs[3]'s own grid does not fit this container's memory under the reference
    "c3s": (81, 1024, 64, 512, 9, {}, (32, 4, 16), ("u0", "v0", "w0", "pres0")),
}
OFFSET = (3, 5, 1)


def write_case_deck(d, name):
    from bench import write_deck
    iexp, nx, ny, nz, nsub, kw, _, _ = CASES[name]
    path = write_deck(d, iexp, nx, ny, nz, nsub, **kw)
    with open(path) as f:
        txt = f.read().replace(f"nsub = {nsub}", f"nsub = {nsub}\ndump_at = {nsub}")
    with open(path, "w") as f:
        f.write(txt)
    return path


def records(path):
    """(name, lower bounds, memory-mapped [k, j, i] array) of every 3-D record of a ref_driver dump, without reading the file whole"""
    size = os.path.getsize(path)
    pos = 0
    with open(path, "rb") as f:
        while pos < size:
            f.seek(pos)
            name = f.read(16).decode("ascii").strip()
            hdr = struct.unpack("<7i", f.read(28))
            pos += 44
            if hdr[0] == 1:
                cnt = hdr[4] - hdr[1] + 1
            else:
                lb, ub = hdr[1:4], hdr[4:7]
                shp = tuple(ub[d] - lb[d] + 1 for d in range(3))
                cnt = shp[0] * shp[1] * shp[2]
                yield name, lb, np.memmap(path, dtype="<f8", mode="r", offset=pos, shape=(shp[2], shp[1], shp[0]))
            pos += 8 * cnt


def sample(a, lb, n, strides, halo_of_field):
    """interior cells of a dumped array (Fortran bounds lb: cell 1 is index 1 - lb) -> the kept values"""
    nx, ny, nz = n
    o = [1 - lb[0], 1 - lb[1], 1 - lb[2]]
    inner = a[o[2]:o[2] + nz, o[1]:o[1] + ny, o[0]:o[0] + nx]
    sx, sy, sz = strides
    sub = np.array(inner[OFFSET[2]::sz, OFFSET[1]::sy, OFFSET[0]::sx])
    ssum = np.array([np.sum(np.asarray(inner[k]), dtype=np.float64) for k in range(nz)])
    amax = np.array([np.abs(np.asarray(inner[k])).max() for k in range(nz)])
    return sub, ssum, amax


def make(name, workdir=None):
    iexp, nx, ny, nz, nsub, kw, strides, fields = CASES[name]
    d = workdir or tempfile.mkdtemp(prefix=f"gold_{name}_")
    dump = os.path.join(d, "ref.bin")
    if not os.path.exists(dump):
        write_case_deck(d, name)
        r = subprocess.run(f"ulimit -s unlimited; exec {REF} namoptions.{iexp:03d} run ref.bin", shell=True, cwd=d,
                           capture_output=True, text=True, executable="/bin/bash")
        assert r.returncode == 0, r.stderr[-2000:]
    out = {"nsub": np.int64(nsub), "shape": np.array([nx, ny, nz]), "strides": np.array(strides), "offset": np.array(OFFSET)}
    want = {f"s{nsub:03d}.{f}": f for f in fields}
    for rec, lb, a in records(dump):
        if rec in want:
            f = want[rec]
            sub, ssum, amax = sample(a, lb, (nx, ny, nz), strides, 0)
            out[f], out[f + "_sum"], out[f + "_amax"] = sub, ssum, amax
            print(name, f, sub.shape, "max", amax.max())
    missing = [f for f in fields if f not in out]
    assert not missing, missing
    meta = {k: out[k] for k in ("nsub", "shape", "strides", "offset")}
    for f in fields:      # one file per field (each under 300 kB)
        path = os.path.join(HERE, f"full_size_{name}_{f}.npz")
        np.savez_compressed(path, sample=out[f], level_sum=out[f + "_sum"], level_amax=out[f + "_amax"], **meta)
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    args = sys.argv[1:] or ["c1", "c2"]
    for a in args:
        name, _, wd = a.partition("=")
        make(name, wd or None)
