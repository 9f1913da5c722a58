"""BASELINE-size property tests on the device: 512 x 512 x 256 with Smagorinsky + one kappa-advected scalar
(BASELINE.json configs[2]) and 1024 x 512 x 512 neutral Vreman (configs[3] on one GPU).  No CPU reference exists at
these sizes within a test budget, so the checks are the size-independent properties of the path:
  * the projected velocity is divergence-free to round-off (chkdiv's divmax, src/modchecksim.f90:161-203);
  * periodic-shift equivariance: shifting the start state by (sx, sy) cells shifts the result by the same cells
    (x and y are periodic and every coefficient is uniform in x, y);
  * decomposition: the forced-slab code path (the multi-GPU layout: line FFTs, packed exchange buffers) reproduces the
    single-slab path (2-D rocFFT).
Tolerances: 1e-10 relative to the field maximum (the FFTs are not bit-wise shift invariant), divmax < 1e-10."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, json, numpy as np
sys.path[:0] = ["%(root)s/tests", "%(root)s/u-dales_amd"]
from udcore.core import DynCore
from udcore.grid import Grid
from udcore import lib as L
nx, ny, nz, sgs, nsv, shift, nsub = %(nx)d, %(ny)d, %(nz)d, %(sgs)d, %(nsv)d, %(shift)r, %(nsub)d
g = Grid.uniform(nx, ny, nz)
rng = np.random.default_rng(7)
noise = 0.04 * (rng.random((nz, ny, nx)) - 0.5)
def field(base, roll):
    a = np.zeros(g.mshape())
    a[1:-1, 1:-1, 1:-1] = base + np.roll(noise, roll, axis=(1, 2))
    return a
def run(roll):
    core = DynCore(g, sgs=sgs, nsv=nsv, lbottom=True, z0=0.05)
    core.set_forcing(np.full(nz, -1e-4), np.zeros(nz))
    for k, base, extra in (("u0", 1.0, 0), ("v0", 0.0, 3), ("w0", 0.0, 5)):
        a = field(base, (roll[0] + extra, roll[1] + 2 * extra))
        if k == "w0":
            a[1] = 0.
        core.upload(k, a); core.upload(k.replace("0", "m"), a)
    for n in range(nsv):
        c = np.zeros(g.cshape())
        c[2:-2, 2:-2, 2:-2] = (np.arange(nz)[:, None, None] + 0.5) / nz + np.roll(noise, (roll[0] + 7, roll[1] + 11), axis=(1, 2))
        c[0] = c[2]; c[1] = c[2]; c[-1] = c[-3]; c[-2] = c[-3]
        core.upload(L.scalar_field(L.SV0, n), c); core.upload(L.scalar_field(L.SVM, n), c)
    core.halos(); core.boundary()
    core.run(nsub, 0.25)
    out = {k: core.download(k)[1:-1, 1:-1, 1:-1] for k in ("u0", "w0", "pres0")}
    for n in range(nsv):
        out["sv0"] = core.download(L.scalar_field(L.SV0, n), halo=2)[2:-2, 2:-2, 2:-2]
    out["div"] = core.divergence()[0]
    core.close()
    return out
ref = run((0, 0))
res = {"div": ref["div"], "umax": float(np.abs(ref["u0"]).max()), "wmax": float(np.abs(ref["w0"]).max())}
if shift:
    got = run(shift)
    for k in ref:
        if k == "div":
            continue
        a, b = np.roll(ref[k], shift, axis=(1, 2)), got[k]
        res["shift_" + k] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    res["div_shift"] = got["div"]
np.save("%(out)s", {k: (v if k == "div" else v[::4, ::4, ::4].copy()) for k, v in ref.items()}, allow_pickle=True)
print("RESULT " + json.dumps(res))
'''


def _run(tmp_path, nx, ny, nz, sgs, nsv, shift, nsub, slab, tag):
    out = os.path.join(tmp_path, tag + ".npy")
    code = CODE % dict(root=ROOT, nx=nx, ny=ny, nz=nz, sgs=sgs, nsv=nsv, shift=shift, nsub=nsub, out=out)
    env = dict(os.environ)
    if slab:
        env["UDC_FORCE_SLAB"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-3000:]
    import json
    return json.loads(line[0][7:]), np.load(out, allow_pickle=True).item()


@pytest.mark.parametrize("nx,ny,nz,sgs,nsv,shift", [(512, 512, 256, 1, 1, (37, 130)), (1024, 512, 512, 2, 0, None)])
def test_baseline_size_properties(nx, ny, nz, sgs, nsv, shift, tmp_path):
    res, ref = _run(tmp_path, nx, ny, nz, sgs, nsv, shift, 3, False, "single")
    assert res["div"] < 1e-10 and 0.9 < res["umax"] < 1.2 and res["wmax"] > 1e-4
    if shift:
        assert res["div_shift"] < 1e-10
        for k in ("u0", "w0", "sv0"):
            assert res["shift_" + k] <= 1e-10, (k, res)
        assert res["shift_pres0"] <= 1e-8, res      # (p solves for a round-off-level divergence: compared on its own scale)
    res2, slab = _run(tmp_path, nx, ny, nz, sgs, nsv, None, 3, True, "slab")
    assert res2["div"] < 1e-10
    for k in ref:
        if k == "div":
            continue
        e = np.abs(ref[k] - slab[k]).max() / max(np.abs(ref[k]).max(), 1e-300)
        assert e <= (1e-8 if k == "pres0" else 1e-10), (k, e)
