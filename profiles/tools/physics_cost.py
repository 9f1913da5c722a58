"""Cost of the optional physics on top of the neutral dynamical core, one GPU: ms per RK3 substep at 256^3 (floor wall
function on) for (a) neutral, (b) + temperature equation with dry buoyancy, (c) + total water with the moist
thermodynamics (thermo / diagfld / fromztop / calthv every substep).  Prints one JSON line per configuration with the
per-kernel split of the added pieces.  Usage: python profiles/tools/physics_cost.py [n]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "u-dales_amd"))
from udcore.core import DynCore          # noqa: E402
from udcore.grid import Grid             # noqa: E402


def state(g, rng):
    nx, ny, nz = g.nx, g.ny, g.nz

    def wrap(a):
        a[:, 0, :] = a[:, ny, :]; a[:, ny + 1, :] = a[:, 1, :]
        a[:, :, 0] = a[:, :, nx]; a[:, :, nx + 1] = a[:, :, 1]
        return a
    st = {}
    for k, mean in (("u0", 1.0), ("v0", 0.0), ("w0", 0.0)):
        a = np.zeros(g.mshape())
        a[1:-1, 1:-1, 1:-1] = mean + 0.05 * rng.standard_normal((nz, ny, nx))
        st[k] = wrap(a)
        st[k[0] + "m"] = st[k].copy()
    st["pres0"] = np.zeros(g.mshape())
    t = np.zeros(g.mshape())
    t[1:-1, 1:-1, 1:-1] = 288. + 0.01 * g.zf[1:nz + 1, None, None] + 0.05 * rng.standard_normal((nz, ny, nx))
    t = wrap(t); t[0] = t[1]; t[nz + 1] = t[nz]
    q = np.zeros(g.mshape())
    q[1:-1, 1:-1, 1:-1] = 0.0113 - 5e-6 * g.zf[1:nz + 1, None, None] + 2e-4 * rng.standard_normal((nz, ny, nx))
    q = wrap(q); q[nz + 1] = q[nz]
    return st, t, q


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    g = Grid.uniform(n, n, n)
    rng = np.random.default_rng(1)
    st, t, q = state(g, rng)
    for label in ("neutral", "thl+buoyancy", "thl+qt+moist thermodynamics"):
        core = DynCore(g, sgs=2, lbottom=True, z0=0.05)
        s = dict(st)
        if label != "neutral":
            core.set_tempeq(bctopt=1, wttop=0., wtsurf=0.02)
            s.update(thl0=t, thlm=t.copy())
        if label.endswith("thermodynamics"):
            core.set_moisture(bctopq=1, wqtop=0., wqsurf=2e-5)
            core.set_moist_thermo(288., 0.0105)
            s.update(qt0=q, qtm=q.copy())
        if label != "neutral":
            core.set_buoyancy(True)
        core.load_state(s)
        core.set_forcing(np.full(n, -1e-4), np.zeros(n))
        core.halos(); core.boundary()
        dt = 0.02
        core.run(30, dt)
        core.sync()
        t0 = time.perf_counter()
        core.run(90, dt)
        core.sync()
        ms = (time.perf_counter() - t0) / 90 * 1e3
        core.profile(True); core.profile_reset()
        core.run(30, dt)
        core.sync()
        prof = {k: round(v[0] / 30, 4) for k, v in core.profile_get().items()}
        keep = {k: v for k, v in prof.items() if k in ("buoyancy", "thermodynamics", "mom", "closure", "integrate") or k.startswith("scalar")}
        out = dict(config=label, n=n, ms_per_substep=round(ms, 4), cell_updates_per_s=float(f"{n ** 3 / ms * 1e3:.4g}"), kernels_ms=keep)
        if label.endswith("thermodynamics"):
            th = core.thermo_state()
            out["cloud_levels"] = int((th["ql0av"] > 0).sum())
        print(json.dumps(out), flush=True)
        core.close()


if __name__ == "__main__":
    main()
