"""Rate of the substep with open x boundaries (BCxm = 2, udc_create_open_x) beside the periodic one, same grid, same box.
   python profiles/tools/open_x_rate.py [n]      # n^3 cells, default 256
Prints ms per substep and the per-kernel profile of the open-x substep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
import numpy as np
from udcore.grid import Grid
from udcore.core import DynCore
from udcore import lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = Grid.uniform(n, n, n, 2.0, 2.0, 2.0) if hasattr(Grid, "uniform") else None
if g is None:
    raise SystemExit("Grid.uniform missing")
rng = np.random.default_rng(7)
res = {}
for mode in ("periodic", "open_x"):
    prof = np.concatenate(([0.], np.full(n, 1.0), [0.]))
    core = DynCore(g, sgs=L.SGS_VREMAN, bctopm=3, lbottom=True, z0=0.05,
                   open_x=(prof, 0.1 * prof) if mode == "open_x" else None)
    core.set_forcing(np.zeros(n), np.zeros(n))
    if mode == "open_x":
        core.set_open_x_outflow(g.dzf[1:n + 1] / (g.zh[n + 1] - g.zh[2]), 1.0)
    sh = (n + 2, n + 2, n + 2)
    for k, m in (("u0", 1.0), ("v0", 0.1), ("w0", 0.0)):
        a = m + 0.05 * (rng.random(sh) - 0.5)
        if k == "w0":
            a[:2] = 0.
        core.upload(k, a); core.upload(k.replace("0", "m"), a)
    core.halos(); core.boundary()
    dt = 0.05
    for s in range(6):
        core.substep(s % 3 + 1, dt)
    core.sync()
    t0 = time.perf_counter()
    K = 30
    for s in range(K):
        core.substep(s % 3 + 1, dt)
    core.sync()
    res[mode] = (time.perf_counter() - t0) / K * 1e3
    print(f"{mode}: {res[mode]:.3f} ms per substep ({n}^3), divergence {core.divergence()[0]:.2e}")
    if mode == "open_x":
        core.profile(True)
        for s in range(9):
            core.substep(s % 3 + 1, dt)
        core.sync()
        for name, (ms, cnt) in sorted(core.profile_get().items(), key=lambda kv: -kv[1][0]):
            print(f"    {name:28s} {ms / 9:.4f} ms per substep ({cnt} launches)")
    core.close()
print(f"open x / periodic = {res['open_x'] / res['periodic']:.3f}")
