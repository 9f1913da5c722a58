"""Exploratory parity fuzz (GPU box): random grid shapes / closures / scalars / lids / layouts, three to six fused substeps of the device
against the C oracle on seeded random fields.  Prints every case that exceeds 1e-9 or raises.  Test infrastructure (uses oracle/).
    python tests/fuzz_parity.py [ncases] [seed]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "u-dales_amd")]
import oracle_lib as ol  # noqa: E402
from common import interior, nocorner, relerr  # noqa: E402
from test_gpu_parity import oracle_state, random_state  # noqa: E402
from udcore import lib as L  # noqa: E402
from udcore.grid import Grid  # noqa: E402


def one(rng, idx):
    nx = int(rng.choice([4, 6, 8, 10, 12, 16, 20, 24, 30, 32, 36, 40, 48, 64, 66, 96, 128]))
    ny = int(rng.choice([4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64]))
    nz = int(rng.choice([3, 4, 5, 6, 8, 10, 12, 16, 17, 24, 32, 40]))
    sgs = int(rng.choice([0, 1, 2]))
    nsv = int(rng.choice([0, 0, 1, 2]))
    stretch = float(rng.choice([1.0, 1.0, 1.03, 1.08]))
    floor = bool(rng.integers(0, 2))
    bctopm = int(rng.choice([1, 1, 2, 3]))
    slab = bool(rng.integers(0, 2))
    chunks = int(rng.choice([0, 1, 2, 4]))
    nsub = int(rng.choice([3, 6]))
    env = {}
    if slab:
        env["UDC_FORCE_SLAB"] = "1"
        if chunks and nz % chunks == 0:
            env["UDC_A2A_CHUNKS"] = str(chunks)
    if rng.integers(0, 4) == 0:
        env["UDC_PTOTAL"] = "0"
    if rng.integers(0, 5) == 0:
        env["UDC_P_TRANSPOSE"] = "0"
    if rng.integers(0, 5) == 0:
        env["UDC_MOM_PIPE"] = str(int(rng.choice([0, 1])))
    desc = f"#{idx} {nx}x{ny}x{nz} sgs={sgs} nsv={nsv} stretch={stretch} floor={floor} bctopm={bctopm} nsub={nsub} env={env}"
    for k, v in env.items():
        os.environ[k] = v
    try:
        from udcore.core import DynCore
        dz = 0.5 * stretch ** np.arange(nz)
        zf = np.cumsum(dz) - 0.5 * dz
        g = Grid.from_levels(nx, ny, nz, nx * 0.5, ny * 0.4, zf)
        core = DynCore(g, sgs=sgs, nsv=nsv, lbottom=floor, z0=0.03, bctopm=bctopm, uinf=1.1, vinf=0.1)
        o = ol.Oracle(nx, ny, nz, g.dx, g.dy, g.dzf, g.dzh, sgs=sgs, nsv=nsv, csz=0.21658244510412, lbottom=floor, z0=0.03, bctopm=bctopm,
                      uinf=1.1, vinf=0.1)
        st = random_state(g, seed=idx * 7 + 1, nsv=nsv)
        if bctopm == 2:
            for k, val in (("u0", 1.1), ("v0", 0.1)):
                st[k][-1] = 2 * val - st[k][-2]
            st["um"], st["vm"] = st["u0"].copy(), st["v0"].copy()
        dp = np.zeros(nz + 2); dp[1:nz + 1] = -1e-3
        dq = np.zeros(nz + 2); dq[1:nz + 1] = 2e-4
        core.load_state(st)
        core.set_forcing(dp[1:nz + 1], dq[1:nz + 1])
        ost = oracle_state(st, g, nsv)
        ost["dpdxl"], ost["dpdyl"] = dp, dq
        dt = 0.05
        for s in range(nsub):
            rk = s % 3 + 1
            core.substep(rk, dt, with_forces=True)
            o.substep(ost, rk, dt)
        worst = 0.
        pm = lambda a: a - a[1:-1, 1:-1, 1:-1].mean()      # noqa: E731  (pres0 up to its constant)
        for k in ("u0", "v0", "w0", "um"):
            worst = max(worst, relerr(nocorner(core.download(k)[1:-1]), nocorner(ost[k][1:-1])))
        worst = max(worst, relerr(nocorner(pm(core.download("pres0"))[1:-1, 1:-1, 1:-1]), nocorner(pm(ost["pres0"])[1:-1, 1:-1, 1:-1]),
                                  max(np.abs(pm(ost["pres0"])[1:-1, 1:-1, 1:-1]).max(), 1e-3)))
        if bctopm == 3:
            worst = max(worst, relerr(core.download("w0")[nz + 1, 1:-1, 1:-1], ost["w0"][nz + 1, 1:-1, 1:-1], max(np.abs(ost["w0"]).max(), 1e-6)))
        for n in range(nsv):
            got = core.download(L.scalar_field(L.SV0, n), halo=2)
            worst = max(worst, relerr(interior(got, 2), interior(ost["sv0"][n], 2)))
        divmax, _ = core.divergence()
        plan = core.last_plan()
        core.close()
        bad = worst > 1e-9 or (bctopm != 3 and divmax > 1e-10)
        print(("FAIL " if bad else "ok   ") + desc + f" worst={worst:.2e} div={divmax:.1e}" + (f" plan={plan}" if bad else ""), flush=True)
        return bad
    except Exception:      # noqa: BLE001
        print("EXC  " + desc, flush=True)
        traceback.print_exc()
        return True
    finally:
        for k in env:
            os.environ.pop(k, None)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    nbad = sum(one(rng, i) for i in range(n))
    print(f"{nbad} of {n} cases failed")
