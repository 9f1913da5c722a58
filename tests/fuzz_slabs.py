"""Exploratory decomposition fuzz (GPU box): random grids on 2 / 4 / 8 virtual ranks (one process, one GPU, the test library's in-process
transport) against the same run on one rank -- six substeps, 1e-10.  Test infrastructure.
    python tests/fuzz_slabs.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "u-dales_amd")]
from common import relerr  # noqa: E402
from test_gpu_parity import random_state  # noqa: E402
from test_gpu_slabs import run_virtual  # noqa: E402
from udcore.grid import Grid  # noqa: E402


def one(rng, idx):
    nx = int(rng.choice([16, 24, 32, 40, 64, 128]))
    ny = int(rng.choice([16, 32, 48, 64, 128]))
    nz = int(rng.choice([8, 12, 16, 20, 32]))
    sgs = int(rng.choice([1, 2]))
    extras = int(rng.choice([0, 0, 1, 2]))
    chunks = int(rng.choice([1, 2, 4]))
    if nz % chunks:
        chunks = 1
    env = {"UDC_A2A_CHUNKS": str(chunks)}
    for k, vals, p in (("UDC_PTOTAL", ["0"], 4), ("UDC_P_TRANSPOSE", ["0"], 4), ("UDC_MOM_PIPE", ["0", "1"], 4), ("UDC_HALO_OVERLAP", ["0"], 5)):
        if rng.integers(0, p) == 0:
            env[k] = str(rng.choice(vals))
    desc = f"#{idx} {nx}x{ny}x{nz} sgs={sgs} extras={extras} env={env}"
    for k, v in env.items():
        os.environ[k] = v
    try:
        g = Grid.uniform(nx, ny, nz)
        st = random_state(g, seed=idx + 11)
        if extras:
            r2 = np.random.default_rng(5)
            t = np.zeros(g.mshape())
            t[1:-1, 1:-1, 1:-1] = 288. + 0.2 * g.zf[1:nz + 1, None, None] + 0.05 * r2.standard_normal((nz, ny, nx))
            t[:, 0, :] = t[:, ny, :]; t[:, ny + 1, :] = t[:, 1, :]
            t[:, :, 0] = t[:, :, nx]; t[:, :, nx + 1] = t[:, :, 1]
            t[0] = t[1]; t[nz + 1] = 2 * 291. - t[nz]
            st["thl0"], st["thlm"] = t, t.copy()
            if extras == 2:
                q = np.zeros(g.mshape())
                q[1:-1, 1:-1, 1:-1] = 0.0118 - 8e-5 * g.zf[1:nz + 1, None, None] + 2e-4 * r2.standard_normal((nz, ny, nx))
                q[:, 0, :] = q[:, ny, :]; q[:, ny + 1, :] = q[:, 1, :]
                q[:, :, 0] = q[:, :, nx]; q[:, :, nx + 1] = q[:, :, 1]
                q[nz + 1] = 2 * 0.0105 - q[nz]
                st["qt0"], st["qtm"] = q, q.copy()
        ref = run_virtual(1, g, None, st, 6, 0.05, sgs, extras=extras)
        worst = 0.
        for P in [p for p in (2, 4, 8) if ny % p == 0 and ny // p >= 4]:
            got = run_virtual(P, g, None, st, 6, 0.05, sgs, extras=extras)
            for k in ("u0", "v0", "w0", "pres0") + (("thl0",) if extras else ()) + (("qt0",) if extras == 2 else ()):
                worst = max(worst, relerr(got[k][1:-1], ref[k][1:-1], 1.0 if k == "thl0" else None))
        bad = worst > 1e-10
        print(("FAIL " if bad else "ok   ") + desc + f" worst={worst:.2e}", flush=True)
        return bad
    except Exception as e:      # noqa: BLE001
        print("EXC  " + desc + " " + repr(e)[:300], flush=True)
        return True
    finally:
        for k in env:
            os.environ.pop(k, None)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    nbad = sum(one(rng, i) for i in range(n))
    print(f"{nbad} of {n} cases failed")
