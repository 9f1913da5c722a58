"""The multi-rank code path at the REAL per-rank sizes, on one GPU: P virtual ranks (one host thread and one library handle each, the
test library's in-process transport doing the exchanges as device-to-device copies) run the 1024 x 512 x 512 channel of BASELINE
configs[3] as P y-slabs.  All P slabs share the one device, so the wall time per substep is the SUM of the slabs' work (plus the
copies that stand in for xGMI): wall / P is what one rank's kernels need at its real slab size -- the number the >= 6x estimate of
DESIGN.md section 6 rests on -- next to the same grid on the single-slab path.

    python profiles/tools/virtual_ranks.py [P] [nx ny nz] [substeps]      (through gpurun; needs libudcore_test.so)
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
from udcore import lib as L          # noqa: E402
from udcore.core import DynCore      # noqa: E402
from udcore.grid import Grid         # noqa: E402


def run(P, g, nsub, dt=0.25):
    lib = L.load()
    group = lib.udc_local_group_create(P) if P > 1 else 0
    nyl = g.ny // P
    errs, times, tables = [], {}, {}
    start = threading.Barrier(P)

    def worker(r):
        try:
            core = DynCore(g, sgs=L.SGS_VREMAN, rank=r, nranks=P, lbottom=True, z0=0.05)
            core.set_forcing(np.full(g.nz, -1e-4), np.zeros(g.nz))
            if P > 1:
                core.comm_init_local(group)
            rng = np.random.default_rng(100 + r)
            for k, base in (("u0", 1.0), ("v0", 0.0), ("w0", 0.0)):
                a = np.zeros((g.nz + 2, nyl + 2, g.nx + 2))
                a[1:-1, 1:-1, 1:-1] = base + 0.02 * (rng.random((g.nz, nyl, g.nx)) - 0.5)
                if k == "w0":
                    a[1] = 0.
                core.upload(k, a); core.upload(k.replace("0", "m"), a)
            core.halos(); core.boundary()
            for q in range(6):
                core.substep(q % 3 + 1, dt, True)
            core.sync()
            start.wait()
            t0 = time.perf_counter()
            for q in range(nsub):
                core.substep(q % 3 + 1, dt, True)
            core.sync()
            start.wait()
            times[r] = (time.perf_counter() - t0) / nsub * 1e3
            if r == 0:      # the kernels of one slab, every launch marked (while the other slabs run beside it)
                core.profile(True)
                for q in range(6):
                    core.substep(q % 3 + 1, dt, True)
                core.sync()
                tables[0] = {k: round(ms / max(n, 1), 4) for k, (ms, n) in core.profile_get().items()}
                core.profile(False)
            else:
                for q in range(6):
                    core.substep(q % 3 + 1, dt, True)
                core.sync()
            div = core.divergence()[0]
            if r == 0:
                tables["div"] = div
            core.close()
        except Exception as e:      # noqa: BLE001
            errs.append((r, repr(e)))
            try:
                start.abort()
            except Exception:      # noqa: BLE001
                pass

    th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise SystemExit(str(errs))
    return max(times.values()), tables


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nx, ny, nz = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1024, 512, 512)
    nsub = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    g = Grid.uniform(nx, ny, nz)
    with L.test_transport():
        out = {"grid": [nx, ny, nz], "substeps": nsub}
        for p in (1, P):
            ms, tab = run(p, g, nsub)
            out[f"P{p}"] = {"ms_per_substep_all_slabs_on_one_device": round(ms, 3), "ms_per_slab": round(ms / p, 3),
                            "divmax": tab.pop("div", None), "rank0_kernels_avg_ms": tab.get(0)}
        out["ratio_P_over_single"] = round(out[f"P{P}"]["ms_per_substep_all_slabs_on_one_device"] / out["P1"]["ms_per_substep_all_slabs_on_one_device"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
