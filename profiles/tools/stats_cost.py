"""Cost of one statistics sample (udc_stats_sample: tdump accumulators + xytdump profiles) and of one xyt table at a given
grid, per kernel, from the library's own HIP-event markers (udc_profile_enable).  Usage (on a GPU box, from the repo root):
    python profiles/tools/stats_cost.py [NXxNYxNZ] [nsv] > gpurun_out/stats_cost.json
Algorithmic bytes per cell and sample: stats_mom reads um, vm, wm, pres0 (32 B) and reads + writes 19 accumulators (304 B);
stats_scalar reads w, phi, ekh (24 B) and reads + writes 5 accumulators (80 B); stats_xyt (sample) reads um, vm, wm, pres0,
ekm (+ thl, ekh) and one mask byte: 40-57 B; xyt table: 18 accumulators + mask, 145 B."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import bench
    import udcore
    from udcore import cold_start, read_deck
    from udcore.stats import TDump
    size = sys.argv[1] if len(sys.argv) > 1 else "256x256x256"
    nsv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nx, ny, nz = (int(x) for x in size.lower().split("x"))
    with tempfile.TemporaryDirectory() as tmp:
        deck = read_deck(bench.write_deck(tmp, 902, nx, ny, nz, 0, dt=0.25, nsv=nsv, sgs="vreman", floor=True))
    core = udcore.from_deck(deck)
    core.load_state(cold_start(core.g, deck, nsv=nsv))
    core.halos(); core.boundary()
    dt = 0.25
    td = TDump(core, tsample=dt, tstatsdump=1e9, xyt=True)
    for rk in (1, 2, 3):
        core.substep(rk, dt, True)
    td.step(3, dt, dt)                      # first touch
    core.sync()
    core.profile(True)
    core.profile_reset()
    n = 20
    for s in range(n):
        for rk in (1, 2, 3):
            core.substep(rk, dt, True)
        td.step(3, dt, dt * (s + 2))
    core.sync()
    prof = core.profile_get()
    core.profile(False)
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        td.xyt()
    t_table = (time.perf_counter() - t0) / 5
    cells = nx * ny * nz
    bytes_per_cell = {"stats_mom": 32 + 2 * 8 * 19, "stats_scalar": 24 + 2 * 8 * 5, "stats_xyt": 8 * 5 + 1}
    out = {"grid": size, "nsv": nsv, "samples": n, "kernels": {}, "xyt_table_ms_incl_sync_and_copy": round(t_table * 1e3, 4)}
    substep_ms = sum(ms for name, (ms, cnt) in prof.items() if not name.startswith("stats")) / (3 * n)
    for name, (ms, cnt) in sorted(prof.items()):
        if not name.startswith("stats"):
            continue
        e = {"avg_ms": round(ms / cnt, 4), "launches": cnt}
        if name in bytes_per_cell:
            e["algo_bytes_per_cell"] = bytes_per_cell[name]
            e["achieved_GBs"] = round(bytes_per_cell[name] * cells / (ms / cnt * 1e-3) / 1e9, 1)
            e["frac_of_8TBs"] = round(e["achieved_GBs"] / 8000., 3)
        out["kernels"][name] = e
    per_sample = sum(ms for name, (ms, cnt) in prof.items() if name.startswith("stats")) / n
    out["one_sample_ms"] = round(per_sample, 4)
    out["one_substep_ms_marked"] = round(substep_ms, 4)
    out["sample_in_substeps"] = round(per_sample / substep_ms, 2)
    print(json.dumps(out, indent=1))
    core.close()


if __name__ == "__main__":
    main()
