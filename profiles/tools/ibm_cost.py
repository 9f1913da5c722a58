"""Cost of the immersed boundary on a staggered cube array (the layout of BASELINE configs[4]): sparse corrections (ibmnorm /
solid, diffu/v/w/c_corr) and the facet wall functions for momentum and heat (wallfunmom with the stability functions,
wallfunheat on the facet temperatures) on every fluid-boundary point, from the library's own HIP-event markers.
Usage (on a GPU box, from the repo root):  python profiles/tools/ibm_cost.py [NXxNYxNZ] > gpurun_out/ibm_cost.json
Sections: one per boundary point (an oblique facet normal, so that nothing is skipped and the stress is rotated); every third
takes the velocity at a reconstruction point (trilinear interpolation), the others at the cell."""
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
    size = sys.argv[1] if len(sys.argv) > 1 else "512x512x256"
    nx, ny, nz = (int(x) for x in size.lower().split("x"))
    edge, pitch = 32, 128
    with tempfile.TemporaryDirectory() as tmp:
        deck = read_deck(bench.write_deck(tmp, 903, nx, ny, nz, 0, dt=0.25, nsv=0, sgs="vreman", floor=True))
    deck.nml.setdefault("PHYSICS", {}).update(ltempeq=True, lbuoyancy=True)
    deck.nml.setdefault("RUN", {})["libm"] = False      # the lists are handed over below, not read from files
    core = udcore.from_deck(deck)
    g = core.g
    counts = bench.cube_array_ibm(core, nx, ny, nz, iwallmom=2, heat=True)
    core.load_state(cold_start(g, deck, nsv=0))
    core.halos(); core.boundary()
    dt = 0.05
    for rk in (1, 2, 3):
        core.substep(rk, dt, True)
    core.sync()
    core.profile(True)
    core.profile_reset()
    nsub = 12
    for s in range(nsub):
        core.substep(s % 3 + 1, dt, True)
    core.sync()
    prof = core.profile_get()
    core.profile(False)
    total = sum(ms for ms, _ in prof.values()) / nsub
    out = {"grid": size, "cubes": {"edge": edge, "pitch": pitch, "solid_c_cells": counts["solid_c_cells"]},
           "points": counts["points"], "sections": counts["sections"],
           "substep_ms_marked": round(total, 4), "kernels": {}}
    for name, (ms, cnt) in sorted(prof.items()):
        if name.startswith("ibm"):
            out["kernels"][name] = {"avg_ms": round(ms / cnt, 4), "launches": cnt, "share_of_substep": round(ms / cnt / total, 4)}
    out["divergence"] = core.divergence()[0]
    print(json.dumps(out, indent=1))
    core.close()


if __name__ == "__main__":
    main()
