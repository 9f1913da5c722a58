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
    c = np.zeros((nz + 2, ny, nx), dtype=bool)
    for jb, j0 in enumerate(range(pitch // 4, ny - edge, pitch)):
        for i0 in range(pitch // 4 + (pitch // 2 if jb % 2 else 0), nx - edge, pitch):
            c[1:edge + 1, j0:j0 + edge, i0:i0 + edge] = True
    u = c | np.roll(c, 1, axis=2)
    v = c | np.roll(c, 1, axis=1)
    w = c.copy(); w[1:] |= c[:-1]
    lists = {}
    for name, sol in (("u", u), ("v", v), ("w", w), ("c", c)):
        nb = np.zeros_like(sol)
        for ax, sh in ((2, 1), (2, -1), (1, 1), (1, -1)):
            nb |= np.roll(sol, sh, axis=ax)
        nb[1:] |= sol[:-1]; nb[:-1] |= sol[1:]
        bnd = nb & ~sol

        def pts(m, lo):
            m = m.copy(); m[:lo] = False; m[nz + 1:] = False
            kji = np.argwhere(m)
            return np.ascontiguousarray(np.stack([kji[:, 2] + 1, kji[:, 1] + 1, kji[:, 0]], axis=1), dtype=np.int32)
        lists[name] = (pts(sol, 1), pts(bnd, 2 if name == "w" else 1))
    for q, name in enumerate("uvwc"):
        core.set_ibm_points(q, *lists[name])
    core.ibm_commit()
    dx, dy = g.dx, g.dy
    xh, xf = np.arange(nx + 1) * dx, (np.arange(nx + 1) + 0.5) * dx
    yh, yf = np.arange(ny + 1) * dy, (np.arange(ny + 1) + 0.5) * dy
    zf, zh = g.zf[1:nz + 2], g.zh[1:nz + 2]
    facets = {"norm": np.array([[0.6, 0., 0.8]]), "z0": np.array([0.01]), "z0h": np.array([0.001]), "tsurf": np.array([289.])}
    core.set_ibm_wallfun(2, 0.71, zf, zh)
    nsec = {}
    for q, name in enumerate("uvwc"):
        bnd = lists[name][1]
        n = len(bnd)
        xg, yg, zg = {"u": (xh, yf, zf), "v": (xf, yh, zf), "w": (xf, yf, zh), "c": (xf, yf, zf)}[name]
        i, j, k = bnd[:, 0], bnd[:, 1], bnd[:, 2]
        p = np.stack([xg[i - 1] + 0.3 * dx, yg[j - 1] + 0.2 * dy, zg[k - 1] + 0.3 * (zf[1] - zf[0])], axis=1)
        ids = np.zeros((n, 4, 3), dtype=np.int32)
        for t, (a, b, cc) in enumerate(((xh, yf, zf), (xf, yh, zf), (xf, yf, zh), (xf, yf, zf))):
            ids[:, t, 0] = np.searchsorted(a, p[:, 0], side="right")
            ids[:, t, 1] = np.searchsorted(b, p[:, 1], side="right")
            ids[:, t, 2] = np.searchsorted(cc, p[:, 2], side="right")
        ok = (ids.min(axis=(1, 2)) >= 1) & (ids[:, :, 0].max(axis=1) <= nx) & (ids[:, :, 1].max(axis=1) <= ny) & (ids[:, :, 2].max(axis=1) <= nz)
        rec = ok & (np.arange(n) % 3 == 0)
        S = {"n": n, "cell": bnd, "area": np.full(n, 0.1), "dist": np.full(n, 0.25), "fac": np.ones(n, dtype=np.int32),
             "comprec": np.where(rec, 0, 1).astype(np.int32), "recpt": p, "recids": np.where(rec[:, None, None], ids, 1).astype(np.int32)}
        core.set_ibm_sections(q, S, facets, np.ones((n, 2)))
        nsec[name] = {"sections": int(n), "with_reconstruction": int(rec.sum())}
    core.set_ibm_wallheat(2)
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
    out = {"grid": size, "cubes": {"edge": edge, "pitch": pitch, "solid_c_cells": int(c.sum())},
           "points": {n: {"solid": int(len(lists[n][0])), "boundary": int(len(lists[n][1]))} for n in "uvwc"}, "sections": nsec,
           "substep_ms_marked": round(total, 4), "kernels": {}}
    for name, (ms, cnt) in sorted(prof.items()):
        if name.startswith("ibm"):
            out["kernels"][name] = {"avg_ms": round(ms / cnt, 4), "launches": cnt, "share_of_substep": round(ms / cnt / total, 4)}
    out["divergence"] = core.divergence()[0]
    print(json.dumps(out, indent=1))
    core.close()


if __name__ == "__main__":
    main()
