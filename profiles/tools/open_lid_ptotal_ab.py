"""The pressure-total form under the open lid against the reference's form (UDC_PTOTAL=0): six fused substeps at n^3, periodic x and open x,
   max relative difference per field and ms per substep.   python profiles/tools/open_lid_ptotal_ab.py [n]"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
    import numpy as np
    from udcore.grid import Grid
    from udcore.core import DynCore
    from udcore import lib as L
    n = int(sys.argv[1]); mode = sys.argv[3]
    g = Grid.uniform(n, n, n, 2.0, 2.0, 2.0)
    prof = np.concatenate(([0.], 1.0 + 0.2 * np.arange(n) / n, [0.]))
    rng = np.random.default_rng(11)
    sh = g.mshape()
    st = {k: m + 0.05 * (rng.random(sh) - 0.5) for k, m in (("u0", 1.1), ("v0", 0.1), ("w0", 0.))}
    st["w0"][:2] = 0.
    core = DynCore(g, sgs=L.SGS_SMAGORINSKY, bctopm=3, lbottom=True, z0=0.05, open_x=(prof, 0.1 * (prof > 0)) if mode == "open_x" else None)
    core.set_forcing(np.zeros(n), np.zeros(n))
    if mode == "open_x":
        core.set_open_x_outflow(g.dzf[1:n + 1] / (g.zh[n + 1] - g.zh[2]), 1.1)
    for k, a in st.items():
        core.upload(k, a); core.upload(k.replace("0", "m"), a)
    core.halos(); core.boundary()
    dt = 0.05
    nsub = int(os.environ.get("NSUB", "6"))
    for isub in range(1, nsub + 1):
        core.substep((isub - 1) % 3 + 1, dt)
    out = {k: core.download(k) for k in ("u0", "v0", "w0", "pres0")}
    core.sync(); t0 = time.perf_counter()
    for isub in range(1, 31):
        core.substep((isub - 1) % 3 + 1, dt)
    core.sync(); ms = (time.perf_counter() - t0) / 30 * 1e3
    np.savez(sys.argv[4], ms=ms, plan=json.dumps(core.last_plan()), **out)
    sys.exit(0)
import numpy as np, tempfile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for mode in ("periodic", "open_x"):
    res = {}
    for pt in ("1", "0"):
        f = tempfile.mktemp(suffix=".npz")
        subprocess.check_call([sys.executable, __file__, str(n), "child", mode, f], env=dict(os.environ, UDC_PTOTAL=pt))
        res[pt] = np.load(f); os.remove(f)
    a, b = res["1"], res["0"]
    print(f"{mode} {n}^3: pressure-total form {float(a['ms']):.3f} ms, reference's form {float(b['ms']):.3f} ms per substep; ptotal in the plan: {json.loads(str(a['plan']))['pressure_total_form']} / {json.loads(str(b['plan']))['pressure_total_form']}")
    for k in ("u0", "v0", "w0", "pres0"):
        x, y = a[k][1:-1, 1:-1, 1:-1], b[k][1:-1, 1:-1, 1:-1]
        print(f"    {k}: max |diff| / max |ref| = {np.abs(x - y).max() / np.abs(y).max():.2e}   (top plane {np.abs(a[k][-2] - b[k][-2]).max():.2e})")
