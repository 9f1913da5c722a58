"""The pressure-total form against the reference's form over a long run: the 256^3 bench deck (neutral channel, floor, Vreman; configs[1])
from the same cold start, both forms in turn (UDC_PTOTAL is read in udc_create), fields compared at 300, 3000 and 9000 substeps
(100 / 1000 / 3000 time steps), with max |div u| and the resolved kinetic energy of each.  Run on a GPU box:
    python profiles/tools/ptotal_long.py [nx ny nz]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))
from bench import write_deck      # noqa: E402
import udcore                     # noqa: E402
from udcore import read_deck, cold_start      # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 256)
MARKS = (300, 3000, 9000)


def run(form):
    os.environ["UDC_PTOTAL"] = str(form)
    d = tempfile.mkdtemp()
    path = write_deck(d, 77, nx, ny, nz, MARKS[-1])
    path = path if isinstance(path, str) else os.path.join(d, "namoptions.077")
    dk = read_deck(path)
    core = udcore.from_deck(dk)
    core.load_state(cold_start(core.g, dk, nsv=core.nsv, pre_boundary=True))
    core.start_up()
    dt = float(dk.get("RUN", "dtmax"))
    out, done = {}, 0
    for m in MARKS:
        core.run(m - done, dt, done % 3 + 1, True)
        done = m
        f = {k: core.download(k)[1:-1, 1:-1, 1:-1].copy() for k in ("u0", "v0", "w0", "pres0")}
        divmax, _ = core.divergence()
        ke = 0.5 * float(np.mean(f["u0"] ** 2 + f["v0"] ** 2 + f["w0"] ** 2))
        out[m] = (f, divmax, ke)
    plan = core.last_plan()
    core.close()
    return out, plan


a, pa = run(0)
b, pb = run(1)
print(f"{nx}x{ny}x{nz} bench deck; reference's form: pressure_total_form={pa['pressure_total_form']}; other run: {pb['pressure_total_form']}")
for m in MARKS:
    fa, da, ka = a[m]
    fb, db, kb = b[m]
    errs = {k: float(np.abs(fa[k] - fb[k]).max() / max(np.abs(fa[k]).max(), 1e-300)) for k in fa}
    print(f"after {m:5d} substeps: max rel difference " + "  ".join(f"{k} {e:.1e}" for k, e in errs.items()) +
          f" | max|div u| {da:.1e} / {db:.1e} | kinetic energy {ka:.12f} / {kb:.12f}")
