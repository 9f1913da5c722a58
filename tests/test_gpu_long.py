"""North-star acceptance: u, v, w, p within 1e-6 (relative) of the reference CPU path after
100 full time steps (300 RK3 substeps) on identical namoptions.

The CPU side is the reference's own Fortran (oracle/_ref/udales_ref, built by oracle/Makefile from
/root/reference/src; the binary travels to the GPU box).  Default size 64^3 (BASELINE configs[0]'s
plumbing size) is covered by the fixtures; the default here is 128 x 128 x 32 (~20 s of CPU: the suite has a time limit on the
driver's box), UDC_LONG_SIZE=128 the cube, UDC_LONG_SIZE=256 BASELINE configs[1] itself (~10 min of CPU; the acceptance runs kept
under profiles/).
"""
import os
import subprocess

import numpy as np
import pytest

from common import nocorner, relerr
from refdump import read_dump

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "udales_ref")


def test_100_steps_against_reference_cpu(tmp_path):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/udales_ref not built")
    import sys
    sys.path.insert(0, ROOT)
    from bench import write_deck
    import udcore
    from udcore import read_deck, cold_start
    n = int(os.environ.get("UDC_LONG_SIZE", "128"))
    nzl = n if "UDC_LONG_SIZE" in os.environ else 32
    nsub = 300
    path = write_deck(str(tmp_path), 77, n, n, nzl, nsub)
    with open(path) as f:
        txt = f.read().replace(f"nsub = {nsub}", f"nsub = {nsub}\ndump_at = {nsub}")
    with open(path, "w") as f:
        f.write(txt)
    r = subprocess.run(f"ulimit -s unlimited; exec {REF} namoptions.077 run ref.bin", shell=True, cwd=tmp_path,
                       capture_output=True, text=True, timeout=3000, executable="/bin/bash")
    assert r.returncode == 0, r.stderr[-2000:]
    ref = read_dump(os.path.join(tmp_path, "ref.bin"))
    d = read_deck(path)
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, pre_boundary=True))
    core.start_up()
    core.run(nsub, float(d.get("RUN", "dtmax")), 1, True)
    worst = {}
    for k in ("u0", "v0", "w0", "pres0"):
        a = core.download(k)[1:-1]
        b = ref[f"s{nsub:03d}.{k}"].data[1:-1]
        worst[k] = relerr(nocorner(a), nocorner(b))
    divmax, _ = core.divergence()
    core.close()
    print("100-step parity", n, worst, "divmax", divmax)
    assert max(worst.values()) <= 1e-6, worst
    assert divmax < 1e-10


def test_100_steps_all_physics_against_reference_cpu(tmp_path):
    """The same acceptance with every built physics option switched on at once: temperature and total water with the
    moist thermodynamics (a cloud layer near the floor), buoyancy, the stability-dependent floor wall function (&WALLS iwalltemp = 2:
    with the default of 1 checkinitvalues makes the floor neutral, src/modstartup.f90:811-816), Coriolis,
    large-scale subsidence, nudging, the gravity-wave sponge, a kappa scalar -- 300 RK3 substeps against the
    reference's own Fortran on identical namoptions."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/udales_ref not built")
    import udcore
    from udcore import read_deck, cold_start
    from udcore.forcings import LevelForcings
    n = int(os.environ.get("UDC_LONG_SIZE", "64"))
    nz = 48
    nsub = 300
    dz = 0.5
    with open(tmp_path / "namoptions.078", "w") as f:
        f.write(f"""&RUN
iexpnr = 78
runtime = 1000000.
dtmax = 0.2
ladaptive = .false.
irandom = 43
randu = 0.01
nprocx = 1
nprocy = 1
libm = .false.
/
&DOMAIN
itot = {n}
jtot = {n}
ktot = {nz}
xlen = {n * 0.5}
ylen = {n * 0.5}
/
&PHYSICS
ltempeq = .true.
lbuoyancy = .true.
lmoist = .true.
lcoriol = .true.
lnudge = .true.
tnudge = 60.
nnudge = 30
igrw_damp = 1
/
&DYNAMICS
ipoiss = 0
/
&BC
BCbotm = 2
BCbotT = 2
z0 = 0.05
z0h = 0.005
thls = 289.0
qts = 0.0105
BCtopT = 1
BCtopq = 1
BCbotq = 1
wqsurf = 3.e-5
/
&WALLS
nfcts = 0
lbottom = .true.
iwalltemp = 2
/
&SCALARS
nsv = 1
/
&NAMSUBGRID
lvreman = .true.
/
&ORACLE
nsub = {nsub}
dump_at = {nsub}
/
""")
    with open(tmp_path / "prof.inp.078", "w") as f:
        f.write("# all physics\n# z thl qt u v tke\n")
        for k in range(nz):
            z = (k + 0.5) * dz
            f.write(f"{z:.15f} {288.0 + 0.05 * z!r} {0.0116 - 4e-5 * z!r} 1.0 0.0 0.0\n")
    with open(tmp_path / "scalar.inp.078", "w") as f:
        f.write("# all physics\n# z sv1\n")
        for k in range(nz):
            f.write(f"{(k + 0.5) * dz:.15f} {(k + 0.5) / nz!r}\n")
    with open(tmp_path / "lscale.inp.078", "w") as f:
        f.write("# all physics\n# z uq vq pqx pqy wfls dqtdxls dqtdyls dqtdtls dthlrad\n")
        for k in range(nz):
            z = (k + 0.5) * dz
            f.write(f"{z:.15f} 1.0 0.0 0.0001 0.0 {-0.002 * z / (nz * dz)!r} 0.0 0.0 1.e-8 -1.e-4\n")
    r = subprocess.run(f"ulimit -s unlimited; exec {REF} namoptions.078 run ref.bin", shell=True, cwd=tmp_path,
                       capture_output=True, text=True, timeout=3000, executable="/bin/bash")
    assert r.returncode == 0, r.stderr[-2000:]
    ref = read_dump(os.path.join(tmp_path, "ref.bin"))
    d = read_deck(str(tmp_path / "namoptions.078"))
    core = udcore.from_deck(d)
    core.load_state(cold_start(core.g, d, nsv=1, pre_boundary=True))      # the reference's start-up order: the fields as read,
    ls = LevelForcings(core, d)
    core.start_up(before_boundary=ls.capture_startup)                       # thermodynamics (and diagfld's averages), then boundary
    assert ls.active and ls.subsidence and core.moist_thermo
    dt = float(d.get("RUN", "dtmax"))
    for isub in range(nsub):
        ls.update()
        core.substep(isub % 3 + 1, dt, True)
    assert (core.thermo_state()["ql0av"] > 0).sum() >= 2          # there is a cloud layer
    worst = {}
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0"):
        a = core.download(k)[1:-1]
        b = ref[f"s{nsub:03d}.{k}"].data[1:-1]
        worst[k] = relerr(nocorner(a), nocorner(b), 1.0 if k == "thl0" else None)
    from udcore import lib as L
    sv = core.download(L.scalar_field(L.SV0, 0), halo=2)
    worst["sv0"] = relerr(sv[2:-2, 2:-2, 2:-2], ref[f"s{nsub:03d}.sv0_01"].data[2:-2, 2:-2, 2:-2])
    divmax, _ = core.divergence()
    core.close()
    print("100-step all-physics parity", n, worst, "divmax", divmax)
    assert max(worst.values()) <= 1e-6, worst
    assert divmax < 1e-10
