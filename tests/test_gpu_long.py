"""North-star acceptance: u, v, w, p within 1e-6 (relative) of the reference CPU path after
100 full time steps (300 RK3 substeps) on identical namoptions.

The CPU side is the reference's own Fortran (oracle/_ref/udales_ref, built by oracle/Makefile from
/root/reference/src; the binary travels to the GPU box).  Default size 64^3 (BASELINE configs[0]'s
plumbing size, ~30 s of CPU); UDC_LONG_SIZE=128 or 256 runs the larger cases (minutes of CPU).
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
    n = int(os.environ.get("UDC_LONG_SIZE", "64"))
    nsub = 300
    path = write_deck(str(tmp_path), 77, n, n, n, nsub)
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
    core.load_state(cold_start(core.g, d))
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
