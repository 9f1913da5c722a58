"""The command-line runner (u-dales_amd/udcore/run.py): deck checks on the CPU, an end-to-end run on the GPU whose
restart files are compared with the reference's dumps."""
import os
import shutil
import subprocess
import sys

import pytest

from common import GOLDEN, RUN_CASES, deck_path, load_fixture, marr, nocorner, relerr
from udcore import read_deck
from udcore import restart as R
from udcore.run import check_supported, courant_default

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unsupported_decks_are_refused(tmp_path):
    src = os.path.join(GOLDEN, "cases", "run_16x16x8")
    for fn in os.listdir(src):
        shutil.copy(os.path.join(src, fn), tmp_path)
    path = os.path.join(tmp_path, "namoptions.021")
    check_supported(read_deck(path))                       # the fixture deck itself is fine
    txt = open(path).read()
    for bad in ("&BC\nBCzp = 2", "&BC\nBCxm = 3"):
        grp = bad.split("\n")[0]
        with open(path, "w") as f:
            f.write(txt.replace(grp, bad, 1))
        with pytest.raises(SystemExit):
            check_supported(read_deck(path))
    with open(path, "w") as f:
        f.write(txt.replace("libm = .false.", "libm = .true.").replace("&ORACLE", "&WALLS\nnfcts = 12\n/\n&ORACLE"))
    with pytest.raises(SystemExit):
        check_supported(read_deck(path))


def test_courant_default():
    d = read_deck(deck_path("run_16x16x8", 21))
    assert courant_default(d) == 1.5                       # cd2 only, src/modglobal.f90:565-567
    d = read_deck(deck_path("run_smag_scalar_16x8x12s", 22))
    assert courant_default(d) == 1.1                       # kappa scalars, :570-571


@pytest.mark.gpu
def test_cli_run_writes_the_reference_state(tmp_path):
    name, iexp = "run_16x16x8", RUN_CASES["run_16x16x8"]
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_case.py"), f"namoptions.{iexp:03d}", "--steps", "3"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "3 steps" in r.stdout
    fix = load_fixture(name)
    d = read_deck(deck_path(name, iexp))
    n = int(d.get("DOMAIN", "itot")), int(d.get("DOMAIN", "jtot")), int(d.get("DOMAIN", "ktot"))
    got = R.read_initd(os.path.join(tmp_path, R.restart_name(3, 0, iexp)), *n)
    assert abs(got["timee"] - 3 * float(d.get("RUN", "dtmax"))) < 1e-12
    for k in ("u0", "v0", "w0", "pres0"):                  # 3 steps = 9 substeps = the fixture's last dump
        ref = marr(fix, f"s009.{k}", n[2])
        assert relerr(nocorner(got[k][1:-1]), nocorner(ref[1:-1])) <= 1e-9, k
