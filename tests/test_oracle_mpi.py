"""The all-cores CPU baseline (oracle/_ref/udales_ref_mpi: reference Fortran + MPICH + the y-slab
decomposition shim oracle/shims/decomp_2d_mpi.f90) must be decomposition invariant, like the
reference's own processor_boundaries test demands of 2decomp-fft
(tests/integration/processor_boundaries/test_processor_boundaries.py:28-34)."""
import os
import re
import shutil
import subprocess

import pytest

from common import GOLDEN, RUN_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "udales_ref")
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "udales_ref_mpi")
MPIEXEC = "/opt/conda/bin/mpiexec"


def run(cmd, cwd):
    r = subprocess.run(f"ulimit -s unlimited; exec {cmd}", shell=True, cwd=cwd, capture_output=True, text=True,
                       timeout=300, executable="/bin/bash")
    m = re.search(r"sum_u0sq=\s*([0-9.Ee+-]+)\s+divmax=\s*([0-9.Ee+-]+)", r.stdout)
    assert m, r.stdout[-1000:] + r.stderr[-1000:]
    return float(m.group(1)), float(m.group(2))


def test_mpi_baseline_is_decomposition_invariant(tmp_path):
    if not (os.path.exists(REF) and os.path.exists(REF_MPI) and os.path.exists(MPIEXEC)):
        pytest.skip("reference CPU builds or MPICH not available here")
    name, iexp = "run_16x16x8", RUN_CASES["run_16x16x8"]
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    deck = f"namoptions.{iexp:03d}"
    s1, d1 = run(f"{REF} {deck} time x.bin", tmp_path)
    for p in (2, 4, 8):
        sp, dp = run(f"{MPIEXEC} -n {p} {REF_MPI} {deck} time x.bin", tmp_path)
        assert abs(sp - s1) <= 1e-11 * abs(s1), (p, sp, s1)
        assert dp < 1e-12
