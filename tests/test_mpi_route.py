"""The Fortran / MPI route into the library, as far as one box allows: the reference's real program with the drop-in modules over a
real MPI (oracle/_ref/udales_full_dropin_mpi: every file of the reference's src/ but the ten replaced ones, MPICH, the y-slab
decomposition stand-in), launched as a user launches the reference: `mpiexec -n 2 <exe> namoptions.NNN` with nprocx = 1, nprocy = 2.

Each rank reads the deck (rank 0 reads, MPI_BCAST of every value: src/modstartup.f90:175-520), sets up its slab, creates its library
handle (udc_create with nranks = 2, its rank, its device) and -- u-dales_amd/fortran/udc_iface.f90: rank 0 asks for the RCCL id,
MPI_BCAST carries it over comm3d, every rank calls udc_comm_init -- joins the communicator.
  * Without a GPU (here): both ranks must stop at udc_create with the library's refusal, not hang and not fall back.
  * On a one-GPU box: with both ranks mapped onto the one device (UDC_GPUS_PER_NODE=1) both must get through udc_create and the
    broadcast and into udc_comm_init, where RCCL refuses two ranks on one device; both leave with that error.
The real two-GPU run is the driver's scaling bench."""
import os
import re
import shutil
import subprocess

import pytest

from common import GOLDEN, RUN_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "udales_full_dropin_mpi")
MPIEXEC = "/opt/conda/bin/mpiexec"


def launch(tmp_path, env_extra):
    name, iexp = "run_16x16x8", RUN_CASES["run_16x16x8"]
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    deck = os.path.join(tmp_path, f"namoptions.{iexp:03d}")
    with open(deck) as f:
        txt = f.read()
    with open(deck, "w") as f:
        f.write(re.sub(r"nprocy\s*=\s*\d+", "nprocy = 2", txt))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", UDC_RESIDENCY="2", **env_extra)
    return subprocess.run(f"ulimit -s unlimited; exec {MPIEXEC} -n 2 {EXE} namoptions.{iexp:03d}", shell=True, cwd=tmp_path, env=env,
                          capture_output=True, text=True, timeout=600, executable="/bin/bash")


def have_gpu():
    import torch
    return torch.cuda.is_available()


def test_two_mpi_ranks_without_a_gpu_stop_at_udc_create(tmp_path):
    if not (os.path.exists(EXE) and os.path.exists(MPIEXEC)):
        pytest.skip("oracle/_ref/udales_full_dropin_mpi or MPICH not available")
    if have_gpu():
        pytest.skip("this box has a GPU")
    r = launch(tmp_path, {})
    out = r.stdout + r.stderr
    assert r.returncode != 0, out[-2000:]
    assert out.count("ERROR: libudcore udc_create") >= 1 and "no HIP device" in out, out[-2000:]


@pytest.mark.gpu
def test_two_mpi_ranks_on_one_gpu_reach_udc_comm_init(tmp_path):
    import torch
    if not (os.path.exists(EXE) and os.path.exists(MPIEXEC)):
        pytest.skip("oracle/_ref/udales_full_dropin_mpi or MPICH not available")
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU here: the real N = 2 run is the driver's scaling bench")
    r = launch(tmp_path, {"UDC_GPUS_PER_NODE": "1"})
    out = r.stdout + r.stderr
    assert r.returncode != 0, out[-2000:]
    # both ranks got their handle and the id; RCCL itself refuses the second rank on the same device
    assert "ERROR: libudcore udc_comm_init" in out, out[-3000:]
    assert "udc_create" not in out, out[-3000:]
