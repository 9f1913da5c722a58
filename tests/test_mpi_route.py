"""The Fortran / MPI route into the library, as far as one box allows: the reference's real program with the drop-in modules over a
real MPI (u-dales_amd/bin/udales_full_dropin_mpi: every file of the reference's src/ but the ten replaced ones, MPICH, the y-slab
decomposition stand-in), launched as a user launches the reference: `mpiexec -n 2 <exe> namoptions.NNN` with nprocx = 1, nprocy = 2.

Each rank reads the deck (rank 0 reads, MPI_BCAST of every value: src/modstartup.f90:175-520), sets up its slab, creates its library
handle (udc_create with nranks = 2, its rank, its device) and -- u-dales_amd/fortran/udc_iface.f90: rank 0 asks for the RCCL id,
MPI_BCAST carries it over comm3d, every rank calls udc_comm_init -- joins the communicator.
  * Without a GPU (here): both ranks must stop at udc_create with the library's refusal, not hang and not fall back.
  * On a one-GPU box: with both ranks mapped onto the one device (UDC_GPUS_PER_NODE=1) both must get through udc_create and the
    broadcast and into udc_comm_init, where RCCL refuses two ranks on one device; both leave with that error.
  * On a box with a GPU per rank the decks below run over RCCL itself (common.mpi_transport picks the product executable there,
    the shared-memory test transport on a one-GPU box): 2, 4 and 8 ranks against the one-rank fixtures."""
import os
import re
import shutil
import subprocess

import pytest

from common import GOLDEN, RUN_CASES, gpu_count, mpi_transport

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "u-dales_amd", "bin", "udales_full_dropin_mpi")
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
        pytest.skip("u-dales_amd/bin/udales_full_dropin_mpi or MPICH not available")
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
        pytest.skip("u-dales_amd/bin/udales_full_dropin_mpi or MPICH not available")
    if torch.cuda.device_count() >= 2:
        pytest.skip("a one-GPU behaviour (RCCL's refusal of two ranks per device); here test_mpi_ranks_run_the_deck_to_the_end runs over RCCL")
    r = launch(tmp_path, {"UDC_GPUS_PER_NODE": "1"})
    out = r.stdout + r.stderr
    assert r.returncode != 0, out[-2000:]
    # both ranks got their handle and the id; RCCL itself refuses the second rank on the same device
    assert "ERROR: libudcore udc_comm_init" in out, out[-3000:]
    assert "udc_create" not in out, out[-3000:]


# ---- two MPI ranks to the end, on one GPU: the test build of the same program (udales_full_dropin_mpi_test: udc_iface.f90 compiled
# with -DUDC_TEST_TRANSPORT, linked against libudcore_test.so) exchanges through shared memory instead of RCCL (UDC_TEST_SHM)
EXE_TEST = os.path.join(ROOT, "u-dales_amd", "bin", "udales_full_dropin_mpi_test")


def run_ranks(name, iexp, tmp_path, nranks, deck_edit=None):
    """The deck through the test build on `nranks` MPI ranks sharing device 0, stopped at the fixture's last dump -> working directory"""
    import numpy as np      # noqa: F401
    from common import load_fixture
    fix = load_fixture(name)
    tags = sorted(set(k.split(".")[0] for k in fix if re.match(r"s\d\d\d\.", k)))
    last = tags[-1]
    nsub = int(last[1:])
    for fn in os.listdir(os.path.join(GOLDEN, "cases", name)):
        shutil.copy(os.path.join(GOLDEN, "cases", name, fn), tmp_path)
    deck = os.path.join(tmp_path, f"namoptions.{iexp:03d}")
    with open(deck) as f:
        txt = f.read()
    dtmax = float(re.search(r"dtmax\s*=\s*([0-9.eE+-]+)", txt).group(1))
    tend = fix[last + ".time"].data[0] if last + ".time" in fix else dtmax * nsub / 3
    runtime = tend * (1. - 1e-9)
    txt = re.sub(r"runtime\s*=\s*[0-9.eE+-]+", f"runtime = {runtime!r}\ntrestart = {0.999 * runtime!r}", txt)
    txt = re.sub(r"nprocy\s*=\s*\d+", f"nprocy = {nranks}", txt)
    if deck_edit:
        txt = deck_edit(txt)
    with open(deck, "w") as f:
        f.write(txt)
    exe, env, _ = mpi_transport(nranks, "test")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip(f"{exe} or MPICH not available")
    r = subprocess.run(f"ulimit -s unlimited; exec {MPIEXEC} -n {nranks} {exe} namoptions.{iexp:03d}", shell=True, cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=900, executable="/bin/bash")
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    return fix, last


@pytest.mark.gpu
@pytest.mark.parametrize("name,P", [("run_16x16x8", 2), ("run_16x16x8", 4), ("run_smag_scalar_16x8x12s", 2),
                                    ("run_moist_16x8x12s", 2), ("run_ibm_wf2_16x12x10", 2),
                                    ("run_ibm_moistwq_16x12x10", 2)])
def test_mpi_ranks_run_the_deck_to_the_end(name, P, tmp_path):
    """mpiexec -n P (2, 4) of the reference's real program with the drop-in modules, nprocy = P: MPI start-up and broadcasts, P library
    handles with P slabs, the Fortran modules' slab logic (rows of the point lists and facet sections, masks, per-rank restart
    files), ghost rows and the Poisson transposes through the library's multi-rank path -- against the one-rank fixture of the
    all-reference executable, through the restart files the ranks write.  Transport: RCCL where the box has P GPUs (the product
    executable), the shared-memory stand-in on a one-GPU box; a box with 2 <= GPUs < P skips the P-rank case."""
    if 1 < gpu_count() < P:
        pytest.skip(f"{P} ranks need {P} GPUs (or the one-GPU test transport)")
    import numpy as np
    from common import nocorner, relerr
    from udcore import restart
    iexp = RUN_CASES[name]
    fix, last = run_ranks(name, iexp, tmp_path, P)
    nx, ny, nz = (int(v) for v in fix["meta"].data[:3])
    nyl = ny // P
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("initd"))
    assert len(files) == P and [f[14:21] for f in files] == [f"000_{r:03d}" for r in range(P)], files
    parts = [restart.read_initd(str(tmp_path / f), nx, nyl, nz) for f in files]
    checked = 0
    for k in ("u0", "v0", "w0", "pres0", "thl0", "qt0"):
        key = f"{last}.{k}"
        if key not in fix:
            continue
        ref = fix[key].data
        got = np.concatenate([p[k][1:nz + 1, 1:nyl + 1, 1:nx + 1] for p in parts], axis=1)
        assert relerr(got, ref[1:nz + 1, 1:ny + 1, 1:nx + 1], 1.0 if k == "thl0" else None) <= 1e-9, k
        checked += 1
    nsv = int(fix["meta"].data[12])
    if nsv:
        sv = [restart.read_inits(str(tmp_path / f.replace("initd", "inits")), nx, nyl, nz, nsv) for f in files]
        for n in range(nsv):
            got = np.concatenate([s["sv0"][n][1:nz + 1, 1:nyl + 1, 1:nx + 1] for s in sv], axis=1)
            ref = fix[f"{last}.sv0_{n + 1:02d}"].data[2:nz + 2, 2:-2, 2:-2]
            assert relerr(got, ref) <= 1e-9, n
            checked += 1
    assert checked >= 4
    del nocorner


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["run_stats_ibm_16x12x10", "run_ytstats_ibm_16x12x10"])
def test_statistics_tables_on_two_mpi_ranks(name, tmp_path):
    """The drop-in statsdump on two ranks: xytdump / ytdump (rank 0 writes the all-reduced tables), xydump / ydump, and tdump (one
    file per rank, its own rows) equal the one-rank run's."""
    import numpy as np
    from refdump import read_ncrec
    iexp = RUN_CASES[name]
    edit = lambda t: re.sub(r"tstatsdump\s*=\s*1000\.", "tstatsdump = 1.0", t.replace("&OUTPUT", "&OUTPUT\nlxydump = .true.\nlydump = .true."))      # noqa: E731
    out = {}
    for P in (1, 2):
        d = tmp_path / f"p{P}"
        d.mkdir()
        run_ranks(name, iexp, d, P, edit)
        out[P] = {fn: read_ncrec(str(d / fn)) for fn in sorted(os.listdir(d)) if fn.endswith(".nc") and "dump" in fn and "field" not in fn}
    checked = 0
    for fn, ref in out[1].items():
        if fn.startswith("tdump"):      # per-rank files: stitch the rows
            parts = [out[2][fn.replace("000.000", f"000.{r:03d}")] for r in range(2)]
            for var, recs in ref.items():
                if var == "time" or np.squeeze(recs[0][1]).ndim != 3:
                    continue
                for q, (_, a) in enumerate(recs):
                    a = np.squeeze(a)
                    b = np.concatenate([np.squeeze(p[var][q][1]) for p in parts], axis=1)
                    sc = max(np.abs(a).max(), 1e-6)
                    assert np.abs(a - b).max() <= 1e-8 * sc, (fn, var)
                    checked += 1
            continue
        dev = out[2][fn]
        assert list(ref) == list(dev), fn
        for var, recs in ref.items():
            if fn.startswith("ytdump") and ("qt" in var or "sca2" in var or "sca3" in var):
                continue
            for (_, a), (_, b) in zip(recs, dev[var]):
                hole = a < -900.
                assert np.array_equal(hole, b < -900.), (fn, var)
                sc = max(np.abs(a[~hole]).max() if (~hole).any() else 0., 1e-3 if var.startswith("p") else 1e-6)
                assert np.abs(a - b)[~hole].max(initial=0.) <= 1e-8 * sc, (fn, var)
                checked += 1
    assert checked >= 40
