"""bench.py under the driver's multi-GPU launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...): the rendezvous / RANK / LOCAL_RANK / WORLD_SIZE
plumbing, the unique-id broadcast and udc_comm_init.  A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks
on one device), so what is asserted there is that every rank gets as far as udc_comm_init and leaves cleanly -- no
hang, no collective left half-posted -- with the refusal in its message.  On a box without a GPU the launch must stop
with bench.py's own message on every rank."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(n, extra, timeout, env_extra=None, size="32x16x16", single=False):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--no-cpu", "--no-pmc", "--no-dropin", "--size", size] + ([] if single else ["--no-single"]) + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def have_gpu():
    import torch
    return torch.cuda.is_available()


def test_launch_without_gpu_stops_on_every_rank():
    if have_gpu():
        pytest.skip("this box has a GPU")
    r = launch(2, [], 300)
    assert r.returncode != 0
    # (torchrun tears the group down as soon as the first rank has left: the other may not get to print)
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]


@pytest.mark.gpu
def test_two_ranks_need_two_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU here")
    r = launch(2, [], 300)
    assert r.returncode != 0
    assert r.stderr.count("one rank per GPU") >= 2, r.stderr[-2000:]


@pytest.mark.gpu
def test_oversubscribed_ranks_reach_comm_init_and_leave_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU here: the real N = 2 run is the driver's scaling bench")
    r = launch(2, ["--oversubscribe"], 600)
    assert r.returncode != 0
    assert r.stderr.count("udc_comm_init refused") >= 2, r.stderr[-3000:]
    # the JSON line is only printed by a run that finished
    assert '"metric"' not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
def test_bench_line_of_a_multi_rank_run(n):
    """The N > 1 branch of bench.py run to its JSON line on ONE GPU: the ranks share the device (--oversubscribe, rendezvous over gloo)
    and the library is the test build, whose inter-process transport (UDC_TEST_SHM, shared memory) stands in for RCCL.  Checked: the
    line's contract fields, the divergence of the run, the one-GPU reference of the same grid in the same line, and that the slabs'
    answer is the single GPU's (poisson-only and substep times are both there)."""
    import json
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU here: the real run is the driver's scaling bench")
    lib = os.path.join(ROOT, "u-dales_amd", "lib", "libudcore_test.so")
    if not os.path.exists(lib):
        pytest.skip("libudcore_test.so not built")
    r = launch(n, ["--oversubscribe"], 900, {"UDC_LIBPATH": lib, "UDC_TEST_SHM": f"/udc_bench_{os.getpid()}_{n}"}, size="64x32x32", single=True)
    if r.returncode != 0 and os.environ.get("UDC_TEST_KEEP_LOGS"):
        with open(os.path.join(os.environ["UDC_TEST_KEEP_LOGS"], f"bench_launch_{n}_{os.getpid()}.err"), "w") as f:
            f.write(r.stdout + "\n=====\n" + r.stderr)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "cell-updates/s" and d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["divmax_after_run"] < 1e-10
    assert d["config"]["decomposition"] == f"y-slabs x{n}" and d["config"]["grid"] == [64, 32, 32]
    assert d["cpu_baseline"] is None and d["roofline"]["bound"] == "hbm"
    one = d["single_gpu_same_workload"]
    assert one and "error" not in one, one
