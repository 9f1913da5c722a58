"""bench.py under the driver's multi-GPU launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...): the rendezvous / RANK / LOCAL_RANK / WORLD_SIZE
plumbing, the supervisors and their fallback ladder, the unique-id broadcast and udc_comm_init, the line and the field comparison
in it.  On a box with a GPU per rank these run over RCCL itself.  A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks
on one device): there the test library's shared-memory transport carries the same exchanges, and what is asserted of RCCL is that
every rank gets as far as udc_comm_init and leaves cleanly -- no hang, no collective left half-posted -- with the refusal in its
message.  On a box without a GPU the launch must stop
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
    r = launch(2, ["--no-ladder"], 300)
    assert r.returncode != 0
    # (torchrun tears the group down as soon as the first rank has left: the other may not get to print)
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]


def test_ladder_without_gpu_walks_every_rung_and_gives_up():
    """No GPU: every attempt dies at once (bench.py's own message), the supervisors agree on that rung by rung, walk the whole ladder and
    leave with code 5 and no JSON line -- no hang, no rank left behind."""
    if have_gpu():
        pytest.skip("this box has a GPU")
    r = launch(2, ["--stall-timeout", "60"], 600)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    for rung in range(5):
        assert f"rung {rung} (" in r.stderr, r.stderr[-3000:]
    assert "no rung of the ladder ran to its end" in r.stderr
    assert r.stderr.count("bench.py needs an MI355X") >= 5


@pytest.mark.gpu
def test_two_ranks_need_two_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a one-GPU behaviour")
    r = launch(2, ["--no-ladder"], 300)
    assert r.returncode != 0
    assert r.stderr.count("one rank per GPU") >= 2, r.stderr[-2000:]


@pytest.mark.gpu
def test_oversubscribed_ranks_reach_comm_init_and_leave_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a one-GPU behaviour (RCCL's refusal of two ranks per device); here test_bench_line_of_a_multi_rank_run runs over RCCL")
    r = launch(2, ["--oversubscribe", "--no-ladder"], 600)
    assert r.returncode != 0
    assert r.stderr.count("udc_comm_init refused") >= 2, r.stderr[-3000:]
    # the JSON line is only printed by a run that finished
    assert '"metric"' not in r.stdout


def transport_args(n, tag):
    """-> (extra arguments, extra environment, label): RCCL with one rank per GPU where the box has n GPUs, else every rank on
    the one device with the test library's shared-memory transport (--oversubscribe, rendezvous over gloo)."""
    import torch
    if torch.cuda.device_count() >= n:
        return [], {}, "rccl"
    lib = os.path.join(ROOT, "u-dales_amd", "lib", "libudcore_test.so")
    if not os.path.exists(lib):
        pytest.skip("libudcore_test.so not built")
    return ["--oversubscribe"], {"UDC_LIBPATH": lib, "UDC_TEST_SHM": f"/udc_{tag}_{os.getpid()}_{n}"}, "shm"


def bench_line(r):
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_line_of_a_multi_rank_run(n):
    """The N > 1 branch of bench.py run to its JSON line under the driver's own launch line.  On a box with n GPUs: one rank per GPU
    over RCCL, the product library.  On a one-GPU box: the ranks share the device and the test build's inter-process transport
    stands in for RCCL.  Checked: the line's contract fields, the divergence of the run, the one-GPU reference of the same grid in
    the same line -- and that the slabs' ANSWER is the single GPU's: `decomposition_invariance`, u0 v0 w0 pres0 of every cell after
    two RK3 steps from the same cold start, at the reference's own tolerance (processor_boundaries, 1e-9)."""
    import torch
    if 1 < torch.cuda.device_count() < n or (torch.cuda.device_count() < 2 and n == 8):
        pytest.skip(f"{n} ranks need {n} GPUs (the one-GPU stand-in runs 2 and 4)")
    extra, env, how = transport_args(n, "bench")
    size = "64x32x32" if how == "shm" else "256x256x256"
    r = launch(n, extra, 1500, env, size=size, single=True)
    if r.returncode != 0 and os.environ.get("UDC_TEST_KEEP_LOGS"):
        with open(os.path.join(os.environ["UDC_TEST_KEEP_LOGS"], f"bench_launch_{n}_{os.getpid()}.err"), "w") as f:
            f.write(r.stdout + "\n=====\n" + r.stderr)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = bench_line(r)
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "cell-updates/s" and d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["divmax_after_run"] < 1e-10
    assert d["config"]["decomposition"] == f"y-slabs x{n}" and d["config"]["grid"] == [int(v) for v in size.split("x")]
    assert d["cpu_baseline"] is None and d["roofline"]["bound"] == "hbm"
    one = d["single_gpu_same_workload"]
    assert one and "error" not in one, one
    inv = d["decomposition_invariance"]
    assert inv["ok"] and inv["substeps"] == 6 and set(inv["max_rel_diff"]) == {"u0", "v0", "w0", "pres0"}, inv
    assert max(inv["max_rel_diff"].values()) <= 1e-9, inv
    assert d["ladder"]["rung"] == 0 and d["ladder"]["outcome"] == "ok" and len(d["ladder"]["attempts"]) == 1, d["ladder"]
    # the line says what the communicator saw and what the exchanges cost (round 5): RCCL's own rank count where RCCL carried it
    rc, ex, plan = d["rccl"], d["exchange"], d["config"]["executed_plan"]
    assert rc["handle_nranks"] == n and rc["handle_rank"] == 0 and rc["transport"] == ("rccl" if how == "rccl" else "test: shared memory"), rc
    if how == "rccl":
        assert rc["nranks"] == n and rc["rank"] == 0 and rc["version"].count(".") == 2, rc
    assert "error" not in ex, ex
    nx_, ny_, nz_ = (int(v) for v in size.split("x"))
    nch = rc["transpose_k_chunks"]
    cx = -(-(nx_ // 2 + 1) // n)                      # modes per rank of the half spectrum
    # (the backward blocks -- the ones udc_comm_stats reports, being the last -- carry p's two ghost rows behind the slab's own when the
    #  planner put them there: the slab path's own line transforms, i.e. power-of-two rows)
    pg = 2 if plan["p_ghost_row"] == "inside the backward transpose" else 0
    fwd, bwd = 16 * (nz_ // nch) * cx * (ny_ // n), 16 * (nz_ // nch) * cx * (ny_ // n + pg)
    assert ex["alltoall_per_substep"] == 2 * nch and ex["alltoall_bytes_per_peer"] == bwd, (ex, nch, cx)
    assert ex["alltoall_bytes_sent_per_substep"] == nch * (n - 1) * (fwd + bwd)
    assert ex["ghost_row_exchanges_per_substep"] >= (3 if pg else 4) and ex["ghost_row_bytes_to_prev_per_substep"] > 0 and ex["ghost_row_bytes_to_next_per_substep"] > 0
    assert ex["alltoall_GBs_per_link"] > 0 and ex["substep_ms_exchanges_off"] > 0 and "exposed_exchange_ms" in ex
    assert plan["slab_layout"] and plan["transpose_k_chunks"] == nch and plan["p_ghost_row"] != "folded", plan


@pytest.mark.gpu
def test_bench_line_of_a_multi_rank_run_with_a_kappa_scalar():
    """BASELINE configs[2]'s physics (Smagorinsky + a kappa scalar) on two ranks: the scalar's two ghost rows and ekh's travel too."""
    extra, env, how = transport_args(2, "benchsv")
    r = launch(2, extra + ["--nsv", "1", "--sgs", "smag"], 1500, env, size="64x32x32" if how == "shm" else "256x256x256", single=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = bench_line(r)
    assert d["divmax_after_run"] < 1e-10 and d["decomposition_invariance"]["ok"], d["decomposition_invariance"]
    assert any(k.startswith("scalar") for k in d["kernels"]) and d["ladder"]["rung"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["hang", "exit"])
def test_fallback_ladder_of_a_multi_rank_run(mode):
    """A first attempt that hangs (one rank stops answering in the warm-up: the others sit in a collective) or dies must not take the
    line down: the supervisors agree that the attempt is over, kill what is left of it and run the next rung of the ladder; the line
    names the rung that produced the number and lists the failed attempt with its reason."""
    extra, env, how = transport_args(2, f"ladder{mode}")
    env = dict(env, UDC_BENCH_INJECT=f"0:warm-up:{mode}:1")
    r = launch(2, extra + ["--stall-timeout", "12"], 1500, env, size="64x32x32", single=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = bench_line(r)
    lad = d["ladder"]
    assert lad["rung"] == 1 and lad["outcome"] == "ok" and lad["env"]["UDC_MOM_PIPE"] == "0", lad
    assert [a["outcome"] for a in lad["attempts"]] == ["failed", "ok"], lad
    assert any(("no progress" in x) if mode == "hang" else ("exit code 7" in x) for x in lad["attempts"][0]["reasons"]), lad
    assert d["decomposition_invariance"]["ok"] and d["divmax_after_run"] < 1e-10
