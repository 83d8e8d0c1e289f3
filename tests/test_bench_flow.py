"""bench.py's control flow under N > 1, on a GPU-less box: the script is launched exactly like the driver launches it
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...`) with `--dry-run-emu`, which swaps the
device for the CPU kernel emulator and RCCL for gloo and changes nothing else: warm-up, timed region, the exclusive
profiling pass, the alt-math loop, every barrier and every collective run on both ranks.  A rank-dependent branch that
issues a collective (the exclusive pass once ran on rank 0 only) shows up here as a hang or a gloo size mismatch."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, extra=(), plain=False, emu_threads=4, timeout=900):
    env = dict(os.environ, UP_EMU_THREADS=str(emu_threads), OMP_NUM_THREADS="2" if nproc <= 2 else "1")
    # the emulator needs ~10-20 s per training step of the full ResNet-101: no warm-up, one timed step (plus the two of the
    # exclusive pass), no alt-math loop (that loop has no rank-dependent branch)
    args = ["--gpus", str(nproc), "--steps", "1", "--warmup", "0", "--batch", "2", "--size", "32", "--dry-run-emu",
            "--no-alt-math", *extra]     # (a later --steps in `extra` wins)
    if nproc == 1 or plain:           # plain: no launcher, bench.py starts its own ranks (bench.self_launch)
        env.pop("WORLD_SIZE", None)
        cmd = [sys.executable, "bench.py", *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", *args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_flow_two_ranks():
    out = _run(2, ["--steps", "2"])     # step 1: unbucketed exchange, step 2: flat-buffer exchange
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 0
    dp = out["data_parallel"]
    assert out["rccl_ranks"] == 2 and dp["comm_ranks"] == 2          # an all-reduce of ones really spans two ranks
    assert len(dp["ms_per_step_by_rank"]) == 2 and len(dp["exchange_ms_by_rank"]) == 2
    assert dp["weights_identical_across_ranks"] is True              # replicas agree after averaged-gradient Adam steps
    assert dp["payload_mb"] > 100                                     # the 190 MB of live gradients went through the buckets
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert "dry_run" in out and out["metric"].startswith("DRY RUN")
    assert "cpu_baseline" not in out                   # rank 0 at N = 1 only


def test_plain_invocation_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the shape of the driver's 1-GPU command with another N): the script
    launches its own two ranks and still prints exactly one JSON line, from rank 0, spanning two ranks."""
    out = _run(2, ["--no-profile"], plain=True)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["data_parallel"]["comm_ranks"] == 2
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["data_parallel"]["weights_identical_across_ranks"] is True


def test_bench_flow_single_rank_contract():
    out = _run(1, ["--no-profile", "--cpu-steps", "1", "--cpu-batch", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "cpu_baseline"):
        assert key in out, key
    assert out["warmup"] == 0 and out["n_gpus"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] == "port"


def test_bench_flow_four_ranks_overlapped_buckets():
    """VERDICT r5 pre-flight: four ranks, the hook-driven bucketed exchange (--overlap): step 1 is the calibration step (plain
    exchange, hand-out order recorded), then rank 0's bucket order is broadcast and the collective verdict on the parameter set is
    taken, steps 2-3 run the overlapped buckets.  A rank whose order or verdict differed would hang or mismatch here."""
    out = _run(4, ["--steps", "3", "--overlap", "--no-profile"], emu_threads=2, timeout=1500)
    dp = out["data_parallel"]
    assert out["n_gpus"] == 4 and out["rccl_ranks"] == 4 and dp["comm_ranks"] == 4
    assert dp["exchange"] == "overlap" and len(dp["ms_per_step_by_rank"]) == 4
    assert dp["weights_identical_across_ranks"] is True
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp4"


def test_bench_flow_eight_ranks():
    """... and the driver's largest launch, eight ranks (one step, flat exchange): one JSON line from rank 0, an all-reduce of ones
    spans eight ranks, the replicas agree after the averaged-gradient step."""
    out = _run(8, ["--no-profile"], emu_threads=1, timeout=1800)
    dp = out["data_parallel"]
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and dp["comm_ranks"] == 8
    assert len(dp["ms_per_step_by_rank"]) == 8 and dp["weights_identical_across_ranks"] is True
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp8"
