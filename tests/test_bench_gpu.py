"""bench.py's multi-rank path checks itself: run it through the one-rank RCCL hook and read the line.
Collected LAST (conftest.pytest_collection_modifyitems): these tests run bench.py as a child process and must never stand between
the driver's `pytest -x` and the parity tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(env_extra, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=1500,
                       cwd=ROOT)
    # (head AND tail of the child's stderr: the HIP / RCCL error string is at the head of an abort message, the stack at its tail)
    assert r.returncode == 0, f"bench.py exit {r.returncode}\n--- stderr head\n{r.stderr[:3000]}\n--- stderr tail\n{r.stderr[-2000:]}"
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_distributed_path_verifies_itself_on_one_rank():
    """DLDKD_BENCH_FORCE_DIST=1: OverlappedShardScorer + the RCCL communicator of the C ABI with world size 1.  The gathered matrix equals a plain
    one-launch recompute of sampled videos bit for bit, its recalls are the one-GPU recalls (same gallery, same queries for
    every rank count), and the C4 workload goes through the same path."""
    import bench
    d = _line({"DLDKD_BENCH_FORCE_DIST": "1"}, "--steps", "2", "--warmup", "1")
    assert d["assembled_max_abs_diff"] == 0.0
    assert d["recall_hip"] == bench.RECALL_N1 and d["recall_matches_n1"] is True
    assert 15.0 > d["recall_hip"]["R@1"] > 5.0                        # planted signal, not chance (1 / 21,793)
    c4 = d["extras"]["c4_sharded"]
    assert c4["assembled_max_abs_diff"] == 0.0 and c4["recall_hip"] == bench.RECALL_C4_N1 and c4["recall_matches_n1"] is True
    assert c4["n_queries"] == 17505 and c4["n_videos"] == 4917 and c4["n_ranges"] >= 4
    # BASELINE configs[4]: the data-parallel Charades step through the same hook (RCCL group of one rank)
    c5 = d["extras"]["c5_ddp"]
    assert c5["grad_bytes"] == 4371968 * 4 + 0 or c5["grad_bytes"] >= 4371968 * 4          # 17.5 MB of fp32 gradients (+ chunk padding)
    for layout, nb in (("single_bucket", 1), ("tower_buckets", 4)):
        r = c5[layout]
        assert r["n_buckets"] == nb and r["replicas_identical"] is True and r["replicas_max_abs_param_diff_after_timed_steps"] == 0.0
        assert 0.2 < r["step_ms"] < 50.0 and 0.2 < r["step_ms_without_allreduce"] < 50.0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d


def test_bench_plain_run_prints_the_same_recalls():
    import bench
    d = _line({}, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras")
    assert d["recall_hip"] == bench.RECALL_N1 and d["n_gpus"] == 1 and d["roofline"]["bound"] == "mfma"


def test_bench_asking_for_more_gpus_than_the_box_has_exits_fast_with_the_reason():
    """`python bench.py --gpus 2` (no WORLD_SIZE) used to print a usage message; now it starts its own ranks - and
    on a box with fewer GPUs it must say so and exit non-zero within seconds, never hang a lease."""
    import time
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2))], env=env, capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert r.returncode == 2 and time.monotonic() - t0 < 60.0
    assert f"needs {max(n, 2)} GPUs" in r.stderr and r.stdout.strip() == ""


def test_self_launched_ranks_relay_their_json_line():
    """The launcher end to end on what this box has: bench.launch_command for ONE rank under torch.distributed.run (the agent's
    rendezvous on 127.0.0.1, RANK / LOCAL_RANK / WORLD_SIZE from it) with the distributed code path forced, relayed by
    bench.relay_child: the parent's stdout is exactly rank 0's JSON line."""
    import io
    import bench
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, DLDKD_BENCH_FORCE_DIST="1",
               DLDKD_COMM_DEADLINE_S="120")
    out, err = io.StringIO(), io.StringIO()
    cmd = bench.launch_command(1, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras"], bench.free_port())
    rc = bench.relay_child(cmd, timeout_s=1200, env=env, out=out, err=err)
    assert rc == 0, err.getvalue()[-3000:]
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["recall_hip"] == bench.RECALL_N1 and d["assembled_max_abs_diff"] == 0.0
