"""bench.py --gpus N starts its own ranks: the launcher builds the torch.distributed.run command, runs it as a
fresh child, relays the child's last JSON line as ITS last stdout line and propagates the exit code; with fewer GPUs than N it
refuses within seconds.  No GPU here: the child is a stub."""
import io
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511, python="/usr/bin/python3")
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]          # the ranks get the caller's flags unchanged


def _stub(body):
    return [sys.executable, "-c", body]


def test_relay_prints_the_childs_json_line_last_and_returns_its_code():
    body = ("import sys, json; print('RCCL banner'); print(json.dumps({'metric': 'm', 'value': 1.5, 'n_gpus': 2}));"
            "print('trailing noise from a rank'); sys.stderr.write('warn\\n'); sys.exit(0)")
    out, err = io.StringIO(), io.StringIO()
    rc = bench.relay_child(_stub(body), timeout_s=60, out=out, err=err)
    assert rc == 0
    assert json.loads(out.getvalue().strip().splitlines()[-1]) == {"metric": "m", "value": 1.5, "n_gpus": 2}
    assert out.getvalue().count("\n") == 1                                      # ONLY the JSON line on stdout
    assert "RCCL banner" in err.getvalue() and "trailing noise" in err.getvalue() and "warn" in err.getvalue()


def test_relay_propagates_a_failing_childs_code_and_a_missing_line_is_a_failure():
    out, err = io.StringIO(), io.StringIO()
    assert bench.relay_child(_stub("import sys; print('{not json}'); sys.exit(3)"), timeout_s=60, out=out, err=err) == 3
    assert out.getvalue() == ""
    out, err = io.StringIO(), io.StringIO()
    assert bench.relay_child(_stub("print('no result here')"), timeout_s=60, out=out, err=err) == 1
    assert "without a JSON line" in err.getvalue()


def test_relay_kills_a_hung_child_with_its_process_group():
    body = ("import subprocess, sys, time; subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']);"
            "print('started', flush=True); time.sleep(600)")
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = bench.relay_child(_stub(body), timeout_s=2.0, out=out, err=err)
    assert rc == 124 and time.monotonic() - t0 < 30
    assert "did not finish within" in err.getvalue()


def test_bench_with_more_gpus_than_the_node_has_refuses_within_seconds():
    """This container has no GPU at all (torch.cuda.device_count() == 0): the same refusal a one-GPU box gives `--gpus 2`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env,
                       timeout=120)
    assert p.returncode == 2 and time.monotonic() - t0 < 60
    assert "needs 64 GPUs" in p.stderr and "device_count()" in p.stderr
    assert p.stdout.strip() == ""
