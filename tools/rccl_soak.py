"""Soak of bench.py's distributed path on one GPU (DLDKD_BENCH_FORCE_DIST=1: world size 1 through the same collectives code as
--gpus N).  Runs it N times, each in a fresh child, and records exit code, wall time and every line of the child's stderr that
names a HIP / RCCL error - the head of the message, which a tail of the stack loses.

    python tools/rccl_soak.py --runs 30 --out profiles/r05/rccl_soak.json [-- extra bench.py arguments]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"hipError|HIP error|NCCL error|RCCL|ncclUnhandled|ncclSystem|ncclInternal|terminate called|what\(\)|Exception raised|"
                 r"Segmentation|Aborted|core dumped|Traceback|Error:", re.I)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=30)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rccl_soak.json"))
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("rest", nargs="*")
    a = ap.parse_args()
    args = a.rest or ["--steps", "2", "--warmup", "1"]
    runs, aborts = [], 0
    for i in range(a.runs):
        env = dict(os.environ, DLDKD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + i % 50))
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                               timeout=a.timeout, cwd=ROOT)
            rc, err, out = r.returncode, r.stderr, r.stdout
        except subprocess.TimeoutExpired as ex:
            rc, err, out = "timeout", (ex.stderr or b"").decode("utf8", "replace") if isinstance(ex.stderr, bytes) else (ex.stderr or ""), ""
        dt = time.time() - t0
        ok = rc == 0
        line = None
        if ok:
            try:
                line = json.loads(out.strip().splitlines()[-1])
                ok = line.get("assembled_max_abs_diff") == 0.0 and line.get("recall_matches_n1") is True
            except Exception:   # noqa: BLE001
                ok = False
        aborts += (not ok)
        hits = [ln[:400] for ln in err.splitlines() if PAT.search(ln)][:40]
        runs.append({"run": i, "rc": rc, "ok": ok, "wall_s": round(dt, 1), "error_lines": hits if not ok else [],
                     "stderr_head": err[:3000] if not ok else ""})
        print(f"run {i}: rc {rc} ok {ok} {dt:.1f}s" + ("" if ok else "\n  " + "\n  ".join(hits[:12])), flush=True)
    res = {"command": "DLDKD_BENCH_FORCE_DIST=1 python bench.py " + " ".join(args), "runs": a.runs, "aborts": aborts, "detail": runs}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(f"soak: {aborts} / {a.runs} failed")


if __name__ == "__main__":
    main()
