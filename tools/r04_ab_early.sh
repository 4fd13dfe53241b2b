#!/bin/bash
cd /root/repo
python - <<'PY' 2>/dev/null
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "dl-dkd_amd")
import bench_train
from dldkd_amd import train as T
for rep in range(2):
    for early in (True, False):
        T.GraphedTrainStep.EARLY_VIDEO_START = early
        for c in ("c3", "c5"):
            r = bench_train.run(c, "bf16", 0.2 if c == "c3" else 0.15, steps=40, warmup=10, modes=("graph",))
            print(c, "early" if early else "late ", round(r["graph"]["stream_ms_median"], 3), round(r["graph_no_loss_sync"]["stream_ms_median"], 3), flush=True)
PY
