#!/bin/bash
cd /root/repo
for rep in 1 2; do for prio in 1 0; do
DLDKD_QUERY_PRIO=$prio python - <<'PY' 2>/dev/null
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "dl-dkd_amd")
import bench_train
for c in ("c3", "c5"):
    r = bench_train.run(c, "bf16", 0.2 if c == "c3" else 0.15, steps=40, warmup=10, modes=("graph",))
    print(c, "prio", os.environ["DLDKD_QUERY_PRIO"], round(r["graph"]["stream_ms_median"], 3), round(r["graph_no_loss_sync"]["stream_ms_median"], 3), flush=True)
PY
done; done
