#!/bin/bash
cd /root/repo
for abl in 0 1 2 3; do
DLDKD_K1P_ABL=$abl python - <<'PY' 2>/dev/null
import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "dl-dkd_amd")
import ablation_simpool_ragged as A
g = torch.Generator().manual_seed(2)
lens = torch.randint(A.LEN_LO, A.L + 1, (A.NV,), generator=g)
a = A.time_scorer(lens, A.NQ, 10)
p = A.time_scorer(lens, A.NQ, 10, pairs=True)
a2 = A.time_scorer(lens, A.NQ, 10)
print("abl", os.environ["DLDKD_K1P_ABL"], "A %.3f %.3f  P %.3f  saved %.2f%%" % (a["ms_median"], a2["ms_median"], p["ms_median"], 100 * (1 - 2 * p["ms_median"] / (a["ms_median"] + a2["ms_median"]))))
PY
done
