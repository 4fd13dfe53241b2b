import sys, types, torch
sys.path[:0] = ["/root/repo/dl-dkd_amd", "/root/repo/tests/golden", "/root/repo/tools"]
import bench_train as B
from dldkd_amd import ops, train as T, functional as F_
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ops.set_gemm_precision(prec)
for rep in range(2):
    from dldkd_amd.model import DLDKD
    for skip, tower in ((False, False), (True, False), (True, True)):
        F_.IN_PROJ_SKIP_PADDING = skip
        DLDKD.TOWER_SKIPS_PADDING = tower
        m, opt, batch = B.build("c3", 0.2, "cuda:0")
        g = T.GraphedTrainStep(m, opt, types.SimpleNamespace(grad_clip=-1), defer_loss_float=True)
        r = B.timed(lambda: g(batch), 30, 10)
        print(prec, "in_proj_skip", skip, "tower_skip", tower, "stream_ms_median %.3f" % r["stream_ms_median"])
