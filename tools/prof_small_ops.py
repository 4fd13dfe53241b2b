"""Which Python lines of the training step launch the small ATen kernels (fills, copies, adds)?  One eager C3 / C5 step under
torch.profiler with stacks; prints every aten op that launched a GPU kernel, grouped by its innermost frame inside this repo.
python tools/prof_small_ops.py [c3|c5]"""
import collections
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tools")]
import torch
import bench_train as B
from dldkd_amd import ops, train as T

cfg = (sys.argv + ["c3"])[1]
ops.set_gemm_precision("bf16")
m, opt, batch = B.build(cfg, 0.2, "cuda:0")
topt = types.SimpleNamespace(grad_clip=-1)
for _ in range(3):
    T.train_step(m, batch, opt, topt)
torch.cuda.synchronize()
import traceback
agg = collections.Counter()


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        t = next((x for x in a if torch.is_tensor(x)), None)
        if t is None or t.is_cuda or name in ("zeros", "zeros_like", "ones_like", "empty_like"):
            fr = [x for x in traceback.extract_stack()[:-1] if "dldkd_amd/" in x.filename or "/tools/bench_train" in x.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:90]}" if fr else "(no repo frame)"
            if name == "contiguous" and t is not None and t.is_contiguous():
                pass
            elif name == "float" and t is not None and t.dtype == torch.float32:
                pass
            else:
                agg[(name, where)] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for n in ("zeros", "zeros_like", "ones_like", "cat", "stack", "where", "sum"):
    wrap(torch, n)
for n in ("copy_", "clone", "contiguous", "fill_", "zero_", "float", "__add__", "__mul__", "__sub__", "__truediv__", "__iadd__", "sum", "to", "masked_fill", "__getitem__"):
    wrap(torch.Tensor, n)
T.train_step(m, batch, opt, topt)
torch.cuda.synchronize()
for (name, where), n in sorted(agg.items(), key=lambda kv: (kv[0][1], -kv[1])):
    print(f"{n:3d} {name:12s} {where}")
