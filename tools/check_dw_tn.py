"""gemm_bf16_tn.hip through dldkd_tower_train_dw / dldkd_inproj_bwd_bf16: weight gradients and bias column sums against torch (fp64
of the bf16 operands)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch

from dldkd_amd import native

dev = "cuda:0"
L_ = native.lib()
H = 384
p = native.ptr


def run(rows, nb, seed=0, ones=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    A = [torch.ones(rows, H, device=dev) if ones else torch.randn(rows, H, generator=g, device=dev) for _ in range(nb)]
    A16 = [a.bfloat16().contiguous() for a in A]
    B16 = [torch.randn(rows, H, generator=g, device=dev).bfloat16().contiguous() for _ in range(nb)]
    dW = torch.empty(nb * H, H, device=dev)
    dB = torch.zeros(nb * H, device=dev)
    wsb = L_.dldkd_tower_train_dw_workspace_bytes(nb, rows)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    hA, hB = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
    hl, hc, h16 = (ctypes.c_int * nb)(), (ctypes.c_int * nb)(), (ctypes.c_int * nb)()
    for i in range(nb):
        hA[i], hB[i], hl[i], hc[i], h16[i] = A16[i].data_ptr(), B16[i].data_ptr(), H, 0, 1
    native.check(L_.dldkd_tower_train_dw(hA, hl, hc, h16, hB, nb, rows, p(dW), p(dB), p(ws), wsb, None, native.stream()), "dw")
    torch.cuda.synchronize()
    for i in range(nb):
        ref = A16[i].double().t() @ B16[i].double()
        refb = A16[i].double().sum(0)
        ew = ((dW[i * H:(i + 1) * H].double() - ref).norm() / ref.norm()).item()
        eb = ((dB[i * H:(i + 1) * H].double() - refb).norm() / refb.norm()).item()
        print(f"rows {rows} block {i}: dW rel {ew:.2e}  dbias rel {eb:.2e}  dbias[:4] {dB[i * H:i * H + 4].tolist()} ref {refb[:4].tolist()}")


for rows, nb, ones in ((256, 1, True), (256, 2, False), (3072, 5, False), (16384, 5, False)):
    run(rows, nb, ones=ones)
