"""GPU: K4 bf16 input projection (LayerNorm folded, both branches in one pass) against the oracle and the fp32
parity path, and its effect on end-to-end R@K."""
import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dv,M", [(3072, 300), (1024, 129), (768, 1), (64, 515)])
def test_in_proj_vs_oracle(dv, M):
    from dldkd_amd import ops
    m = _model(dv, dv, synth.make_params(7, dv, dv))
    g = torch.Generator().manual_seed(dv + M)
    x = torch.nn.functional.normalize(torch.randn(M, dv, generator=g).abs() + 0.1 * torch.randn(M, dv, generator=g), dim=-1)  # i3d-like: positive mean
    p = {k: v.cpu() for k, v in m.state_dict().items()}
    folded = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
    ys = ops.in_proj_bf16(x.to(DEV), folded)
    for y, pre in zip(ys, ("", "exp_")):
        ref = orc.input_projection(x.double(), {k: v.double() for k, v in p.items()}, pre + "visual_input_proj")
        err = (y.double().cpu() - ref).abs().max().item()
        assert err <= 2.5e-2 * max(1.0, ref.abs().max().item()), (pre, err, ref.abs().max().item())     # bf16 operands, K up to 3072
        rel = ((y.double().cpu() - ref).norm() / ref.norm()).item()
        assert rel < 6e-3, rel
    # weights are re-folded when a parameter changes
    with torch.no_grad():
        m.visual_input_proj.net[1].bias.add_(1.0)
    y2 = ops.in_proj_bf16(x.to(DEV), folded)[0]
    assert (y2 - ys[0]).abs().max() > 0.5


def test_fast_path_keeps_rank_parity():
    """End to end (towers + scorer) with fast_input_proj on: R@1/5/10/100 vs the fp32 oracle within the gate."""
    from dldkd_amd import eval as ev
    import types
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    opt = types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False, device=torch.device(DEV),
                                double_branch=True)
    with torch.no_grad():
        ctx0 = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        f0, a0, b0, metas = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx0)
        m.fast_input_proj = True
        ctx1 = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        f1, a1, b1, _ = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx1)
    assert (f0 - f1).abs().max().item() < 1.5e-2          # cosine scores, bf16 input projection vs fp32
    _, t2v = ev.get_gt(ctx0["video_metas"], metas)
    r0, r1 = ev.eval_q2m(-f0, t2v), ev.eval_q2m(-f1, t2v)
    # random-init weights: near-chance, near-tied rankings; allow two of 192 queries to cross a cut
    for x, y in zip(r0[:4], r1[:4]):
        assert abs(x - y) <= 1.05, (r0, r1)
