"""north_star's gate on eval numbers - R@1/5/10/100 within +-0.1 of the reference - for the path eval_epoch runs by default
(throughput mode: bf16 input projection K4, fused bf16 tower kernel K5, bf16 scorer) AND for the parity path, from RAW features,
1,536 videos x 2,048 queries, against the fp32 oracle towers + oracle scoring on the CPU.  The signal is planted in feature
space (tools/rk_gate.py) so that the oracle's R@1 sits where TVR's does (15-40 %), not at chance and not at 100 %."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_recall_gate_from_raw_features_both_modes():
    import rk_gate
    out = rk_gate.run(nv=1536, nq=2048, sigma=14.0, seed=11, modes=("parity", "fast", "resident"))
    assert 15.0 <= out["oracle"][0] <= 40.0, out["oracle"]                  # TVR-like operating point
    assert out["oracle"][3] < 95.0
    for mode in ("parity", "fast", "resident"):                             # resident = eval_epoch's default (bf16 table, bf16 h0)
        d = out[mode]["delta_vs_oracle"]
        assert max(abs(x) for x in d) <= 0.1 + 1e-9, (mode, out[mode])     # the gate: +-0.1 on every cut (2 of 2,048 queries)
    assert out["parity"]["max_abs_score_err"] < 2e-3 and out["fast"]["max_abs_score_err"] < 4e-3, out
    assert out["resident"]["max_abs_score_err"] < 4e-3, out


def test_eval_epoch_runs_throughput_mode_by_default_and_restores_the_precision():
    """eval_epoch's default precision is the gated throughput mode (opt.eval_precision = "parity" opts out); the caller's
    GEMM precision and the model's projection flag are restored afterwards."""
    import synth
    from test_encoder_gpu import _model
    from dldkd_amd import eval as ev, ops
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    mk = lambda **kw: types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False,   # noqa: E731
                                            device=torch.device("cuda:0"), double_branch=True, **kw)
    seen = []
    real, real_res = m.encode_context_into, m.encode_resident_into      # padded super-batches / the resident feature table
    m.encode_context_into = lambda *a, **k: seen.append(ops.gemm_precision()) or real(*a, **k)
    m.encode_resident_into = lambda *a, **k: seen.append(ops.gemm_precision()) or real_res(*a, **k)
    with torch.no_grad():
        fast = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), mk())
        assert seen and set(seen) == {"bf16"}
        assert ops.gemm_precision() == "fp32" and m.fast_input_proj is False
        n = len(seen)
        par = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), mk(eval_precision="parity"))
        assert len(seen) == n                                                 # parity mode never enters the fused path
        # the final / test evaluation (test=True) reports fp32-grade numbers unless the caller asks for throughput mode
        final = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), mk(), test=True)
        assert len(seen) == n and final == par
        ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), mk(eval_precision="throughput"), test=True)
        assert len(seen) > n
        assert ev.eval_precision_mode(mk()) == "throughput" and ev.eval_precision_mode(mk(), test=True) == "parity"
    assert abs(fast - par) <= 4 * 100.0 / 192 * 2 + 1e-9                      # 192 queries, random-init near-ties: <= 2 queries per cut
    with pytest.raises(ValueError):
        ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), mk(eval_precision="fp8"))
