"""GPU: segment-mean + L2-norm + padding kernel against the oracle and the reference's outputs (golden G7)."""
import os

import numpy as np
import pytest
import torch

import dldkd_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ((5, 8), (8, 8), (9, 8), (13, 8), (100, 16), (129, 128), (300, 128), (777, 128), (3, 1))


def test_build_video_batch_vs_golden_g7(golden_dir):
    from dldkd_amd import ingest
    g = np.load(os.path.join(golden_dir, "g7_ingest.npz"))
    rs = np.random.RandomState(71)
    for n, max_len in CASES:
        f = rs.standard_normal((n, 12)).astype(np.float32)
        out, mask = ingest.build_video_batch([f], max_len, DEV)
        ref = g[f"n{n}_L{max_len}"]
        assert out.shape == (1, ref.shape[0], 12) and float(mask.sum()) == ref.shape[0]
        np.testing.assert_allclose(out[0].cpu().numpy(), ref, rtol=0, atol=2e-6)


def test_ragged_batch_padding_and_order():
    from dldkd_amd import ingest
    rs = np.random.RandomState(3)
    arrays = [rs.standard_normal((n, 3072)).astype(np.float32) for n in (40, 500, 1, 128, 129)]
    out, mask = ingest.build_video_batch(arrays, 128, DEV)
    assert out.shape == (5, 128, 3072) and mask.sum(1).tolist() == [40, 128, 1, 128, 128]
    for i, a in enumerate(arrays):
        ref = orc.l2_normalize_rows(orc.uniform_feature_sampling(a, 128))
        np.testing.assert_allclose(out[i, :ref.shape[0]].cpu().numpy(), ref, rtol=0, atol=2e-6)
        assert float(out[i, ref.shape[0]:].abs().sum()) == 0.0
