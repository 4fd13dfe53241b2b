"""Host-side logic of the scoring path that needs no GPU (run by -m "not gpu")."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-dkd_amd"))


def test_pair_waves_plan_covers_every_video_once_and_pairs_only_full_waves():
    """scoring.pair_waves (the host plan of dldkd_simpool_eval_pairs_bf16): every sorted position in exactly one wave; a pair obeys
    112 < round_up(len A, 4) + len B <= 128; no zero-length video in a pair; tile count never above the one-video plan's."""
    import numpy as np
    from dldkd_amd.scoring import pair_waves
    rng = np.random.default_rng(0)
    cases = [rng.integers(24, 129, 21793), rng.integers(0, 129, 5000), np.full(100, 128), np.full(77, 30), np.full(9, 60),
             np.zeros(5, dtype=np.int64), np.arange(128, 0, -1), np.array([64, 64, 64]), np.array([], dtype=np.int64), np.array([7])]
    for lens in cases:
        sl = np.sort(np.asarray(lens, dtype=np.int64))[::-1]
        plan = pair_waves(sl)
        assert plan.dtype == np.int32 and plan.shape[1] == 2
        used = np.concatenate([plan[:, 0], plan[plan[:, 1] >= 0, 1]])
        assert np.array_equal(np.sort(used), np.arange(sl.shape[0]))
        pa = plan[plan[:, 1] >= 0]
        if pa.shape[0]:
            a, b = sl[pa[:, 0]], sl[pa[:, 1]]
            rows = (a + 3) // 4 * 4 + b
            assert (rows > 112).all() and (rows <= 128).all() and (b >= 1).all() and (a >= 1).all()
        single = sl[plan[plan[:, 1] < 0, 0]]
        assert 8 * pa.shape[0] + ((single + 15) // 16).sum() <= ((sl + 15) // 16).sum()
    big = np.sort(cases[0])[::-1]
    plan = pair_waves(big)
    assert plan.shape[0] < 0.65 * big.shape[0]                   # TVR-like lengths: 21.8 k videos in < 14.2 k waves
    with pytest.raises(Exception):
        pair_waves(np.array([3, 5]))                             # not descending
