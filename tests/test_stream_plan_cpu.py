"""CPU: the host planner of scorer v4 (dldkd_simpool_plan_stream) - pure host code, no GPU needed.
Invariants checked against a direct reading of the plan: every valid clip appears exactly once and in order, every
16-row tile has at most one segment end, gap rows are zero rows, every video's 1-2 units are where the plan says its
rows end, and a wave's open tail is the video that continues into the next wave."""
import ctypes

import numpy as np
import pytest

from dldkd_amd import native


def _plan(lens, lp):
    L = native.lib()
    lens = np.ascontiguousarray(np.asarray(lens, np.int32))
    nv = len(lens)
    mw = int((int(lens.sum()) + 15 * nv) // 128 + 2)
    rowsrc = np.empty(mw * 128, np.int32)
    te, tu, tail = np.empty(mw * 8, np.int32), np.empty(mw * 8, np.int32), np.empty(mw, np.int32)
    u0, u1 = np.empty(max(nv, 1), np.int32), np.empty(max(nv, 1), np.int32)
    nw, nu = ctypes.c_int(0), ctypes.c_int(0)
    hp = lambda a: ctypes.c_void_p(a.ctypes.data)   # noqa: E731
    rc = L.dldkd_simpool_plan_stream(hp(lens), nv, lp, mw, hp(rowsrc), hp(te), hp(tu), hp(tail), hp(u0), hp(u1),
                                     ctypes.cast(ctypes.byref(nw), ctypes.c_void_p), ctypes.cast(ctypes.byref(nu), ctypes.c_void_p))
    return rc, rowsrc, te, tu, tail, u0[:nv], u1[:nv], nw.value, nu.value


@pytest.mark.parametrize("seed,nv,lo,hi,lp", [(0, 1, 1, 1, 32), (1, 50, 1, 128, 128), (2, 300, 24, 128, 128), (3, 200, 1, 15, 32),
                                              (4, 64, 128, 128, 128), (5, 500, 1, 40, 64), (6, 97, 16, 16, 32)])
def test_plan_invariants(seed, nv, lo, hi, lp):
    rs = np.random.RandomState(seed)
    lens = rs.randint(lo, hi + 1, size=nv)
    rc, rowsrc, te, tu, tail, u0, u1, nw, nu = _plan(lens, lp)
    assert rc == 0
    used = rowsrc[:nw * 128]
    assert (rowsrc[nw * 128:] == -1).all()
    # every clip once, in order
    valid = used[used >= 0]
    expect = np.concatenate([v * lp + np.arange(n) for v, n in enumerate(lens)])
    assert (valid == expect).all()
    # walk the stream: segment ends, units, gaps
    pos_of = {int(r): i for i, r in enumerate(used) if r >= 0}
    ends_per_tile = np.zeros(nw * 8, int)
    units_seen = set()
    for v, n in enumerate(lens):
        start, end = pos_of[v * lp], pos_of[v * lp + n - 1] + 1
        assert end - start == n                              # a video's rows are contiguous in the stream
        T = (end - 1) // 16
        ends_per_tile[T] += 1
        assert (te[T] & 31) == end - 16 * T
        if start // 128 != (end - 1) // 128:                 # straddles a wave boundary: open tail + closing unit
            assert tail[start // 128] == u0[v] and tu[T] == u1[v] and u1[v] >= 0
            units_seen.update((int(u0[v]), int(u1[v])))
        else:
            assert tu[T] == u0[v] and u1[v] == -1
            units_seen.add(int(u0[v]))
        gap = bool(te[T] & 256)
        rest = used[end:(T + 1) * 16]
        assert (rest == -1).all() if gap else (rest >= 0).all() or end == (T + 1) * 16
    assert ends_per_tile.max() <= 1
    assert (te[:nw * 8][ends_per_tile == 0] == 0).all() and (te[nw * 8:] == 0).all()
    assert units_seen == set(range(nu))
    assert sum(1 for w in range(nw) if tail[w] >= 0) == int((u1 >= 0).sum())
    assert nw <= (int(lens.sum()) + 15 * nv) // 128 + 1


def test_plan_rejects_bad_lengths():
    assert _plan([5, 0, 3], 32)[0] != 0
    assert _plan([5, 33], 32)[0] != 0
    assert _plan([129], 160)[0] != 0


def test_plan_invariants_hypothesis():
    """Property test over arbitrary length lists (hypothesis): the planner never loses, duplicates or reorders a clip,
    never puts two segment ends in one tile, and every video ends up with 1-2 units that partition its rows."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.integers(min_value=1, max_value=128), min_size=1, max_size=120))
    def check(lens):
        rc, rowsrc, te, tu, tail, u0, u1, nw, nu = _plan(lens, 128)
        assert rc == 0
        used = rowsrc[:nw * 128]
        valid = used[used >= 0]
        assert (valid == np.concatenate([v * 128 + np.arange(n) for v, n in enumerate(lens)])).all()
        ends = np.zeros(nw * 8, int)
        pos = 0
        where = {int(r): i for i, r in enumerate(used) if r >= 0}
        for v, n in enumerate(lens):
            s, e = where[v * 128], where[v * 128 + n - 1] + 1
            assert e - s == n and s >= pos
            pos = e
            ends[(e - 1) // 16] += 1
            two = s // 128 != (e - 1) // 128
            assert (u1[v] >= 0) == two and u0[v] >= 0
        assert ends.max() <= 1 and nu == len(lens) + int((u1 >= 0).sum())
    check()
