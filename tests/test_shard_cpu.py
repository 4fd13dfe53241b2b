"""The persisted packed shard (dldkd_amd.ingest, round 6): container round trips bit for bit, a reader never accepts a file it
cannot vouch for (magic, version, truncation, counts), and a ragged fp32 shard obeys the caption-dataset protocol eval_epoch sees
(method/data_provider.py:344-354).  numpy only - no GPU, no h5py."""
import os
import struct

import numpy as np
import pytest

from dldkd_amd import ingest


def _resident(tmp_path, lens=(3, 0, 5, 1), K=64, seed=0):
    rs = np.random.RandomState(seed)
    n = int(sum(lens))
    rows = rs.standard_normal((n, K)).astype(np.float16)
    mean, rstd = rs.standard_normal(n).astype(np.float32), rs.rand(n).astype(np.float32)
    ids = [f"vid{i:03d}" for i in range(len(lens))]
    p = str(tmp_path / "g.shard")
    ingest.write_shard(p, ingest.SHARD_RESIDENT, K, lens, ids, [rows[:2], rows[2:]], mean, rstd, 1e-5)
    return p, rows, mean, rstd, ids


def test_resident_shard_round_trip_is_bit_exact(tmp_path):
    p, rows, mean, rstd, ids = _resident(tmp_path)
    s = ingest.Shard(p)
    assert (s.kind, s.K, s.n_items, s.n_rows) == (ingest.SHARD_RESIDENT, 64, 4, 9) and s.ids == ids
    assert s.rows.dtype == np.float16 and (np.asarray(s.rows).view(np.uint16) == rows.view(np.uint16)).all()
    assert (np.asarray(s.mean).view(np.uint32) == mean.view(np.uint32)).all() and (np.asarray(s.rstd) == rstd).all()
    assert list(s.lens) == [3, 0, 5, 1] and s.item(2).shape == (5, 64) and (s.item(2) == rows[3:8]).all()
    assert abs(s.ln_eps - 1e-5) < 1e-12
    assert os.path.getsize(p) % 4096 == 0 and not os.path.exists(p + ".tmp")


def test_reader_rejects_what_it_cannot_vouch_for(tmp_path):
    p, *_ = _resident(tmp_path)
    raw = bytearray(open(p, "rb").read())
    bad = str(tmp_path / "bad.shard")
    open(bad, "wb").write(b"NOTASHRD" + bytes(raw[8:]))
    with pytest.raises(ingest.ShardError, match="magic"):
        ingest.Shard(bad)
    v2 = bytearray(raw); struct.pack_into("<I", v2, 8, 99)
    open(bad, "wb").write(bytes(v2))
    with pytest.raises(ingest.ShardError, match="version"):
        ingest.Shard(bad)
    open(bad, "wb").write(bytes(raw[:len(raw) - 4096]))
    with pytest.raises(ingest.ShardError, match="truncated"):
        ingest.Shard(bad)
    with pytest.raises(ingest.ShardError, match="ids for"):
        ingest.write_shard(bad, ingest.SHARD_RAGGED_F32, 4, [1, 2], ["only-one"], [np.zeros((3, 4), np.float32)])
    with pytest.raises(ingest.ShardError, match="rows written"):
        ingest.write_shard(bad, ingest.SHARD_RAGGED_F32, 4, [1, 2], ["a", "b"], [np.zeros((2, 4), np.float32)])
    with pytest.raises(ingest.ShardError, match="needs mean"):
        ingest.write_shard(bad, ingest.SHARD_RESIDENT, 4, [1], ["a"], [np.zeros((1, 4), np.float16)])


def test_ragged_shard_is_a_caption_dataset(tmp_path):
    """What a maintainer's HDF5 -> shard converter produces for the word features: item i = (len_i, Dq) fp32 rows
    + the caption id "<video_id>#..."; the dataset cuts to max_desc_l and L2-normalises per token like the reference's
    TxtDataSet4DLDKD.__getitem__ (data_provider.py:347-352: norm + 1e-5)."""
    rs = np.random.RandomState(3)
    arrs = [rs.standard_normal((n, 48)).astype(np.float32) * 3 for n in (4, 37, 1)]
    ids = ["v1#enc#0", "v1#enc#1", "v2#enc#0"]
    p = str(tmp_path / "t.shard")
    ingest.save_ragged_f32(p, arrs, ids)
    ds = ingest.RaggedShardDataset(p, max_len=30)
    assert len(ds) == 3 and ds.ids == ids
    f, i, cid = ds[1]
    assert f.shape == (30, 48) and i == 1 and cid == "v1#enc#1"
    ref = arrs[1][:30] / (np.linalg.norm(arrs[1][:30], axis=-1, keepdims=True) + 1e-5)
    assert np.array_equal(f.numpy(), ref.astype(np.float32))
    raw = ingest.RaggedShardDataset(p, normalize=False)[0][0].numpy()
    assert np.array_equal(raw, arrs[0])
    empty = str(tmp_path / "e.shard")
    ingest.write_shard(empty, ingest.SHARD_RAGGED_F32, 48, [], [], [])
    assert len(ingest.RaggedShardDataset(empty)) == 0
