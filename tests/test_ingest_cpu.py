"""CPU: ingest host logic - the oracle's down-sampling against the reference's outputs (golden G7), our
sampling_bounds against the oracle's, and the BigFile reader on a file written here."""
import os

import numpy as np

import dldkd_oracle as orc

CASES = ((5, 8), (8, 8), (9, 8), (13, 8), (100, 16), (129, 128), (300, 128), (777, 128), (3, 1))


def test_oracle_sampling_vs_golden_g7(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_ingest.npz"))
    rs = np.random.RandomState(71)
    for n, max_len in CASES:
        f = rs.standard_normal((n, 12)).astype(np.float32)
        mine = orc.l2_normalize_rows(orc.uniform_feature_sampling(f, max_len))
        np.testing.assert_allclose(mine, g[f"n{n}_L{max_len}"], rtol=0, atol=1e-6)


def test_sampling_bounds_match_oracle():
    from dldkd_amd import ingest
    for n in list(range(1, 40)) + [127, 128, 129, 255, 256, 257, 1000, 4097]:
        for max_len in (1, 7, 8, 128):
            s, e = ingest.sampling_bounds(n, max_len)
            if n <= max_len:
                assert len(s) == n and (s == np.arange(n)).all()
            else:
                os_, oe = orc.sampling_bounds(n, max_len)
                assert (s == os_).all() and (e == oe).all() and len(s) == max_len


def test_bigfile_reader(tmp_path, golden_dir):
    from dldkd_amd import ingest
    g = np.load(os.path.join(golden_dir, "g7_ingest.npz"))
    rows = g["bigfile_rows"]
    ids = [f"vid{i}_f{i*3}" for i in range(7)]
    rows.tofile(tmp_path / "feature.bin")
    (tmp_path / "id.txt").write_text(" ".join(ids))
    (tmp_path / "shape.txt").write_text("7 6")
    bf = ingest.BigFile(str(tmp_path))
    assert bf.shape() == [7, 6]
    np.testing.assert_array_equal(bf.rows([ids[4], ids[0], ids[6]]), rows[[4, 0, 6]])
    assert bf.read_one(ids[2]) == rows[2].tolist()
    names, vecs = bf.read([ids[5], ids[1], "missing"])                      # reference semantics: sorted, unknown skipped
    assert names == list(g["bigfile_read_names"]) and np.allclose(np.array(vecs, np.float32), g["bigfile_read_vecs"])
    (tmp_path / "video2frames.txt").write_text(repr({"vid0": ids[:3], "vid1": ids[3:]}))
    assert ingest.read_dict(str(tmp_path / "video2frames.txt")) == {"vid0": ids[:3], "vid1": ids[3:]}
