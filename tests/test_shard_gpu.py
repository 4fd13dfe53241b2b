"""GPU: the resident feature table persisted and read back (ingest.save_resident / load_resident, opt.eval_resident_shard) and the
BigFile -> shard converter (utils/basic_utils.py:9-68; method/data_provider.py:283-309)."""
import os
import types

import numpy as np
import pytest
import torch

import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_resident_table_round_trips_through_a_shard_bit_for_bit(tmp_path):
    from dldkd_amd import ingest, ops
    rs = np.random.RandomState(0)
    t = ops.ResidentRows(3072, DEV)
    lens_all, ids = [], []
    for b in range(3):
        lens = rs.randint(1, 40, size=17)
        feat = torch.from_numpy(rs.standard_normal((17, 40, 3072)).astype(np.float32)).to(DEV)
        t.append(feat, lens)
        lens_all += list(lens)
        ids += [f"v{b}_{i}" for i in range(17)]
    p = str(tmp_path / "gallery.shard")
    n_items, n_rows = ingest.save_resident(p, t, ids)
    assert (n_items, n_rows) == (51, int(sum(lens_all))) == (51, t.rows)
    t2, ids2 = ingest.load_resident(p, DEV)
    assert ids2 == ids and t2.rows == t.rows and t2.lens == [int(v) for v in lens_all] and t2.K == 3072
    assert torch.equal(t2.xb[:t.rows].view(torch.int16), t.xb[:t.rows].view(torch.int16))
    assert torch.equal(t2.mean[:t.rows], t.mean[:t.rows]) and torch.equal(t2.rstd[:t.rows], t.rstd[:t.rows])
    # a ring smaller than the table: several buffers in flight, same bits
    dst = torch.empty_like(t.xb[:t.rows])
    ingest.upload_rows(ingest.Shard(p).rows, dst, ring_bytes=1 << 16)
    assert torch.equal(dst.view(torch.int16), t.xb[:t.rows].view(torch.int16))


def test_eval_epoch_starts_from_the_shard_without_touching_the_features(tmp_path):
    from dldkd_amd import eval as ev
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    shard = str(tmp_path / "val_gallery.shard")
    opt = types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False, device=torch.device(DEV),
                                double_branch=True, eval_precision="throughput", eval_resident_shard=shard)
    with torch.no_grad():
        first = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
    assert os.path.exists(shard)

    class _NoFeatures(synth.ListDataset):                       # a "fresh process": same ids, features must not be read
        def __getitem__(self, i):
            raise AssertionError("the gallery features were read although the shard exists")

    with torch.no_grad():
        again = ev.eval_epoch(m, _NoFeatures(list(vids)), synth.ListDataset(list(txts)), opt)
    assert again == first
    other = types.SimpleNamespace(**{**vars(opt), "eval_resident_shard": shard})
    with pytest.raises(RuntimeError, match="do not match the gallery"):
        with torch.no_grad():
            ev.eval_epoch(m, _NoFeatures(list(vids[:10])), synth.ListDataset(list(txts)), other)


def test_bigfile_converter_writes_the_table_a_first_epoch_builds(tmp_path):
    from dldkd_amd import ingest, ops
    rs = np.random.RandomState(7)
    d = tmp_path / "feat"
    d.mkdir()
    frames = {f"vid{i}": [f"vid{i}_{j}" for j in range(int(rs.randint(3, 200)))] for i in range(9)}
    names = [n for v in frames.values() for n in v]
    mat = rs.standard_normal((len(names), 128)).astype(np.float32)
    (d / "shape.txt").write_text(f"{len(names)} 128")
    (d / "id.txt").write_text(" ".join(names))
    mat.tofile(str(d / "feature.bin"))
    bf = ingest.BigFile(str(d))
    vids = list(frames)
    p = str(tmp_path / "bf.shard")
    ingest.bigfile_to_resident_shard(bf, frames, vids, 128, p, DEV, batch=4)
    t, ids = ingest.load_resident(p, DEV)
    assert ids == vids and t.lens == [min(len(frames[v]), 128) for v in vids]
    ref = ops.ResidentRows(128, DEV)
    for lo in range(0, 9, 4):
        feat, mask = ingest.load_gallery_batch(bf, frames, vids[lo:lo + 4], 128, DEV)
        ref.append(feat, mask.sum(1).long().cpu().numpy())
    assert torch.equal(t.xb[:t.rows].view(torch.int16), ref.xb[:ref.rows].view(torch.int16)) and torch.equal(t.rstd[:t.rows], ref.rstd[:ref.rows])
