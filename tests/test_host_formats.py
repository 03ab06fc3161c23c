"""CPU tests of the file formats and host logic either side of the search path:
negatives.json text, the consumer's sampling rule, Karpathy JSON handling, and the resumable driver's
bookkeeping with the device search replaced by the oracle (tests only; the product has no CPU path)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from vllm_amd import evaluate, mining, search


def test_negatives_json_text_equals_json_dump(tmp_path):
    rng = np.random.RandomState(0)
    arr = rng.randint(0, 1000, size=(37, 12)).astype(np.int64)
    ref = json.dumps({i: arr[i].tolist() for i in range(37)})
    path = str(tmp_path / "negatives.json")
    n = mining.write_negatives_json(path, iter([(0, arr[:10]), (10, arr[10:30]), (30, arr[30:])]))
    assert n == 37
    assert open(path).read() == ref
    assert mining.load_negatives(path)["36"] == arr[36].tolist()
    # holes (-1) are dropped, empty input gives "{}"
    arr2 = np.array([[5, 3, -1], [-1, -1, -1]])
    mining.write_negatives_json(path, iter([(0, arr2)]))
    assert open(path).read() == json.dumps({0: [5, 3], 1: []})
    mining.write_negatives_json(path, iter([]))
    assert open(path).read() == "{}"
    with pytest.raises(mining.AbcError):
        mining.write_negatives_json(path, iter([(5, arr2)]))


def test_consumer_sampling_rule_matches_reference():
    meta = {str(i): list(range(1000 + 100 * i, 1100 + 100 * i)) for i in range(4)}
    for negatives in (1, 7, 12, 20):
        np.random.seed(123)
        ref = [oracle.attach_negative_rows(meta, i, negatives) for i in range(4)]
        np.random.seed(123)
        got = [mining.sample_negative_rows(meta, i, negatives) for i in range(4)]
        assert got == ref
    mining.check_negative_index(4, {k: [0, 1, 2, 3, 0] * 20 for k in meta}, negatives=20)
    with pytest.raises(mining.AbcError):
        mining.check_negative_index(5, meta)
    with pytest.raises(mining.AbcError):
        mining.check_negative_index(4, {k: [0, 1, 2] for k in meta}, negatives=1)


def _karpathy(tmp_path, n_img=12, flickr=False):
    images = []
    for i in range(n_img + 3):
        sents = [{"raw": f"caption {i}-{j}" if (i, j) != (3, 1) else "caption 2-0"} for j in range(5)]   # a duplicated string
        images.append({"split": "test" if i < n_img else "train", "filepath": "val2014", "filename": f"im{i}.jpg",
                       "sentences": sents})
    p = tmp_path / ("dataset_flickr30k.json" if flickr else "dataset_coco.json")
    p.write_text(json.dumps({"images": images}))
    return str(p)


def test_karpathy_paths_match_reference(tmp_path):
    p = _karpathy(tmp_path)
    assert [x["image"] for x in evaluate.load_karpathy_test(p)] == [x["image"] for x in oracle.karpathy_paths(p)]
    pf = _karpathy(tmp_path, flickr=True)
    got = evaluate.load_karpathy_test(pf, image_dir="flickr30k_images")
    ref = oracle.karpathy_paths(pf, image_dir="flickr30k_images")
    assert [x["image"] for x in got] == [x["image"] for x in ref] and len(got) == 12
    assert evaluate.intersect(["a", "b"], ["b"]) and not evaluate.intersect(["a"], ["b"])


class _CpuEmb:
    """Stand-in for search.Embeddings in the bookkeeping test (CPU tensors, oracle arithmetic)."""

    def __init__(self, x):
        self.data, self.scale, self.compute = x.float(), None, "f32"

    @property
    def shape(self):
        return self.data.shape


def test_driver_resume_bookkeeping(tmp_path, monkeypatch):
    q, c = oracle.make_clustered(300, 300, 64, seed=3, n_centres=6)
    d = str(tmp_path / "negative")
    mining.save_embeddings(d, q, c)
    calls = []

    def fake_mine(qb, shard, k, ratio=0.95, offset=0, **kw):
        calls.append(offset)
        nd = {}
        for i in range(qb.shape[0]):
            oracle.negatives_for_sample(offset + i, nd, q, c.clone(), top=k)
        idx = torch.tensor([nd[offset + i] for i in range(qb.shape[0])])
        return idx, None, None

    class FakeStage:                               # mining._StagedRows without a GPU: the block as it is
        def __init__(self, x, block, compute, device):
            self.x, self.nxt = x, None

        def prefetch(self, lo, hi):
            assert self.nxt is None
            self.nxt = (lo, hi)

        def take(self):
            (lo, hi), self.nxt = self.nxt, None
            return _CpuEmb(self.x[lo:hi])

        def close(self):
            pass

    class FakeDownload:                            # mining._Download without a GPU
        def __init__(self, block, top, device):
            pass

        def start(self, slot, idx):
            return idx

        def wait(self, handle):
            return handle.numpy(), bool((handle[:, -1] < 0).any())

    monkeypatch.setattr(mining, "_StagedRows", FakeStage)
    monkeypatch.setattr(mining, "_Download", FakeDownload)
    monkeypatch.setattr(mining, "upload", lambda x, compute, rows=None, **kw: _CpuEmb(x[rows[0]:rows[1]]))
    monkeypatch.setattr(mining.search, "Embeddings", lambda x, compute, dev=None: _CpuEmb(x))
    monkeypatch.setattr(mining.parallel.ShardedCorpus, "mine_negatives",
                        lambda self, qb, k, ratio=0.95, offset=0, **kw: fake_mine(qb, self.shard, k, ratio, offset))

    class Crash(Exception):
        pass

    def crash_after_two(done, total):
        if done >= 128:
            raise Crash()

    with pytest.raises(Crash):
        mining.create_negative_index(d, top=100, block=64, progress=crash_after_two)
    # the driver is one block ahead of what it has written: block 128 was searched, blocks 0 and 64 are on disk
    assert calls == [0, 64, 128]
    assert sorted(f for f in os.listdir(os.path.join(d, mining.PARTS_DIR)) if f.endswith(".npy")) == \
        ["part-000000000000.npy", "part-000000000064.npy"]
    assert not os.path.exists(os.path.join(d, "negatives.json")) and not os.path.exists(os.path.join(d, "negatives.json.tmp"))
    # torn file from the crash window must be redone, finished blocks are kept
    open(os.path.join(d, mining.PARTS_DIR, "part-000000000128.npy"), "wb").write(b"\x93NUMPY garbage")
    calls.clear()
    out = mining.create_negative_index(d, top=100, block=64)
    assert calls == [128, 192, 256]
    assert open(out).read() == oracle.create_negative_index_ref(q, c, top=100)
    assert not os.path.exists(os.path.join(d, mining.PARTS_DIR))
    # a changed configuration invalidates the parts
    calls.clear()
    mining.create_negative_index(d, top=50, block=64, keep_parts=True)
    assert len(calls) == 5 and os.path.isdir(os.path.join(d, mining.PARTS_DIR))
    calls.clear()
    mining.create_negative_index(d, top=50, block=64)          # everything found: no search at all
    assert calls == []
    neg = mining.load_negatives(out)
    mining.check_negative_index(300, neg, negatives=10)
