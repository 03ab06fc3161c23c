"""The HIP path behind the reference's call signatures against outputs of the reference's OWN functions
(tests/golden/glue.npz: evaluate/eval_mscoco.py:18-50,56-99, evaluate/eval_flickr30k.py:56-100,
pretrain/create_negative_index.py:45-52 run on CPU by oracle/gen_golden.py::gen_glue; see tests/glue_util.py for why
scores are bit-identical on this data and where the reference's order is unspecified)."""
import json

import pytest
import torch

from glue_util import Glue
from vllm_amd import evaluate, mining, search

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_get_topk_candidates_dicts(golden_dir, compute):
    """a1: same dicts (keys, collapsed duplicate captions, best-first lists) for k = 1, 3, 5, 10, both directions."""
    g = Glue(golden_dir)
    g.check_dicts(lambda q, c, k: search.get_topk_candidates(q, c, k, compute=compute))
    images, text = g.lists()
    assert all(len(v) == 3 for v in search.get_topk_candidates(images, text).values())       # default k = 3


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_eval_scripts_print_the_reference_lines(golden_dir, compute, monkeypatch):
    """a2 / f2: eval_mscoco(fxn) and eval_flickr(fxn) print the six lines the reference printed."""
    g = Glue(golden_dir)
    for name, var, fn, sub in (("mscoco", "MSCOCO_EVAL", evaluate.eval_mscoco, None),
                               ("flickr", "FLICKR_EVAL", evaluate.eval_flickr, "flickr30k_images")):
        monkeypatch.setenv(var, g.json_path)
        out = []
        fn(g.fxn(sub), compute=compute, out=out.append)
        assert out == g.lines(name)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_negatives_for_sample_lists(golden_dir, compute):
    """a3: negatives_for_sample(idx, negative_dict, q, c) per row, and the batched mine_negatives, against the
    reference's 100-entry lists incl. the edge cases (negative positive score, fewer than 100 survivors, exact ties)."""
    g = Glue(golden_dir)
    q, c = g.mining_inputs()
    cc = search.Embeddings(c, compute)
    d = {}
    for idx in range(0, q.shape[0], 7):                      # the per-sample entry point on a stride of rows ...
        search.negatives_for_sample(idx, d, q, cc, compute=compute)
    for idx in (5, 6, 7, 40, 41, 199, 200, 300, 309, 310, 315):   # ... and on every edge-case row
        search.negatives_for_sample(idx, d, q, cc, compute=compute)
    st = g.check_negatives(d, rows=sorted(d))
    assert st["short_rows"] >= 2 and d[5][0] == 5
    ids, vals, s_pos = search.mine_negatives(search.Embeddings(q, compute), cc, 100)       # ... the batched form on all
    d2 = {i: row for i, row in enumerate(ids.cpu().tolist())}
    st = g.check_negatives(d2)
    assert st["rows"] == q.shape[0]
    exact = (c * q).sum(1)                                    # exact in fp32 on this grid
    assert torch.equal(s_pos.cpu(), exact)


def test_mining_driver_writes_the_reference_lists(golden_dir, tmp_path):
    """a4: query.pt / cand.pt -> negatives.json through the resumable driver, same lists as the reference's loop."""
    g = Glue(golden_dir)
    q, c = g.mining_inputs()
    d = str(tmp_path / "negative")
    mining.save_embeddings(d, q, c)
    out = mining.create_negative_index(d, top=100, compute="f32", block=128)
    got = json.loads(open(out).read())
    assert list(got) == [str(i) for i in range(q.shape[0])]
    g.check_negatives({int(k): v for k, v in got.items()})


def test_stacked_vector_embeddings(golden_dir):
    """The candidate form of evaluate/mmeb_vqa.py:24 ([D] vectors, torch.stack) gives the same dicts as [1, D] pieces;
    a list mixing both forms is refused."""
    g = Glue(golden_dir)
    images, text = g.lists()
    flat = [(i, e[0]) for i, e in text]
    assert search.get_topk_candidates(images, flat, 5) == search.get_topk_candidates(images, text, 5)
    flat_q = [(i, e[0]) for i, e in images]
    assert search.get_topk_candidates(flat_q, flat, 5) == search.get_topk_candidates(images, text, 5)
    with pytest.raises(search.AbcError):
        search.get_topk_candidates(images, flat[:3] + text[3:], 5)
