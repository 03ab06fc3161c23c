"""GPU parity of the host-side mirror (vllm_amd.search) against the reference's outputs (golden
neg_mine.batch fixtures) and the oracle's restatements of eval_mscoco / create_negative_index."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _bf16(bits):
    return torch.from_numpy(bits.copy()).view(torch.bfloat16).float()


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_batch_matches_reference_golden(golden_dir, compute):
    from vllm_amd import search
    z = np.load(os.path.join(golden_dir, "neg_mine_batch.npz"))
    q, c = _bf16(z["s_q_bf16"]), _bf16(z["s_c_bf16"])      # bf16-exact inputs: both computes see the same numbers
    for k in (1, 5, 10, 64, 100):
        sc, ix, frac, pos = search.batch(q, c, 0, 3000, k, compute=compute)
        ref_sc = torch.from_numpy(z[f"s_k{k}_score"])
        assert torch.allclose(sc, ref_sc, rtol=2e-5, atol=1e-6)
        oracle.check_topk_parity(q, c, sc, ix, rtol=2e-5)
        assert torch.allclose(pos, torch.from_numpy(z["s_pos"]), rtol=2e-5, atol=1e-6)
    # compare-count: candidates within tolerance of the positive may fall either side
    s = oracle.scores_f32(q, c, fp64=True)
    p64 = s.diagonal().unsqueeze(1)
    lo = (s > p64 + 2e-5 * p64.abs()).sum(1)
    hi = (s > p64 - 2e-5 * p64.abs()).sum(1)
    cnt = torch.round(frac.double() * c.shape[0]).long()
    assert bool(((cnt >= lo) & (cnt <= hi)).all())
    ref_cnt = torch.round(torch.from_numpy(z["s_frac"]).double() * c.shape[0]).long()
    assert float((cnt == ref_cnt).float().mean()) > 0.98
    sc, ix, frac, pos = search.batch(q, c, 300, 200, 10, compute=compute)
    assert torch.allclose(sc, torch.from_numpy(z["s_b300_score"]), rtol=2e-5, atol=1e-6)
    assert torch.allclose(pos, torch.from_numpy(z["s_b300_pos"]), rtol=2e-5, atol=1e-6)


def test_batch_large_seeded_golden(golden_dir):
    from vllm_amd import search
    z = np.load(os.path.join(golden_dir, "neg_mine_batch.npz"))
    q, c = oracle.make_clustered(3000, 4000, 256, seed=int(z["l_seed"]), n_centres=64)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    sc, ix, frac, pos = search.batch(q, c, 0, 3000, 10, compute="bf16")
    assert torch.allclose(sc, torch.from_numpy(z["l_score"]), rtol=2e-5, atol=1e-6)
    ref_ix = torch.from_numpy(z["l_idx"]).long()
    assert float((ix == ref_ix).float().mean()) > 0.999          # exact except inside fp32 ties
    oracle.check_topk_parity(q, c, sc, ix, rtol=2e-5)


def test_get_topk_candidates_recall_identical():
    """a1/a2 on a synthetic Karpathy-shaped split (5 captions per image, some duplicated strings):
    recall@1/5/10 identical to the CPU reference path, both directions."""
    from vllm_amd import search
    g = torch.Generator().manual_seed(4)
    n_img, d = 200, 256
    img = torch.nn.functional.normalize(torch.randn(n_img, d, generator=g), dim=-1)
    test, text = [], []
    for i in range(n_img):
        sents = [{"raw": f"caption {i} {j}" if not (i % 17 == 0 and j == 4) else "a photo"} for j in range(5)]
        test.append({"image": f"/data/img{i}.jpg", "sentences": sents})
    images = [(x["image"], img[i:i + 1]) for i, x in enumerate(test)]
    for i, x in enumerate(test):
        for s in x["sentences"]:
            e = torch.nn.functional.normalize(img[i] + 1.2 * torch.randn(d, generator=g) / d ** 0.5 * 3, dim=-1)
            text.append((s["raw"], e[None]))
    multi_i2t = search.get_topk_candidates_multi(images, text, (1, 5, 10))
    multi_t2i = search.get_topk_candidates_multi(text, images, (1, 5, 10))
    for k in (1, 5, 10):
        ref1, ref2 = oracle.get_topk_candidates(images, text, k), oracle.get_topk_candidates(text, images, k)
        got1, got2 = search.get_topk_candidates(images, text, k), search.get_topk_candidates(text, images, k)
        assert got1 == multi_i2t[k] and got2 == multi_t2i[k]
        r = (oracle.recall_i2t(test, ref1), oracle.recall_t2i(test, ref2))
        gr = (oracle.recall_i2t(test, got1), oracle.recall_t2i(test, got2))
        assert f"{r[0]:.4f}" == f"{gr[0]:.4f}" and f"{r[1]:.4f}" == f"{gr[1]:.4f}"
        same = sum(ref1[q] == got1[q] for q in ref1) / len(ref1)
        assert same > 0.99
    assert 0.2 < r[0] <= 1.0


def test_negatives_for_sample_and_batched_mining():
    from vllm_amd import search
    q, c = oracle.make_clustered(300, 300, 128, seed=8, n_centres=8)
    c[10] = c[3]                                  # duplicate row -> exact tie
    q[7] = -q[7]                                  # negative positive score
    ce = search.Embeddings(c, "f32")
    qe = search.Embeddings(q, "f32", ce.data.device)
    idx, vals, s_pos = search.mine_negatives(qe, ce, 100, fill_masked_tail=False)
    rv, ri, rp = oracle.mine_negatives_ref(q, c, 100)
    assert torch.allclose(s_pos.cpu(), rp, rtol=2e-5, atol=1e-6)
    oracle.check_topk_parity(q, c, vals.cpu(), idx.cpu(), rtol=2e-5, upper_bound=rp * 0.95)
    for i in (0, 3, 7, 150):
        d_ref, d_got = {}, {}
        oracle.negatives_for_sample(i, d_ref, q.clone(), c.clone(), top=100)
        search.negatives_for_sample(i, d_got, q, ce, top=100)
        assert len(d_got[i]) == 100 and len(set(d_got[i])) == 100   # tail filled with masked columns, as argsort does
        sc = (c.double() @ q[i].double())
        ub = sc[i] * 0.95
        n_hi = int((sc <= ub + 2e-5 * abs(ub)).sum())               # survivors, boundary columns counted
        n_cmp = min(100, n_hi)
        got = torch.tensor(d_got[i][:n_cmp])
        while n_cmp and sc[got[n_cmp - 1]] > ub + 2e-5 * abs(ub):   # a boundary column fell on the masked side
            n_cmp -= 1
        got = got[:n_cmp]
        pad = torch.full((100 - n_cmp,), -1, dtype=torch.int64)
        vals = torch.cat([sc[got].float(), torch.full((100 - n_cmp,), float("-inf"))])
        oracle.check_topk_parity(q[i:i + 1], c, vals[None], torch.cat([got, pad])[None], rtol=2e-5,
                                 upper_bound=ub.float().reshape(1))
        same = sum(a == b for a, b in zip(d_ref[i][:n_cmp], d_got[i][:n_cmp]))
        assert same >= n_cmp - 2 or n_cmp < 100                     # identical unless a boundary column flipped
    d7 = {}
    search.negatives_for_sample(7, d7, q, ce, top=100)
    assert 7 in d7[7]                              # negative s_pos: the positive is not removed


def test_int8_search_recall_vs_fp32():
    from vllm_amd import search
    q, c = oracle.make_clustered(256, 5000, 512, seed=12, n_centres=32)
    c8 = search.Embeddings(c, "i8")
    q8 = search.Embeddings(q, "i8", c8.data.device)
    v, i, _ = search.topk(q8, c8, 10)
    # device quantiser == oracle quantiser, bit for bit
    qa, sa = oracle.quantize_i8(q)
    cb, sb = oracle.quantize_i8(c)
    assert torch.equal(q8.data.cpu(), qa) and torch.equal(q8.scale.cpu(), sa)
    assert torch.equal(c8.data.cpu(), cb) and torch.equal(c8.scale.cpu(), sb)
    rv, ri = oracle.topk_ref_i8(qa, sa, cb, sb, 10)
    assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu(), ri)
    # reported, not asserted tightly: recall of int8 vs fp32 top-10
    fv, fi = oracle.topk_ref(q, c, 10)
    rec = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10 for a, b in zip(i.cpu(), fi)])
    assert rec > 0.9


def test_merge_and_sharded_equivalence():
    """Row-sharding the corpus and merging per-shard lists == searching the whole corpus."""
    from vllm_amd import search, kernels
    q, c = oracle.make_clustered(500, 3000, 128, seed=2, n_centres=16)
    qe = search.Embeddings(q, "bf16")
    ce = search.Embeddings(c, "bf16", qe.data.device)
    full_v, full_i, _ = search.topk(qe, ce, 64)
    bounds = [0, 700, 1500, 1501, 3000]
    vs, ids = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        v, i, _ = search.topk(qe, ce.rows(lo, hi), 64, idx_offset=lo)
        vs.append(v)
        ids.append(i)
    mv, mi = kernels.merge_topk(torch.stack(vs), torch.stack(ids))
    assert torch.equal(mv, full_v) and torch.equal(mi, full_i)


def test_empty_and_error_paths():
    from vllm_amd import kernels, _lib
    dev = torch.device("cuda:0")
    q = torch.zeros((3, 64), dtype=torch.bfloat16, device=dev)
    v, i, _ = kernels.topk(q, torch.zeros((0, 64), dtype=torch.bfloat16, device=dev), 5)
    assert bool((i == -1).all()) and bool(torch.isinf(v).all())
    v, i, _ = kernels.topk(torch.zeros((0, 64), dtype=torch.bfloat16, device=dev), q, 5)
    assert v.shape == (0, 5)
    with pytest.raises(_lib.AbcError):
        kernels.topk(torch.zeros((3, 72), dtype=torch.bfloat16, device=dev), torch.zeros((4, 72), dtype=torch.bfloat16, device=dev), 5)
    with pytest.raises(_lib.AbcError):
        kernels.topk(q, q, 129)
    with pytest.raises(_lib.AbcError):
        kernels.topk(q.cpu(), q.cpu(), 2)


@pytest.mark.parametrize("compute,d", [("f32", 100), ("f32", 7), ("bf16", 100), ("bf16", 200), ("i8", 100)])
def test_any_embedding_width(compute, d):
    """The reference's get_topk_candidates / negatives_for_sample take any D (torch.matmul); the kernels take rows of whole
    128-byte pieces: Embeddings pads other widths with zero columns, which changes no score.  Exact-grid data (integers / 16):
    every score is exact in any summation order, so values AND indices equal the oracle's (ties: lower index first)."""
    from vllm_amd import search
    g = torch.Generator().manual_seed(d)
    q = torch.randint(-8, 9, (300, d), generator=g).float() / 16
    c = torch.randint(-8, 9, (1500, d), generator=g).float() / 16
    if compute == "i8":
        q[:, 0], c[:, 0] = 127 / 16, 127 / 16                      # per-row scale 1/16 exactly: quantised values = 16 x
        qa, sa = oracle.quantize_i8(q)
        cb, sb = oracle.quantize_i8(c)
        rv, ri = oracle.topk_ref_i8(qa, sa, cb, sb, 10)
    else:
        rv, ri = oracle.topk_ref(q, c, 10)
    qe = search.Embeddings(q, compute)
    ce = search.Embeddings(c, compute, qe.data.device)
    assert ce.shape[1] % (128 // ce.data.element_size()) == 0 and ce.shape[1] - d < 128
    v, i, _ = search.topk(qe, ce, 10)
    assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu(), ri)
    if compute != "i8":
        idx, vals, s_pos = search.mine_negatives(qe.rows(0, 200), ce, 10)
        mv, mi, mp = oracle.mine_negatives_ref(q[:200], c, 10)
        assert torch.equal(s_pos.cpu(), mp)
        keep = mi >= 0
        assert torch.equal(idx.cpu()[keep], mi[keep]) and torch.equal(vals.cpu()[keep], mv[keep])
        images = [(f"q{j}", q[j:j + 1]) for j in range(40)]
        cands = [(f"c{j}", c[j:j + 1]) for j in range(500)]
        assert search.get_topk_candidates(images, cands, 5, compute=compute) == oracle.get_topk_candidates(images, cands, 5)


def test_per_sample_calls_keep_the_resident_corpus():
    """The reference's loop passes the same corpus TENSOR to every negatives_for_sample call: its resident form is built once,
    rebuilt when the tensor changes (in place, or another tensor), and gives the lists of the explicit Embeddings form."""
    from vllm_amd import search
    q, c = oracle.make_clustered(64, 900, 128, seed=21, n_centres=4)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    search.release_resident()
    d1, d2 = {}, {}
    search.negatives_for_sample(3, d1, q, c, top=20, compute="bf16")
    first = search._resident_last[0][2]
    search.negatives_for_sample(5, d1, q, c, top=20, compute="bf16")
    assert search._resident_last[0][2] is first                       # kept
    ce = search.Embeddings(c, "bf16")
    search.negatives_for_sample(3, d2, q, ce, top=20, compute="bf16")
    search.negatives_for_sample(5, d2, q, ce, top=20, compute="bf16")
    assert d1 == d2
    c[7] = c[3]                                                         # in place: the version counter moves
    search.negatives_for_sample(3, d1, q, c, top=20, compute="bf16")
    assert search._resident_last[0][2] is not first
    search.negatives_for_sample(3, d2, q, search.Embeddings(c, "bf16"), top=20, compute="bf16")
    assert d1[3] == d2[3]
    search.release_resident()
    assert not search._resident_last
