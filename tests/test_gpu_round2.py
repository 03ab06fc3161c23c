"""GPU tests for rows the round-1 review found untested or loosely tested:

    a6   search.compute_topk and the top_k.pt / relative_scoring.pt / absolute_scoring.pt layout
         (visualization/neg_mine.py:44-77)
    a3   the positive score comes out of the SAME arithmetic as the tile (create_negative_index.py:47-49):
         abc_rowdot_* is bit-identical to the score abc_topk_* reports for column i of query i, so the
         0.95 * s_pos mask needs no slack in the checker
    b    boundary hygiene: one workspace per (device, stream)
"""
import os

import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_compute_topk_file_layout(tmp_path):
    """compute_topk on saved query.pt / cand.pt: top_k.pt is ONE float32 [2, Nq, k] tensor, [0] = scores, [1] =
    indices cast to float (neg_mine.py:69-75 -- note the swapped variable names there), relative_scoring.pt [Nq],
    absolute_scoring.pt [Nq]; equal to oracle.topk_file_layout over oracle.neg_mine_batch, batch by batch."""
    from vllm_amd import mining, search
    nq, d, k, bs = 700, 128, 10, 300                   # 3 batches, the last one ragged (neg_mine.py:60-66)
    q, c = oracle.make_clustered(nq, nq, d, seed=44, n_centres=12)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)  # exact in fp32: the f32 kernel sees the oracle's numbers
    path = str(tmp_path)
    mining.save_embeddings(path, q, c)
    search.compute_topk(path, k, compute="f32", batch_size=bs)
    top = torch.load(os.path.join(path, "top_k.pt"), weights_only=True)
    rel = torch.load(os.path.join(path, "relative_scoring.pt"), weights_only=True)
    ab = torch.load(os.path.join(path, "absolute_scoring.pt"), weights_only=True)
    assert top.dtype == torch.float32 and tuple(top.shape) == (2, nq, k)
    assert rel.dtype == torch.float32 and tuple(rel.shape) == (nq,) and tuple(ab.shape) == (nq,)
    sc, ix, fr, ps = [], [], [], []
    for i in range(0, nq, bs):
        s, t, f, p = oracle.neg_mine_batch(q, c, i, bs, k)
        sc.append(s), ix.append(t), fr.append(f), ps.append(p)
    want = oracle.topk_file_layout(torch.cat(sc), torch.cat(ix))
    assert want.dtype == top.dtype and want.shape == top.shape
    assert torch.allclose(top[0], want[0], rtol=2e-5, atol=1e-6)
    oracle.check_topk_parity(q, c, top[0], top[1].long(), rtol=2e-5)       # [1] really holds the indices
    assert float((top[1] == want[1]).float().mean()) > 0.999               # identical except inside fp32 ties
    assert torch.equal(top[1], top[1].round()) and int(top[1].min()) >= 0 and int(top[1].max()) < nq
    assert torch.allclose(ab, torch.cat(ps), rtol=2e-5, atol=1e-6)
    # relative_scoring.pt * Nc = #{j : s_ij > s_pos_i}: bounded by the fp64 score gaps like every other index result --
    # at least the columns clearly above the positive (by more than tol), at most those not clearly below it (the own
    # column sits inside the tolerance: neg_mine.py:37 counts it when matmul and bmm round differently, the build never)
    cnt, want_cnt = torch.round(rel.double() * nq), torch.round(torch.cat(fr).double() * nq)
    s64 = q.double() @ c.double().t()
    p64 = (q.double() * c.double()[:nq]).sum(1)
    tol = 2e-5 * p64.abs() + 1e-7
    lo = (s64 > (p64 + tol)[:, None]).sum(1).double()
    hi = (s64 > (p64 - tol)[:, None]).sum(1).double()
    assert bool(((cnt >= lo) & (cnt <= hi)).all()) and bool(((want_cnt >= lo) & (want_cnt <= hi)).all())
    assert float((cnt == want_cnt).float().mean()) > 0.98
    # the exact int64 indices the build writes beside the reference's float file
    idx64 = torch.load(os.path.join(path, "top_k_idx.pt"), weights_only=True)
    assert idx64.dtype == torch.int64 and torch.equal(idx64.float(), top[1])


@pytest.mark.parametrize("compute,d", [("bf16", 4096), ("f32", 4096), ("i8", 3584), ("bf16", 64)])
def test_positive_score_is_the_tile_score(compute, d):
    """s_pos[i] from abc_rowdot_* == the score abc_topk_* holds for (query i, column i), bit for bit: with the bound
    at exactly s_pos (ratio 1.0) the positive survives and every exact duplicate of it ties with it; nothing scoring
    above it does."""
    from vllm_amd import search
    n = 96
    q, c = oracle.make_clustered(n, n, d, seed=d + len(compute), n_centres=4, q_noise=0.2)
    c[40] = c[7]                                       # an exact duplicate of a positive
    qe = search.Embeddings(q, compute, DEV)
    ce = search.Embeddings(c, compute, DEV)
    s_pos = search.positive_scores(qe, ce)
    v, i, _ = search.topk(qe, ce, n)                   # k = nc: every column's score comes back
    tile = torch.gather(v, 1, (i == torch.arange(n, device=DEV)[:, None]).float().argmax(1, keepdim=True)).squeeze(1)
    assert torch.equal(tile, s_pos), float((tile - s_pos).abs().max())
    idx, vals, sp = search.mine_negatives(qe, ce, 16, ratio=1.0, fill_masked_tail=False)
    assert torch.equal(sp, s_pos)
    assert bool((vals[:, 0] == s_pos).all())           # the best survivor scores exactly s_pos ...
    assert bool((vals <= s_pos[:, None]).all())
    first = idx[:, 0].cpu()
    assert bool(((first == torch.arange(n)) | ((torch.arange(n) == 40) & (first == 7))).all())   # ... and is the positive
    assert idx[7, :2].tolist() == [7, 40] and idx[40, :2].tolist() == [7, 40]                  # duplicates tie, index order


def test_mining_bound_without_slack():
    """The 0.95 rule with the checker's slack switched off: the bound is compared against fp64 scores exactly.  A
    candidate may only differ where its fp64 score and the fp32 tile score fall on different sides of the bound,
    which needs |score - bound| < fp32 rounding (none in this data; asserted)."""
    from vllm_amd import search
    q, c = oracle.make_clustered(512, 2048, 1024, seed=77, n_centres=16)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    for compute in ("bf16", "f32"):
        qe = search.Embeddings(q, compute, DEV)
        ce = search.Embeddings(c, compute, DEV)
        idx, vals, s_pos = search.mine_negatives(qe, ce, 64, fill_masked_tail=False)
        ub = (s_pos * 0.95).cpu()
        s64 = oracle.scores_f32(q, c, fp64=True)
        assert float((s64 - ub.double()[:, None]).abs().min()) > 1e-6       # no score within rounding of its bound
        oracle.check_topk_parity(q, c, vals.cpu(), idx.cpu(), rtol=2e-5, upper_bound=ub, scores=s64, ub_slack=False)


def test_workspace_is_per_stream():
    """kernels.workspace(): calls on different torch streams must not share candidate lists."""
    from vllm_amd import kernels
    dev = torch.device(DEV)
    a = kernels.workspace(1024, dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        b = kernels.workspace(1024, dev)
        assert kernels.workspace(512, dev).data_ptr() == b.data_ptr()       # grow-only within a stream
    assert a.data_ptr() != b.data_ptr()
    assert kernels.workspace(1024, dev).data_ptr() == a.data_ptr()
    # two searches in flight on two streams give the results of two searches one after the other
    q, c = oracle.make_clustered(600, 4000, 128, seed=3, n_centres=8)
    qd, cd = q.to(torch.bfloat16).to(dev), c.to(torch.bfloat16).to(dev)
    q2 = qd.flip(0).contiguous()
    want1, want2 = kernels.topk(qd, cd, 32)[:2], kernels.topk(q2, cd, 32)[:2]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s1):
        got1 = kernels.topk(qd, cd, 32)[:2]
    with torch.cuda.stream(s2):
        got2 = kernels.topk(q2, cd, 32)[:2]
    torch.cuda.synchronize()
    assert torch.equal(got1[0], want1[0]) and torch.equal(got1[1], want1[1])
    assert torch.equal(got2[0], want2[0]) and torch.equal(got2[1], want2[1])


@pytest.mark.parametrize("compute,d", [("bf16", 1024), ("f32", 256), ("i8", 3584)])
def test_packed_shards_merge_to_the_single_sweep(compute, d):
    """abc_topk_packed per corpus shard + abc_topk_merge_packed == abc_topk_* over the whole corpus, bit for bit
    (values AND indices, ties included: the packed key orders equal scores by global index), with the 0.95 bound on;
    and the packed form of one shard decodes to what the unpacked entry returns for it."""
    from vllm_amd import kernels, search
    nq, nc, k = 300, 5000, 64
    q, c = oracle.make_clustered(nq, nc, d, seed=5 + d, n_centres=10)
    c[1200] = c[3900] = c[17]                          # exact ties across shards
    qe = search.Embeddings(q, compute, DEV)
    ce = search.Embeddings(c, compute, DEV)
    ub = search.positive_scores(qe, ce) * 0.95
    want_v, want_i, _ = kernels.topk(qe.data, ce.data, k, upper_bound=ub, q_scale=qe.scale, c_scale=ce.scale)
    cuts = [0, 1300, 1301, 4000, nc]                   # ragged shards, one of a single row (shorter than k)
    keys = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sh = ce.rows(lo, hi)
        kp = kernels.topk_packed(qe.data, sh.data, k, upper_bound=ub, idx_offset=lo, q_scale=qe.scale, c_scale=sh.scale)
        assert kp.dtype == torch.int64 and tuple(kp.shape) == (nq, k)
        v1, i1, _ = kernels.topk(qe.data, sh.data, k, upper_bound=ub, idx_offset=lo, q_scale=qe.scale, c_scale=sh.scale)
        dv, di = kernels.merge_packed(kp[None])
        assert torch.equal(di, i1) and torch.equal(dv, v1)
        keys.append(kp)
    got_v, got_i = kernels.merge_packed(torch.stack(keys))
    assert torch.equal(got_i, want_i)
    assert torch.equal(got_v, want_v)
    with pytest.raises(Exception):                     # a global index that does not fit the key's 32 bits
        kernels.topk_packed(qe.data, ce.data, k, idx_offset=(1 << 32) - 10, q_scale=qe.scale, c_scale=ce.scale)


@pytest.mark.parametrize("compute", ["bf16", "f32", "i8"])
def test_rank_count_excludes_the_own_column_only(compute):
    """neg_mine.batch's rank (neg_mine.py:21-42): columns scoring above rank_ref, the query's own positive column
    excluded.  With a reference 0.01 BELOW the positive's score the own column would count: the count with the exclusion
    is exactly one less than without it, for every query, wherever the own columns lie (three different tiles here), and
    both agree with the fp64 count outside the rounding gap.  (The kernel tests indices only in the one tile per sweep
    that holds some query's own column.)"""
    from vllm_amd import kernels, search
    nq, nc, d, off = 400, 2500, 128 if compute != "i8" else 256, 700
    q, c = oracle.make_clustered(nq, nc, d, seed=61, n_centres=8)
    c[off:off + nq] = torch.nn.functional.normalize(q + 0.3 * torch.randn(nq, d, generator=torch.Generator().manual_seed(5)), dim=-1)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    qe, ce = search.Embeddings(q, compute, DEV), search.Embeddings(c, compute, DEV)
    s_pos = search.positive_scores(qe, ce, off)
    ref = (s_pos - 0.01).contiguous()
    kw = dict(q_scale=qe.scale, c_scale=ce.scale)
    _, _, r_ex = kernels.topk(qe.data, ce.data, 8, rank_ref=ref, rank_self_offset=off, **kw)
    _, _, r_all = kernels.topk(qe.data, ce.data, 8, rank_ref=ref, **kw)
    assert torch.equal(r_all - r_ex, torch.ones_like(r_ex))
    if compute != "i8":
        s = oracle.scores_f32(q, c, fp64=True)
        own = torch.zeros_like(s, dtype=torch.bool)
        own[torch.arange(nq), off + torch.arange(nq)] = True
        gap = (s - ref.cpu().double()[:, None]).abs() > 2e-5
        lo = ((s > ref.cpu().double()[:, None]) & gap & ~own).sum(1)
        hi = (((s > ref.cpu().double()[:, None]) | ~gap) & ~own).sum(1)
        assert bool(((r_ex.cpu() >= lo) & (r_ex.cpu() <= hi)).all())


@pytest.mark.parametrize("compute,k", [("bf16", 64), ("bf16", 128), ("i8", 100), ("f32", 64)])
def test_ascending_scores_over_many_tiles_per_split(compute, k):
    """The epilogue's worst case on a FULL launch (32 query blocks x 8 splits, 12 tiles per split): every query's scores
    ascend with the row index, so every tile beats every bound -- dense blocks (wave-uniform append loop, queue full,
    direct appends), list overflow (a compaction every other tile) and the selecting merge on long unsorted lists all
    run.  The small adversarial test of round 1 had one tile per split."""
    from vllm_amd import search
    nq, nc, d = 8192, 96 * 256 + 77, 128 if compute != "i8" else 256
    g = torch.Generator().manual_seed(9)
    u = torch.nn.functional.normalize(torch.randn(d, generator=g), dim=0)
    q = torch.nn.functional.normalize(u[None] + 0.01 * torch.randn(nq, d, generator=g) / d ** 0.5, dim=-1)
    alpha = torch.linspace(0.05, 0.95, nc)
    w = torch.randn(nc, d, generator=g)
    w = torch.nn.functional.normalize(w - (w @ u)[:, None] * u[None], dim=-1)
    c = torch.nn.functional.normalize(alpha[:, None] * u[None] + (1 - alpha ** 2).sqrt()[:, None] * w, dim=-1)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    qe, ce = search.Embeddings(q, compute, DEV), search.Embeddings(c, compute, DEV)
    v, i, _ = search.topk(qe, ce, k)
    torch.cuda.synchronize()
    if compute == "i8":
        rv, ri = oracle.topk_ref_i8(qe.data.cpu(), qe.scale.cpu(), ce.data.cpu(), ce.scale.cpu(), k)
        assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu(), ri)
    else:
        sub = torch.arange(0, nq, 16)                      # 512 queries through the fp64 checker
        oracle.check_topk_parity(q[sub], c, v.cpu()[sub], i.cpu()[sub], rtol=2e-5)
        # every query: k distinct in-range indices, scores sorted, and (nearly) the last rows of the corpus
        ic = i.cpu()
        assert int(ic.min()) >= 0 and int(ic.max()) < nc
        assert bool((ic.sort(dim=1).values.diff(dim=1) > 0).all())
        assert bool((v.cpu().diff(dim=1) <= 0).all())
        assert float((ic >= nc - 8 * k).float().mean()) > 0.9


def test_random_shapes_against_the_oracle():
    """Twenty seeded random problems (ragged query blocks and tiles, k from 1 to 128, corpora shorter than k, with and
    without the 0.95 bound, all three compute types) through search.topk / mine_negatives against the fp64 checker."""
    from vllm_amd import search
    rng = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))  # noqa: E731
    for case in range(20):
        compute = ("bf16", "f32", "i8")[case % 3]
        d = (64, 128, 256)[ri(0, 2)] * (2 if compute == "i8" else 1)
        nq = (ri(1, 300), ri(300, 2000), ri(7000, 9000))[ri(0, 2)]
        nc = (ri(1, 200), ri(200, 5000), ri(5000, 40000))[ri(0, 2)]
        k = (ri(1, 16), ri(16, 64), ri(65, 128))[ri(0, 2)]
        q, c = oracle.make_clustered(nq, nc, d, seed=1000 + case, n_centres=ri(1, 24))
        q, c = oracle.bf16_round(q), oracle.bf16_round(c)
        qe, ce = search.Embeddings(q, compute, DEV), search.Embeddings(c, compute, DEV)
        bound = case % 2 == 1 and nc >= nq
        tag = (case, compute, nq, nc, d, k, bound)
        if bound:
            idx, vals, s_pos = search.mine_negatives(qe, ce, k, fill_masked_tail=False)
            ub = (s_pos * 0.95).cpu()
        else:
            vals, idx, _ = search.topk(qe, ce, k)
            ub = None
        torch.cuda.synchronize()
        vals, idx = vals.cpu(), idx.cpu()
        assert tuple(idx.shape) == (nq, k), tag
        if compute == "i8":
            rv, rix = oracle.topk_ref_i8(qe.data.cpu(), qe.scale.cpu(), ce.data.cpu(), ce.scale.cpu(), k, upper_bound=ub)
            assert torch.equal(vals, rv) and torch.equal(idx, rix), tag
        else:
            sub = torch.arange(0, nq, max(nq // 256, 1))
            oracle.check_topk_parity(q[sub], c, vals[sub], idx[sub], rtol=2e-5 if compute == "f32" else 1e-3,
                                     upper_bound=None if ub is None else ub[sub], ub_slack=False, atol=5e-7)
