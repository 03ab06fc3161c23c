"""CPU oracle: exact similarity search (score = Q.C^T, per-row top-k, mining mask).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Citations are relative to
/root/reference (snapshot 2025-03-01).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = [
    "bf16_round", "scores_f32", "topk_ref", "get_topk_candidates", "recall_i2t", "recall_t2i",
    "negatives_for_sample", "mine_negatives_ref", "neg_mine_batch", "quantize_i8", "scores_i8",
    "topk_ref_i8", "check_topk_parity", "make_clustered", "topk_file_layout",
    "create_negative_index_ref", "attach_negative_rows", "eval_lines", "karpathy_paths",
]

NEG_INF = -math.inf


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round fp32 -> bf16 once (round-to-nearest-even) and return as fp32.

    bf16 oracle rule, SURVEY.md 8(c): never use torch CPU ``bf16 @ bf16`` (it
    returns bf16-rounded scores); upcast first and contract in fp32.
    """
    return x.to(torch.bfloat16).to(torch.float32)


def scores_f32(q: torch.Tensor, c: torch.Tensor, fp64: bool = False) -> torch.Tensor:
    """``query_stack @ candidate_stack.T`` -- evaluate/eval_mscoco.py:40, neg_mine.py:26."""
    if fp64:
        return (q.double() @ c.double().t())
    return q.float() @ c.float().t()


def _order_rows(vals: torch.Tensor, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sort each row by (value desc, index asc) -- the build's documented tie order."""
    # stable sort by index first, then stable sort by value descending
    o1 = torch.argsort(idx, dim=1, stable=True)
    vals, idx = torch.gather(vals, 1, o1), torch.gather(idx, 1, o1)
    o2 = torch.argsort(vals, dim=1, descending=True, stable=True)
    return torch.gather(vals, 1, o2), torch.gather(idx, 1, o2)


def topk_from_scores(scores: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-row top-k of a materialised score matrix, ties -> smaller index first.

    Same selection as ``torch.topk(scores, k, dim=1)`` (eval_mscoco.py:41,
    neg_mine.py:29) with the tie order made deterministic.  -inf entries (masked)
    are reported as (value -inf, index -1).
    """
    nq, nc = scores.shape
    kk = min(nc, k)
    margin = min(nc, kk + 32)
    v, i = torch.topk(scores, margin, dim=1)
    v, i = _order_rows(v, i)
    if margin < nc:
        # a tie straddling the margin would make the cut ambiguous: redo those rows in full
        amb = (v[:, margin - 1] == v[:, kk - 1]).nonzero().flatten()
        for r in amb.tolist():
            o = torch.argsort(scores[r], descending=True, stable=True)[:margin]
            v[r], i[r] = scores[r][o], o
    v, i = v[:, :kk].clone(), i[:, :kk].clone()
    if kk < k:
        v = torch.cat([v, torch.full((nq, k - kk), NEG_INF, dtype=v.dtype)], 1)
        i = torch.cat([i, torch.full((nq, k - kk), -1, dtype=i.dtype)], 1)
    i[v == NEG_INF] = -1
    return v, i


def topk_ref(q: torch.Tensor, c: torch.Tensor, k: int, upper_bound: Optional[torch.Tensor] = None,
             q_chunk: int = 1024, fp64: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact top-k of ``q @ c.T`` per query row; optional per-row upper bound.

    ``upper_bound[i]`` restates create_negative_index.py:48-49: every score
    strictly greater than the bound is set to -inf before selection.
    Returns (values [nq,k] float32/64, indices [nq,k] int64).
    """
    outs_v, outs_i = [], []
    for s in range(0, q.shape[0], q_chunk):
        sc = scores_f32(q[s:s + q_chunk], c, fp64=fp64)
        if upper_bound is not None:
            ub = upper_bound[s:s + q_chunk].to(sc.dtype).unsqueeze(1)
            sc = sc.masked_fill(sc > ub, NEG_INF)
        v, i = topk_from_scores(sc, k)
        outs_v.append(v)
        outs_i.append(i)
    if not outs_v:
        return torch.empty(0, k), torch.empty(0, k, dtype=torch.int64)
    return torch.cat(outs_v), torch.cat(outs_i)


# --------------------------------------------------------------------------------------
# a1 / a2: evaluate/eval_mscoco.py
# --------------------------------------------------------------------------------------
def get_topk_candidates(queries: Sequence[Tuple[object, torch.Tensor]],
                        candidates: Sequence[Tuple[object, torch.Tensor]], k: int = 3) -> Dict[object, list]:
    """Restates evaluate/eval_mscoco.py:18-50 (lines 36-48) minus ``.cuda()/.cpu()``.

    Dict keyed by query id, so duplicated ids collapse to the last occurrence (:45-48).
    """
    query_ids, query_embs = zip(*queries)
    candidate_ids, candidate_embs = zip(*candidates)
    query_stack = torch.cat(query_embs, dim=0)
    candidate_stack = torch.cat(candidate_embs, dim=0)
    scores = query_stack @ candidate_stack.T
    _, topk_indices = topk_from_scores(scores, k)
    results = {}
    for i, q_id in enumerate(query_ids):
        results[q_id] = [candidate_ids[int(idx)] for idx in topk_indices[i]]
    return results


def _intersect(l1, l2) -> bool:
    """evaluate/eval_mscoco.py:15-16."""
    return len(set.intersection(set(l1), set(l2))) > 0


def recall_i2t(test: List[dict], cand: Dict[object, list]) -> float:
    """evaluate/eval_mscoco.py:78-83: hit if any of the image's captions is retrieved."""
    acc = 0
    for x in test:
        targets = [i["raw"] for i in x["sentences"]]
        if _intersect(cand[x["image"]], targets):
            acc += 1
    return acc / len(test)


def recall_t2i(test: List[dict], cand: Dict[object, list]) -> float:
    """evaluate/eval_mscoco.py:89-98: hit if the caption's image is retrieved."""
    acc = cntr = 0
    for x in test:
        for t in x["sentences"]:
            if _intersect(cand[t["raw"]], [x["image"]]):
                acc += 1
            cntr += 1
    return acc / cntr


# --------------------------------------------------------------------------------------
# a3: pretrain/create_negative_index.py
# --------------------------------------------------------------------------------------
def negatives_for_sample(idx: int, negative_dict: dict, q: torch.Tensor, c: torch.Tensor, top: int = 100):
    """Restates pretrain/create_negative_index.py:45-52 minus ``.cuda()``.

    The reference argsort is unstable; this restatement uses a stable descending
    sort (ties -> smaller index), the documented order of the build.
    """
    query = q[idx, :].unsqueeze(dim=0)
    scores = (c @ query.t()).squeeze(dim=-1)
    score_threshold = scores[idx] * .95
    scores[scores > score_threshold] = NEG_INF
    sorted_indices = torch.argsort(scores, descending=True, stable=True)
    negative_dict[idx] = sorted_indices[:top].tolist()


def mine_negatives_ref(q: torch.Tensor, c: torch.Tensor, k: int, ratio: float = 0.95):
    """Batched form of a3: returns (values, indices, s_pos).  Masked/missing -> -1."""
    # the reference takes scores[idx] out of the GEMV itself (create_negative_index.py:48)
    sc = scores_f32(q, c)
    s_pos = sc.diagonal().clone()
    v, i = topk_ref(q, c, k, upper_bound=s_pos * ratio)
    return v, i, s_pos


# --------------------------------------------------------------------------------------
# a5: visualization/neg_mine.py
# --------------------------------------------------------------------------------------
def neg_mine_batch(query: torch.Tensor, cand: torch.Tensor, idx: int, batch_size: int, top_k: int):
    """Restates visualization/neg_mine.py:21-42 (no print, deterministic tie order)."""
    batch_idx = idx + batch_size if idx + batch_size < query.shape[0] else query.shape[0]
    q_slice = query[idx:batch_idx, :]
    scores = q_slice @ cand.t()
    score, topk_idx = topk_from_scores(scores, top_k)
    c_slice = cand[idx:batch_idx, :].unsqueeze(-1)
    scores_for_answer = torch.bmm(q_slice.unsqueeze(1), c_slice).squeeze(dim=-1)
    mask_sum = (scores > scores_for_answer).sum(dim=-1)
    mask_sum_norm = mask_sum / cand.shape[0]
    return score, topk_idx, mask_sum_norm, scores_for_answer.squeeze(dim=-1)


def topk_file_layout(scores: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``top_k.pt`` layout, neg_mine.py:69-75: float32 [2,Nq,k]; [0]=scores, [1]=indices as float."""
    return torch.stack((scores.float(), idx.float()))


# --------------------------------------------------------------------------------------
# int8 rule (defined by the build, SURVEY.md 8(c): not in the reference)
# --------------------------------------------------------------------------------------
def quantize_i8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric per-vector absmax: scale=max|x|/127 (1 if the row is all zero),
    q=clamp(rint(x/scale),-127,127) with round-half-even (torch.round)."""
    x = x.float()
    amax = x.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    qv = torch.clamp(torch.round(x / scale.unsqueeze(1)), -127, 127).to(torch.int8)
    return qv, scale


def scores_i8(qa: torch.Tensor, sa: torch.Tensor, qb: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    """score_ij = float(sum_k qa_ik*qb_jk) * (sa_i * sb_j); int32 sum exact, two fp32 multiplies."""
    assert qa.shape[1] * 127 * 127 < 2 ** 31
    if qa.shape[0] * qb.shape[0] * qa.shape[1] > (1 << 30):
        # large cases: the same exact integer sums through the fp64 BLAS (|sum| < 2^31 << 2^53, every partial sum an exact
        # integer), converted to fp32 exactly as the int32 sum would be (one round-to-nearest-even)
        acc = qa.double() @ qb.double().t()
    else:
        acc = qa.to(torch.int32) @ qb.to(torch.int32).t()
    return acc.to(torch.float32) * (sa.float().unsqueeze(1) * sb.float().unsqueeze(0))


def topk_ref_i8(qa, sa, qb, sb, k: int, upper_bound=None, q_chunk: int = 512):
    outs_v, outs_i = [], []
    for s in range(0, qa.shape[0], q_chunk):
        sc = scores_i8(qa[s:s + q_chunk], sa[s:s + q_chunk], qb, sb)
        if upper_bound is not None:
            sc = sc.masked_fill(sc > upper_bound[s:s + q_chunk].unsqueeze(1), NEG_INF)
        v, i = topk_from_scores(sc, k)
        outs_v.append(v)
        outs_i.append(i)
    return torch.cat(outs_v), torch.cat(outs_i)


# --------------------------------------------------------------------------------------
# tolerance-aware parity checker (SURVEY.md 7.3 "Ties", 7.4 item 3)
# --------------------------------------------------------------------------------------
def check_topk_parity(q: torch.Tensor, c: torch.Tensor, got_val: torch.Tensor, got_idx: torch.Tensor,
                      rtol: float, upper_bound: Optional[torch.Tensor] = None, idx_offset: int = 0,
                      scores: Optional[torch.Tensor] = None, ub_slack: bool = True, atol: float = 0.0) -> dict:
    """Compare a device top-k against the fp64 score matrix under the gap rule.

    Index parity is asserted only where score gaps exceed ``rtol*|score| + atol`` (atol, default 0, is for scores near
    zero, where fp32 accumulation of unit vectors is off by ~1e-7 absolute whatever the score):
      1. every returned index is a valid, unmasked column whose true score is within tol of
         the true k-th best (nothing clearly worse was returned);
      2. every column whose true score beats the true k-th best by more than tol is returned
         (nothing clearly better was missed);
      3. returned values equal the true scores of the returned indices within tol;
      4. returned values are non-increasing; equal returned values are in ascending index order;
      5. no duplicates among returned indices.
    With ``rtol == 0`` this is exact equality with ``topk_ref`` including tie order.
    Returns counters; raises AssertionError on violation.
    """
    nq, k = got_val.shape
    if scores is None:
        scores = scores_f32(q, c, fp64=True)
    scores = scores.double()
    nc = scores.shape[1]
    tol = lambda s: rtol * s.abs() + atol  # noqa: E731
    masked = None
    if upper_bound is not None:
        ub = upper_bound.double().unsqueeze(1)
        # columns within tol of the bound may fall either side on the device
        clearly_masked = scores > ub + (tol(ub) if ub_slack else 0)
        maybe_masked = scores > ub - (tol(ub) if ub_slack else 0)
        masked = clearly_masked
    sc_lo = scores.clone()          # pessimistic: ambiguous columns masked
    sc_hi = scores.clone()          # optimistic: ambiguous columns kept
    if masked is not None:
        sc_hi[clearly_masked] = NEG_INF
        sc_lo[maybe_masked] = NEG_INF
    kk = min(k, nc)
    kth_lo = torch.topk(sc_lo, kk, dim=1).values[:, kk - 1]     # k-th best if ambiguous are dropped
    kth_hi = torch.topk(sc_hi, kk, dim=1).values[:, kk - 1]
    gi = got_idx.long() - idx_offset
    valid = gi >= 0
    n_valid = valid.sum(1)
    stats = {"rows": nq, "exact_rows": 0}
    gi_c = gi.clamp(min=0)
    true_of_got = torch.gather(scores, 1, gi_c)
    gv = got_val.double()
    for r in range(nq):
        nv = int(n_valid[r])
        assert bool(valid[r, :nv].all()) and not bool(valid[r, nv:].any()), f"row {r}: holes in the list"
        ids = gi[r, :nv]
        assert ids.unique().numel() == nv, f"row {r}: duplicate indices"
        assert bool((ids < nc).all()), f"row {r}: index out of range"
        ts = true_of_got[r, :nv]
        if masked is not None:
            assert not bool(clearly_masked[r, ids].any()), f"row {r}: returned a masked column"
        # (3) values
        assert bool(((gv[r, :nv] - ts).abs() <= tol(ts) + 1e-30).all()), \
            f"row {r}: value mismatch max {float((gv[r, :nv] - ts).abs().max())}"
        assert bool((gv[r, nv:] == NEG_INF).all()), f"row {r}: tail not -inf"
        # (4) order
        d = gv[r, 1:nv] - gv[r, :nv - 1]
        assert bool((d <= 0).all()), f"row {r}: not sorted"
        eq = d == 0
        assert bool((ids[1:][eq] > ids[:-1][eq]).all()), f"row {r}: tie order"
        # (1) nothing clearly worse
        n_surv_lo = int((sc_lo[r] > NEG_INF).sum())
        n_surv_hi = int((sc_hi[r] > NEG_INF).sum())
        assert min(k, n_surv_lo) <= nv <= min(k, n_surv_hi), f"row {r}: {nv} entries, survivors {n_surv_lo}..{n_surv_hi}"
        if nv:
            thr = kth_lo[r] if kth_lo[r] > NEG_INF else ts.min()
            assert bool((ts >= thr - tol(thr)).all()), f"row {r}: returned a clearly worse column"
        # (2) nothing clearly better missed
        if kth_hi[r] > NEG_INF:
            must = (sc_lo[r] > kth_hi[r] + tol(kth_hi[r])).nonzero().flatten()
            got_set = set(ids.tolist())
            miss = [m for m in must.tolist() if m not in got_set]
            assert not miss, f"row {r}: missed clearly better columns {miss[:5]}"
    return stats


# --------------------------------------------------------------------------------------
# synthetic data of SURVEY.md 8(d): clustered, L2-normalised, row i of C is the positive of q_i
# --------------------------------------------------------------------------------------
def make_clustered(nq: int, nc: int, d: int, seed: int = 1234, n_centres: int = 64,
                   c_noise: float = 0.7, q_noise: float = 0.5, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(n_centres, d, generator=g)
    c = mu[torch.arange(nc) % n_centres] + c_noise * torch.randn(nc, d, generator=g)
    c = torch.nn.functional.normalize(c, dim=-1)
    base = c[torch.arange(nq) % nc]
    # noise scaled by 1/sqrt(d) so that s_pos = q_i.c_i ~ 1/sqrt(1+q_noise^2) (0.89 at 0.5)
    q = torch.nn.functional.normalize(base + q_noise * torch.randn(nq, d, generator=g) / math.sqrt(d), dim=-1)
    return q.to(dtype), c.to(dtype)


# --------------------------------------------------------------------------------------
# a4 / f1: pretrain/create_negative_index.py main(), and the negatives.json consumer
# --------------------------------------------------------------------------------------
def create_negative_index_ref(q: torch.Tensor, c: torch.Tensor, top: int = 100) -> str:
    """Restates create_negative_index.py:33-43: the per-sample loop over all rows and the text that
    json.dump(negative_dict, f) writes (int keys become strings, default separators)."""
    import json
    negative_dict: dict = {}
    for idx in range(q.shape[0]):
        negatives_for_sample(idx, negative_dict, q, c.clone(), top=top)
    return json.dumps(negative_dict)


def attach_negative_rows(negative_meta: dict, idx: int, negatives: int) -> list:
    """Restates dataset_utils/conceptual_captions.py:150-156 up to the row lookup (uses the global
    np.random state, as the reference does)."""
    import numpy as np
    offset = 5
    starts = np.arange(negatives) * offset
    offsets = np.random.randint(0, offset, size=negatives)
    neg_idx = starts + offsets
    negatives_for_idx = negative_meta[str(idx)]
    return [negatives_for_idx[i] for i in neg_idx]


# --------------------------------------------------------------------------------------
# f2: evaluate/eval_mscoco.py:56-99 / eval_flickr30k.py:56-100 (everything after the embedding calls)
# --------------------------------------------------------------------------------------
def eval_lines(test: List[dict], images: list, text: list) -> List[str]:
    """The six lines the eval scripts print, computed the reference's way: one get_topk_candidates per
    direction and k."""
    lines = []
    for topk in [1, 5, 10]:
        cand = get_topk_candidates(images, text, topk)
        lines.append(f"i2t top{topk} is {recall_i2t(test, cand):.4f}")
    for topk in [1, 5, 10]:
        cand = get_topk_candidates(text, images, topk)
        lines.append(f"t2i top{topk} is {recall_t2i(test, cand):.4f}")
    return lines


def karpathy_paths(json_path: str, image_dir=None) -> List[dict]:
    """eval_mscoco.py:58-67 / eval_flickr30k.py:58-68: test split with x["image"] filled in."""
    import json
    import os
    with open(json_path, "rb") as f:
        data = json.loads(f.read())["images"]
    test = list(filter(lambda x: x["split"] == "test", data))
    for x in test:
        if image_dir is None:
            x["image"] = os.path.join(os.path.dirname(json_path), x["filepath"], x["filename"])
        else:
            x["image"] = os.path.join(os.path.dirname(json_path), image_dir, x["filename"])
    return test
