"""Soak run of the search kernels against the oracle: seeded random problems for a time budget (default 240 s).

    python scripts/soak_random.py [seconds] [seed] [first case]

Exact-grid data (integers / 16, d <= 256: every score is exact in fp32 in any summation order, ties are everywhere), so
values AND indices must EQUAL the oracle's (ties: lower index first) for all three compute types, with and without the
0.95-positive bound, and for neg_mine.batch (top-k + compare-count): ragged query blocks and tiles, k from 1 to 128, corpora from one row to several hundred tiles per
split (bounds, refreshes, compactions, both descents).  Test infrastructure: uses oracle/ as the checker."""
import sys
import time

import torch

sys.path.insert(0, ".")
import oracle  # noqa: E402
from vllm_amd import search  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0             # skip the cases before this one (same random sequence)
rng = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))  # noqa: E731
dev = torch.device("cuda:0")
t0 = time.time()
case = 0
while time.time() - t0 < budget:
    compute = ("bf16", "f32", "i8")[case % 3]
    d = (64, 128, 256)[ri(0, 2)]
    nq = (ri(1, 300), ri(300, 2000), ri(7000, 9000))[ri(0, 2)]
    nc = (ri(1, 200), ri(200, 5000), ri(5000, 40000), ri(40000, 300000))[ri(0, 3)]
    k = (ri(1, 16), ri(16, 64), ri(65, 128))[ri(0, 2)]
    spread = (1, 2, 8)[ri(0, 2)]                                   # few distinct values -> many ties
    bound = case % 2 == 1 and nc >= nq
    batch_mode = case % 4 == 3 and compute != "i8" and nc >= nq and k <= nc
    i0, bs = (ri(0, nq - 1), ri(1, nq)) if batch_mode else (0, 0)
    if case < first:
        case += 1
        continue
    g = torch.Generator().manual_seed(seed * 100003 + case)
    q = torch.randint(-spread, spread + 1, (nq, d), generator=g).float() / 16
    c = torch.randint(-spread, spread + 1, (nc, d), generator=g).float() / 16
    if bound:                                                      # positives that ARE similar: q_i = c_i with a few flips
        q = c[:nq].clone()
        flip = torch.rand((nq, d), generator=g) < 0.1
        q[flip] = -q[flip]
    if compute == "i8":
        q[:, 0], c[:, 0] = 127 / 16, 127 / 16                      # per-row scale 1/16 exactly
    qe = search.Embeddings(q, compute, dev)
    ce = search.Embeddings(c, compute, dev)
    tag = (case, compute, nq, nc, d, k, spread, bound)
    if batch_mode:
        # neg_mine.batch (visualization/neg_mine.py:21-42): top-k, the fraction of candidates above the positive, the positive
        got = search.batch(qe, ce, i0, bs, k)
        ref = oracle.neg_mine_batch(q, c, i0, bs, k)
        ok = all(torch.equal(a, b) for a, b in zip(got, ref))
        print(("ok  " if ok else "FAIL"), "batch", tag, i0, bs, f"{time.time() - t0:.0f}s", flush=True)
        if not ok:
            for name, a, b in zip(("score", "idx", "frac", "pos"), got, ref):
                print(name, "equal" if torch.equal(a, b) else "DIFFERS")
                if not torch.equal(a, b):
                    rows = (a != b).reshape(a.shape[0], -1).any(dim=1).nonzero().flatten()
                    print(" rows", rows[:20].tolist(), "of", a.shape[0], "n =", len(rows))
                    r = int(rows[0])
                    print(" got", a[r].flatten()[:8].tolist(), "ref", b[r].flatten()[:8].tolist(), "x nc:", (a[r] * nc).flatten()[:4].tolist(), (b[r] * nc).flatten()[:4].tolist())
            sys.exit(1)
        case += 1
        continue
    if bound:
        idx, vals, s_pos = search.mine_negatives(qe, ce, k, fill_masked_tail=False)
        ub = (s_pos * 0.95).cpu()
    else:
        vals, idx, _ = search.topk(qe, ce, k)
        ub = None
    torch.cuda.synchronize()
    vals, idx = vals.cpu(), idx.cpu()
    if compute == "i8":
        rv, rix = oracle.topk_ref_i8(qe.data.cpu()[:, :d], qe.scale.cpu(), ce.data.cpu()[:, :d], ce.scale.cpu(), k, upper_bound=ub)
    else:
        rv, rix = oracle.topk_ref(q, c, k, upper_bound=ub)
    if rv.shape[1] < k:                                            # corpus shorter than k: the oracle returns nc columns
        pad = k - rv.shape[1]
        rv = torch.cat([rv, torch.full((nq, pad), float("-inf"))], dim=1)
        rix = torch.cat([rix, torch.full((nq, pad), -1, dtype=torch.int64)], dim=1)
    ok = torch.equal(idx, rix) and torch.equal(vals, rv)
    print(("ok  " if ok else "FAIL"), tag, f"{time.time() - t0:.0f}s", flush=True)
    if not ok:
        bad = (idx != rix).any(dim=1).nonzero().flatten()
        print("rows differing:", bad[:10].tolist(), "of", nq)
        r = int(bad[0]) if len(bad) else 0
        print("got", idx[r].tolist()[:16], vals[r].tolist()[:8])
        print("ref", rix[r].tolist()[:16], rv[r].tolist()[:8])
        sys.exit(1)
    case += 1
print(f"{case} cases, all equal to the oracle")
