"""Size-independent properties at the FULL size of BASELINE.json configs[2] (3.3 M x 4096 bf16, k = 64), where
the CPU oracle cannot follow: sortedness and uniqueness, returned scores against a recomputation from the
returned rows, the independent compare-count path as a checksum of exactness (exactly k-1 candidates beat
the k-th returned score), and shard + merge == single sweep bit for bit."""
import pytest
import torch

from vllm_amd import kernels

pytestmark = pytest.mark.gpu
NC, D, K, NQ = 3_300_000, 4096, 64, 1024


@pytest.fixture(scope="module")
def data():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(2024)
    c = torch.empty((NC, D), dtype=torch.bfloat16, device=dev)
    step = 1 << 17
    for s in range(0, NC, step):
        x = torch.randn((min(step, NC - s), D), generator=g, device=dev)
        c[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)
    # queries near corpus rows (real near-duplicates at the top of every list) plus plain random ones
    rows = torch.randint(0, NC, (NQ,), generator=g, device=dev)
    q = c[rows].float() + 0.02 * torch.randn((NQ, D), generator=g, device=dev)
    q[NQ // 2:] = torch.randn((NQ - NQ // 2, D), generator=g, device=dev)
    q = torch.nn.functional.normalize(q, dim=-1).to(torch.bfloat16)
    v, i, _ = kernels.topk(q, c, K)
    torch.cuda.synchronize()
    return q, c, rows, v, i


def test_sorted_unique_in_range(data):
    q, c, rows, v, i = data
    assert bool((v[:, 1:] <= v[:, :-1]).all())
    assert int(i.min()) >= 0 and int(i.max()) < NC
    assert bool((torch.sort(i, dim=1).values.diff(dim=1) > 0).all())            # no index twice
    tie = v[:, 1:] == v[:, :-1]
    assert bool((i[:, 1:][tie] > i[:, :-1][tie]).all())                         # ties: smaller index first
    assert bool((i[:NQ // 2, 0] == rows[:NQ // 2]).all())                       # the near-duplicate is found first


def test_scores_match_recomputation(data):
    q, c, rows, v, i = data
    for a in range(0, NQ, 128):
        cand = c[i[a:a + 128].reshape(-1)].float().view(-1, K, D)               # [128, K, D] gathered rows
        s = torch.bmm(cand, q[a:a + 128].float().unsqueeze(-1)).squeeze(-1)
        assert float((s - v[a:a + 128]).abs().max()) <= 2e-5


def test_compare_count_checksum(data):
    """The COUNT epilogue (neg_mine.batch's compare-count) shares only the contraction with the top-k
    machinery: #{j : score_ij > v_i,k} must equal the number of returned entries above the k-th."""
    q, c, rows, v, i = data
    kth = v[:, K - 1].contiguous()
    _, _, rank = kernels.topk(q, c, 1, rank_ref=kth)
    want = (v > kth[:, None]).sum(dim=1)
    assert torch.equal(rank, want.to(rank.dtype))
    _, _, rank0 = kernels.topk(q, c, 1, rank_ref=v[:, 0].contiguous())
    assert int(rank0.abs().max()) == 0


def test_sweep_is_repeatable(data):
    """Run after run the sweep returns the same bits: the shared bounds are read racily but only ever discard what
    cannot be in the result."""
    q, c, rows, v, i = data
    for _ in range(2):
        v2, i2, _ = kernels.topk(q, c, K)
        assert torch.equal(v2, v) and torch.equal(i2, i)


def test_shard_and_merge_equals_single_sweep(data):
    q, c, rows, v, i = data
    cut = 1_237_001                                                             # uneven, not a tile multiple
    parts_v, parts_i = [], []
    for lo, hi in ((0, cut), (cut, NC)):
        pv, pi, _ = kernels.topk(q, c[lo:hi], K, idx_offset=lo)
        parts_v.append(pv)
        parts_i.append(pi)
    mv, mi = kernels.merge_topk(torch.stack(parts_v), torch.stack(parts_i))
    assert torch.equal(mv, v) and torch.equal(mi, i)


def test_upper_bound_rule_at_full_size(data):
    q, c, rows, v, i = data
    ub = (v[:, 0] * 0.95).contiguous()                                          # the 0.95 * s_pos rule on the best hit
    v2, i2, _ = kernels.topk(q, c, K, upper_bound=ub)
    assert bool((v2 <= ub[:, None]).all())
    # what survives is the unbounded list minus its entries above the bound, as far as that list reaches
    for r in range(0, NQ, 37):
        keep = v[r] <= ub[r]
        n = int(keep.sum())
        assert torch.equal(i2[r, :n], i[r][keep]) and torch.equal(v2[r, :n], v[r][keep])


def test_int8_config_properties():
    """BASELINE configs[4] at its full size (int8, 10 M rows x d = 3584, k = 100, per-vector scales; 35.8 GB): sortedness, exact
    agreement of the returned scores with the int8 rule recomputed from the returned rows, the compare-count
    checksum, shard + merge == single sweep."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(77)
    n, d, k, nq = 10_000_000, 3584, 100, 512
    c8 = torch.empty((n, d), dtype=torch.int8, device=dev)
    cs = torch.empty((n,), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 18):
        x = torch.nn.functional.normalize(torch.randn((min(1 << 18, n - s), d), generator=g, device=dev), dim=-1)
        a, b = kernels.quantize_i8(x)
        c8[s:s + a.shape[0]], cs[s:s + a.shape[0]] = a, b
    q8, qs = kernels.quantize_i8(torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=dev), dim=-1))
    v, i, _ = kernels.topk(q8, c8, k, q_scale=qs, c_scale=cs)
    assert bool((v[:, 1:] <= v[:, :-1]).all()) and int(i.min()) >= 0 and int(i.max()) < n
    assert bool((torch.sort(i, dim=1).values.diff(dim=1) > 0).all())
    # score_ij = float(sum_k q_ik c_jk) * (q_scale_i * c_scale_j): the integer sum is exact in fp32 products of int8 (< 2^24
    # per partial?) -- recompute in int64 on the device, then the two fp32 multiplies in the rule's association
    for a in range(0, nq, 64):
        rows = c8[i[a:a + 64].reshape(-1)].view(-1, k, d).to(torch.int32)
        dots = (rows * q8[a:a + 64].to(torch.int32).unsqueeze(1)).sum(-1)
        ref = dots.to(torch.float32) * (qs[a:a + 64].unsqueeze(1) * cs[i[a:a + 64]])
        assert torch.equal(ref, v[a:a + 64])
    kth = v[:, k - 1].contiguous()
    _, _, rank = kernels.topk(q8, c8, 1, q_scale=qs, c_scale=cs, rank_ref=kth)
    assert torch.equal(rank, (v > kth[:, None]).sum(dim=1).to(rank.dtype))
    cut = 3_750_003
    pv0, pi0, _ = kernels.topk(q8, c8[:cut], k, q_scale=qs, c_scale=cs[:cut])
    pv1, pi1, _ = kernels.topk(q8, c8[cut:], k, q_scale=qs, c_scale=cs[cut:], idx_offset=cut)
    mv, mi = kernels.merge_topk(torch.stack([pv0, pv1]), torch.stack([pi0, pi1]))
    assert torch.equal(mv, v) and torch.equal(mi, i)
