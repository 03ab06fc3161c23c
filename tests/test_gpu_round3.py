"""GPU tests for what the round-2 review found loosely tested:

    n1   int8 bit-exact against the oracle at cfg5's row length (d = 3584: 28 phases per tile, peeled first phase) and
         k = 100, with and without the 0.95 bound
    plan a single list per query with k > 64 (timing knob plan_qb >= 129): the published bound is the k-th best, not the 64th
"""
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("bound", [False, True])
def test_int8_bit_exact_at_cfg5_row_length(bound):
    """BASELINE configs[4] geometry (d = 3584, k = 100, per-vector scales) on 2 048 x 20 000: scores AND indices equal
    the int8 rule of SURVEY.md 8c computed on the CPU (exact integer sums through the fp64 BLAS), ties by smaller index."""
    from vllm_amd import search
    nq, nc, d, k = 2048, 20_000, 3584, 100
    q, c = oracle.make_clustered(nq, nc, d, seed=3584, n_centres=40)
    qe, ce = search.Embeddings(q, "i8", DEV), search.Embeddings(c, "i8", DEV)
    q8, qs, c8, cs = qe.data.cpu(), qe.scale.cpu(), ce.data.cpu(), ce.scale.cpu()
    rq, rs = oracle.quantize_i8(q)
    assert torch.equal(rq, q8) and torch.equal(rs, qs)                      # the quantiser itself, at this row length
    ub = None
    if bound:
        idx, vals, s_pos = search.mine_negatives(qe, ce, k, fill_masked_tail=False)
        sc_diag = (q8.double() * c8[:nq].double()).sum(1).float() * (qs * cs[:nq])
        assert torch.equal(s_pos.cpu(), sc_diag)                            # s_pos by the rule, bit for bit
        ub = s_pos.cpu() * 0.95
    else:
        vals, idx, _ = search.topk(qe, ce, k)
    torch.cuda.synchronize()
    rv, ri = oracle.topk_ref_i8(q8, qs, c8, cs, k, upper_bound=ub)
    assert torch.equal(vals.cpu(), rv)
    assert torch.equal(idx.cpu(), ri)


def test_single_list_plan_with_k_above_64():
    """with plan_qb >= 129 (one split, one list per query) and k > 64, an overflow compaction published the
    64th best instead of the k-th and the merge dropped ranks 65..k.  The knob is a timing experiment, but it is read in
    production builds: results must not depend on it."""
    from vllm_amd import kernels, _lib
    nq, nc, d, k = 33_000, 3000, 128, 128
    g = torch.Generator().manual_seed(5)
    u = torch.nn.functional.normalize(torch.randn(d, generator=g), dim=0)
    q = torch.nn.functional.normalize(u[None] + 0.01 * torch.randn(nq, d, generator=g) / d ** 0.5, dim=-1)
    alpha = torch.linspace(0.05, 0.95, nc)                                  # scores ascend with the row: every tile appends
    w = torch.randn(nc, d, generator=g)
    w = torch.nn.functional.normalize(w - (w @ u)[:, None] * u[None], dim=-1)
    c = torch.nn.functional.normalize(alpha[:, None] * u[None] + (1 - alpha ** 2).sqrt()[:, None] * w, dim=-1)
    q, c = q.to(torch.bfloat16).to(DEV), c.to(torch.bfloat16).to(DEV)
    v0, i0, _ = kernels.topk(q, c, k)
    lib = _lib.load()
    assert lib.abc_debug_set(b"plan_qb", 256) == 0
    try:
        v1, i1, _ = kernels.topk(q, c, k)
        torch.cuda.synchronize()
    finally:
        lib.abc_debug_set(b"plan_qb", 0)
    assert int(i1.min()) >= 0
    assert torch.equal(v1, v0) and torch.equal(i1, i0)
