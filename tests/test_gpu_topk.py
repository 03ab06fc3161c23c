"""GPU parity tests: HIP fused score-GEMM + top-k (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _run(q, c, k, **kw):
    from vllm_amd import kernels
    dev = _dev()
    ub = kw.pop("upper_bound", None)
    v, i, r = kernels.topk(q.to(dev), c.to(dev), k, upper_bound=None if ub is None else ub.to(dev), **kw)
    torch.cuda.synchronize()
    return v.cpu(), i.cpu(), r


@pytest.mark.parametrize("nq,nc,d,k", [(300, 1000, 128, 10), (256, 256, 64, 1), (1000, 5000, 256, 10),
                                        (37, 513, 128, 64), (700, 3001, 192, 100), (5, 40, 64, 64)])
def test_bf16_topk_parity(nq, nc, d, k):
    q, c = oracle.make_clustered(nq, nc, d, seed=nq + nc, n_centres=16)
    qb, cb = q.to(torch.bfloat16), c.to(torch.bfloat16)
    v, i, _ = _run(qb, cb, k)
    # bf16 inputs are exact in fp32/fp64; tolerance 1e-3 relative per BASELINE.json (fp32 accumulate
    # in MFMA order); in practice the scores agree to ~1e-6
    oracle.check_topk_parity(qb.float(), cb.float(), v, i, rtol=1e-3)
    oracle.check_topk_parity(qb.float(), cb.float(), v, i, rtol=2e-5)


def test_f32_topk_parity():
    q, c = oracle.make_clustered(500, 2500, 128, seed=9, n_centres=16)
    v, i, _ = _run(q, c, 10)
    oracle.check_topk_parity(q, c, v, i, rtol=2e-5)


def test_i8_topk_exact():
    q, c = oracle.make_clustered(300, 2000, 256, seed=21, n_centres=16)
    qa, sa = oracle.quantize_i8(q)
    cb, sb = oracle.quantize_i8(c)
    from vllm_amd import kernels
    dev = _dev()
    v, i, _ = kernels.topk(qa.to(dev), cb.to(dev), 100, q_scale=sa.to(dev), c_scale=sb.to(dev))
    rv, ri = oracle.topk_ref_i8(qa, sa, cb, sb, 100)
    assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu(), ri)     # bit-exact scores and indices


def _adversarial(nq, nc, d, seed=0):
    """Corpus ordered so that every query's scores ASCEND with the row index: every tile beats the
    running threshold (the epilogue's worst case)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.nn.functional.normalize(torch.randn(d, generator=g), dim=0)
    q = torch.nn.functional.normalize(u[None] + 0.01 * torch.randn(nq, d, generator=g) / d ** 0.5, dim=-1)
    alpha = torch.linspace(0.05, 0.95, nc)
    w = torch.randn(nc, d, generator=g)
    w = torch.nn.functional.normalize(w - (w @ u)[:, None] * u[None], dim=-1)
    c = alpha[:, None] * u[None] + (1 - alpha ** 2).sqrt()[:, None] * w
    return q, torch.nn.functional.normalize(c, dim=-1)


@pytest.mark.parametrize("k", [64, 128, 7])
def test_adversarial_ascending_scores(k):
    q, c = _adversarial(300, 6000, 128)
    qb, cb = q.to(torch.bfloat16), c.to(torch.bfloat16)
    v, i, _ = _run(qb, cb, k)
    oracle.check_topk_parity(qb.float(), cb.float(), v, i, rtol=1e-3)
    # the true top-k is (nearly) the last k rows
    assert float((i >= 6000 - 8 * k).float().mean()) > 0.9


def test_descending_and_constant_scores():
    q, c = _adversarial(100, 3000, 64, seed=1)
    cb = c.flip(0).contiguous().to(torch.bfloat16)               # best rows first: nothing passes after tile 1
    qb = q.to(torch.bfloat16)
    v, i, _ = _run(qb, cb, 32)
    oracle.check_topk_parity(qb.float(), cb.float(), v, i, rtol=1e-3)
    ones = torch.ones(700, 64, dtype=torch.bfloat16)             # all scores equal: ties -> smallest indices
    v, i, _ = _run(ones[:5], ones, 10)
    assert torch.equal(i, torch.arange(10).expand(5, 10)) and bool((v == 64).all())


def test_upper_bound_and_rank_through_abi():
    q, c = oracle.make_clustered(400, 2500, 128, seed=33, n_centres=8)
    qb, cb = q.to(torch.bfloat16), c.to(torch.bfloat16)
    s = oracle.scores_f32(qb.float(), cb.float())
    pos = s[torch.arange(400), torch.arange(400)].clone()
    from vllm_amd import kernels
    dev = _dev()
    v, i, r = kernels.topk(qb.to(dev), cb.to(dev), 64, upper_bound=(pos * 0.95).to(dev), rank_ref=pos.to(dev),
                           rank_self_offset=0)
    oracle.check_topk_parity(qb.float(), cb.float(), v.cpu(), i.cpu(), rtol=1e-3, upper_bound=pos * 0.95)
    s64 = oracle.scores_f32(qb.float(), cb.float(), fp64=True)
    p64 = pos.double()[:, None]
    off = torch.ones_like(s64, dtype=torch.bool)
    off[torch.arange(400), torch.arange(400)] = False
    lo = ((s64 > p64 + 1e-5 * p64.abs()) & off).sum(1)
    hi = ((s64 > p64 - 1e-5 * p64.abs()) & off).sum(1)
    assert bool(((r.cpu() >= lo) & (r.cpu() <= hi)).all())


@pytest.mark.parametrize("dtype,nq,nc,d,k", [("bf16", 8192, 60000, 64, 64), ("bf16", 8192, 40000, 128, 128),
                                             ("i8", 8192, 50000, 128, 100), ("f32", 4096, 30000, 32, 10)])
def test_long_sweeps_full_launch(dtype, nq, nc, d, k):
    """A full launch (32 query blocks x 8 splits, 15-30 tiles per split): queue flushes, tracked bounds with
    geometric refresh, overflow compactions and the list merge, all against the oracle; with and without the
    0.95-positive upper bound."""
    from vllm_amd import kernels
    dev = _dev()
    q, c = oracle.make_clustered(nq, nc, d, seed=nq + nc + k, n_centres=24)
    if dtype == "i8":
        qa, sa = oracle.quantize_i8(q)
        cb, sb = oracle.quantize_i8(c)
        s_pos = (qa[:, None, :].int() * cb[:nq][:, None, :].int()).sum(-1).squeeze(1).float() * (sa * sb[:nq])
        for ub in (None, s_pos * 0.95):
            v, i, _ = kernels.topk(qa.to(dev), cb.to(dev), k, q_scale=sa.to(dev), c_scale=sb.to(dev),
                                   upper_bound=None if ub is None else ub.to(dev))
            rv, ri = oracle.topk_ref_i8(qa, sa, cb, sb, k, upper_bound=ub)
            assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu(), ri)
        return
    if dtype == "bf16":
        q, c = oracle.bf16_round(q), oracle.bf16_round(c)
        qd, cd = q.to(torch.bfloat16).to(dev), c.to(torch.bfloat16).to(dev)
    else:
        qd, cd = q.to(dev), c.to(dev)
    s_pos = (q * c[:nq]).sum(-1)
    for ub in (None, s_pos * 0.95):
        v, i, _ = kernels.topk(qd, cd, k, upper_bound=None if ub is None else ub.to(dev))
        torch.cuda.synchronize()
        oracle.check_topk_parity(q, c, v.cpu(), i.cpu(), rtol=2e-5, upper_bound=ub)


def test_several_launches_per_call():
    """More queries than one launch takes (32 query blocks): per-launch offsets of queries, upper bounds, rank
    references and outputs; an uneven last launch."""
    from vllm_amd import kernels
    dev = _dev()
    nq, nc, d, k = 2 * 8192 + 300, 3000, 64, 10
    q, c = oracle.make_clustered(nq, nc, d, seed=4, n_centres=12)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    s_ref = (q * c[torch.arange(nq) % nc]).sum(-1)
    ub = s_ref * 0.95
    v, i, r = kernels.topk(q.to(torch.bfloat16).to(dev), c.to(torch.bfloat16).to(dev), k, upper_bound=ub.to(dev),
                           rank_ref=s_ref.to(dev))
    torch.cuda.synchronize()
    oracle.check_topk_parity(q, c, v.cpu(), i.cpu(), rtol=2e-5, upper_bound=ub)
    s = oracle.scores_f32(q, c, fp64=True)
    gap = (s - s_ref[:, None].double()).abs() > 2e-5 * s_ref[:, None].abs().double()
    want_lo = ((s > s_ref[:, None].double()) & gap).sum(1)
    want_hi = ((s > s_ref[:, None].double()) | ~gap).sum(1)
    assert bool(((r.cpu() >= want_lo) & (r.cpu() <= want_hi)).all())


@pytest.mark.parametrize("d", [3, 100, 128, 3584, 8192, 8196, 10000])
def test_quantize_i8_equals_the_oracle_at_any_width(d):
    """abc_quantize_i8 (rows of up to 8192 floats, a multiple of 4: one 16-byte-load pass held in registers, packed stores;
    anything else: two scalar passes) against the oracle's rule, values and scales bit for bit -- zero rows, one huge
    element, halves that round to even."""
    from vllm_amd import kernels
    g = torch.Generator().manual_seed(d)
    x = torch.randn((37, d), generator=g)
    x[3] = 0.0
    x[5, d // 2] = 1e4
    x[7] = torch.arange(d, dtype=torch.float32) % 255 - 127                      # exact grid: scale 1, nothing to round
    x[8] = (torch.arange(d, dtype=torch.float32) % 255 - 127) / 2                  # halves: rint goes to even
    q8, s8 = kernels.quantize_i8(x.to(torch.device("cuda:0")))
    rq, rs = oracle.quantize_i8(x)
    assert torch.equal(s8.cpu(), rs) and torch.equal(q8.cpu(), rq)


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 4099, 1 << 20])
def test_cast_bf16_equals_torch_rounding(n):
    """abc_cast_bf16 against torch's round-to-nearest-even, incl. ties, infinities and values that round up into the next
    exponent; an aligned and a misaligned buffer."""
    from vllm_amd import kernels
    g = torch.Generator().manual_seed(n)
    x = torch.randn((n + 1,), generator=g) * 3
    x[0] = 1.00390625                                   # exactly halfway between two bf16 values: to even
    if n > 8:
        x[1:8] = torch.tensor([float("inf"), -float("inf"), 0.0, -0.0, 3.3895313892515355e38, 1.0039062, 65280.0])
    xd = x.to(torch.device("cuda:0"))
    for view in (xd[:n], xd[1:]):                       # the second view starts 4 bytes into the buffer
        got = kernels.cast_bf16(view.reshape(1, -1))
        assert torch.equal(got.reshape(-1), view.to(torch.bfloat16))
