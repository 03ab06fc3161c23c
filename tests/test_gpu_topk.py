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
