"""The allocation-free forward (abc_infonce_fwd_noalloc) against the workspace form (abc_infonce_fwd), through the
C ABI: identical loss / logsumexp / argmax bits on two calls in a row, the second with a smaller input (the library's
device array keeps the first call's contents), and a refusal beyond its stated maximum."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, q, p, t, eps):
    dev = q.device
    B, d = q.shape
    C = p.shape[0]
    st = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for noalloc in (False, True):
        loss = torch.full((), float("nan"), device=dev)
        lse = torch.full((B,), float("nan"), device=dev)
        am = torch.full((B,), -7, dtype=torch.int32, device=dev)
        if noalloc:
            rc = lib.abc_infonce_fwd_noalloc(q.data_ptr(), p.data_ptr(), t.data_ptr(), eps, B, C, d, loss.data_ptr(),
                                             lse.data_ptr(), am.data_ptr(), st)
        else:
            ws = torch.empty(lib.abc_infonce_workspace_bytes(B, C, d), dtype=torch.uint8, device=dev)
            rc = lib.abc_infonce_fwd(q.data_ptr(), p.data_ptr(), t.data_ptr(), eps, B, C, d, loss.data_ptr(), lse.data_ptr(),
                                     am.data_ptr(), ws.data_ptr(), ws.numel(), st)
        assert rc == 0, lib.abc_last_error()
        torch.cuda.synchronize()
        out.append((loss.cpu(), lse.cpu(), am.cpu()))
    return out


def test_no_alloc_forward_matches_workspace_forward():
    from vllm_amd import _lib
    import oracle
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    t = torch.tensor([0.07], device=dev)
    unit = lambda n, d: torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)  # noqa: E731
    for B, C, d in ((128, 1024, 4096), (33, 100, 72)):           # the second call is smaller than the first
        q, p = unit(B, d), unit(C, d)
        p[:B] = torch.nn.functional.normalize(q + 0.8 * p[:B], dim=-1)
        (l0, s0, a0), (l1, s1, a1) = _both(lib, q.to(dev), p.to(dev), t, 0.1)
        assert torch.equal(l0, l1) and torch.equal(s0, s1) and torch.equal(a0, a1)      # same kernel, same layout: same bits
        ref, _ = oracle.compute_contrastive_loss(q, p, oracle.OracleTemperature(0.07), 0.1)
        assert abs(float(l1) - float(ref)) <= 1e-5 * abs(float(ref))
    # beyond the stated maximum: refused before anything is launched
    cap = lib.abc_infonce_noalloc_max_bytes()
    assert cap == 16 << 20
    B, C, d = 256, 16384, 64
    assert lib.abc_infonce_workspace_bytes(B, C, d) > cap
    x = torch.zeros((C, d), device=dev)
    o = torch.zeros((B,), device=dev)
    rc = lib.abc_infonce_fwd_noalloc(x.data_ptr(), x.data_ptr(), t.data_ptr(), 0.1, B, C, d, o.data_ptr(), o.data_ptr(),
                                     o.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    assert rc == -2 and b"workspace too small" in lib.abc_last_error()
