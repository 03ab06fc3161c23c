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
        acc = torch.full((), float("nan"), device=dev)
        if noalloc:
            rc = lib.abc_infonce_fwd_noalloc(q.data_ptr(), p.data_ptr(), t.data_ptr(), eps, B, C, d, loss.data_ptr(),
                                             lse.data_ptr(), am.data_ptr(), acc.data_ptr(), st)
        else:
            ws = torch.empty(lib.abc_infonce_workspace_bytes(B, C, d), dtype=torch.uint8, device=dev)
            rc = lib.abc_infonce_fwd(q.data_ptr(), p.data_ptr(), t.data_ptr(), eps, B, C, d, loss.data_ptr(), lse.data_ptr(),
                                     am.data_ptr(), acc.data_ptr(), ws.data_ptr(), ws.numel(), st)
        assert rc == 0, lib.abc_last_error()
        torch.cuda.synchronize()
        assert float(acc) == float((am.cpu() == torch.arange(B, dtype=torch.int32)).sum() / B)     # abc_util.py:52-54
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
                                     o.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream)
    assert rc == -2 and b"workspace too small" in lib.abc_last_error()


def test_no_alloc_forward_refuses_overlap_and_replays_from_a_graph():
    """(1) a call on a second stream while the first stream still has work pending is refused (the calls
    share one array per GPU) instead of racing; (2) after the warm-up call the entry point runs under stream capture,
    the graph replays with the same bits as the eager call, and neither capture nor replay allocates device memory."""
    from vllm_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    B, C, d = 64, 256, 512
    unit = lambda n: torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(dev)  # noqa: E731
    q, p = unit(B), unit(C)
    t = torch.tensor([0.07], device=dev)
    loss, acc = torch.zeros((), device=dev), torch.zeros((), device=dev)
    lse, am = torch.zeros(B, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)

    def call(stream):
        return lib.abc_infonce_fwd_noalloc(q.data_ptr(), p.data_ptr(), t.data_ptr(), 0.1, B, C, d, loss.data_ptr(),
                                           lse.data_ptr(), am.data_ptr(), acc.data_ptr(), stream.cuda_stream)

    assert lib.abc_infonce_noalloc_max_bytes() == 16 << 20                      # the warm-up call
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    big = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()                                                    # the earlier calls' stream is idle
    with torch.cuda.stream(s1):
        for _ in range(20):
            big = big @ big * 1e-4                                              # keeps s1 busy for a while
        assert call(s1) == 0, lib.abc_last_error()
    rc = call(s2)                                                               # s1 still running: refused, nothing launched
    assert rc == -1 and b"another stream" in lib.abc_last_error()
    torch.cuda.synchronize()
    assert call(s2) == 0, lib.abc_last_error()                                  # once s1 has drained it is fine
    torch.cuda.synchronize()
    eager = (loss.clone(), lse.clone(), am.clone(), acc.clone())

    # under capture: no allocation, same bits on replay
    free0 = torch.cuda.mem_get_info(dev)[0]
    graph = torch.cuda.CUDAGraph()
    s3 = torch.cuda.Stream(dev)
    with torch.cuda.graph(graph, stream=s3):
        rc = lib.abc_infonce_fwd_noalloc(q.data_ptr(), p.data_ptr(), t.data_ptr(), 0.1, B, C, d, loss.data_ptr(),
                                         lse.data_ptr(), am.data_ptr(), acc.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, lib.abc_last_error()
    for x in (loss, lse, acc):
        x.fill_(float("nan"))
    am.fill_(-1)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss, eager[0]) and torch.equal(lse, eager[1]) and torch.equal(am, eager[2]) and torch.equal(acc, eager[3])
    assert torch.cuda.mem_get_info(dev)[0] >= free0 - (8 << 20)                 # graph bookkeeping only, no 16 MiB scratch
