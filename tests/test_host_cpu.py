"""CPU-only checks: the C-ABI library loads and exports every declared symbol, argument errors are
reported without touching a GPU, the product path refuses to run without one, and the sharded
exchange logic (world_size 2, gloo) is correct when fed an exact checker in place of the kernels."""
import ctypes
import os
import re

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vllm_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "abcsearch.h")).read()
    declared = set(re.findall(r"\b(abc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.abc_version()


def test_argument_errors_without_gpu():
    from vllm_amd import _lib
    lib = _lib.load()
    assert lib.abc_topk_workspace_bytes(10, 10, 72, 5, _lib.DT_BF16) == 0          # d*2 % 128 != 0
    assert b"multiple of 128" in lib.abc_last_error()
    assert lib.abc_topk_workspace_bytes(10, 10, 64, 129, _lib.DT_BF16) == 0        # k too large
    assert lib.abc_topk_workspace_bytes(8192, 3_300_000, 4096, 64, _lib.DT_BF16) > 0
    assert lib.abc_topk_bf16(None, 4, None, 4, 64, 5, None, 0, None, None, None, _lib.NO_SELF, None, None, 0, None) == -1
    assert lib.abc_topk_merge(None, None, 0, 4, 5, 100, None, None, None) == -1
    assert lib.abc_infonce_workspace_bytes(8, 4, 64) == 0                           # C < B
    assert lib.abc_infonce_workspace_bytes(128, 1024, 4096) > 0


def test_product_path_refuses_cpu_tensors():
    from vllm_amd import kernels, _lib, loss
    q = torch.zeros((4, 64), dtype=torch.bfloat16)
    with pytest.raises(_lib.AbcError):
        kernels.topk(q, q, 2)
    with pytest.raises(_lib.AbcError):
        loss.compute_contrastive_loss(torch.zeros(4, 64), torch.zeros(4, 64), loss.Temperature())


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "vllm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_shard_bounds_cover():
    from vllm_amd.parallel import shard_bounds
    for n in (0, 1, 7, 3_300_000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


class _Emb:
    """CPU stand-in for search.Embeddings in the exchange test (rows(), shape, data)."""

    def __init__(self, x):
        self.data, self.scale = x, None

    @property
    def shape(self):
        return self.data.shape

    def rows(self, lo, hi):
        return _Emb(self.data[lo:hi])

    def take(self, index):
        return _Emb(self.data[index])


def _worker(rank, world, port, ret):
    import oracle
    from vllm_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, c = oracle.make_clustered(40, 301, 64, seed=6, n_centres=4)
    c[100] = c[200]                                                     # a tie across two shards
    c[8] = -q[5]                                                        # noisy pair (row 3+5): s_pos = -1, only the positive survives
    lo, hi = parallel.shard_bounds(c.shape[0], world, rank)

    def _topk(qe, ce, k, ub, off):
        v, i = oracle.topk_ref(qe.data, ce.data, k, upper_bound=ub)
        return v, torch.where(i >= 0, i + off, i)

    def _merge(vals, idx, idx_limit=None):
        parts, nq, k = vals.shape
        v = vals.permute(1, 0, 2).reshape(nq, parts * k)
        i = idx.permute(1, 0, 2).reshape(nq, parts * k)
        key_i = torch.where(i >= 0, i, torch.full_like(i, 1 << 40))
        o1 = torch.argsort(key_i, dim=1, stable=True)
        v, i = torch.gather(v, 1, o1), torch.gather(i, 1, o1)
        o2 = torch.argsort(v, dim=1, descending=True, stable=True)
        return torch.gather(v, 1, o2)[:, :k].contiguous(), torch.gather(i, 1, o2)[:, :k].contiguous()

    # the packed pair, restated with numpy: key = orderable(score) << 32 | (0xFFFFFFFF - index), 0 = no entry
    import numpy as np

    def _pack(v, i):
        u = v.numpy().view(np.uint32).astype(np.uint64)
        o = np.where(u >> 31, u ^ np.uint64(0xFFFFFFFF), u ^ np.uint64(0x80000000))
        key = (o << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - i.numpy().astype(np.uint64))
        return torch.from_numpy(np.where(i.numpy() >= 0, key, np.uint64(0)).view(np.int64))

    def _topk_packed(qe, ce, k, ub, off):
        v, i = _topk(qe, ce, k, ub, off)
        return _pack(v.float().contiguous(), i)

    def _merge_packed(keys):
        parts, nq, k = keys.shape
        kk = np.sort(keys.numpy().view(np.uint64).transpose(1, 0, 2).reshape(nq, parts * k), axis=1)[:, ::-1][:, :k]
        o = (kk >> np.uint64(32)).astype(np.uint32)
        u = np.where(o >> 31, o ^ np.uint32(0x80000000), o ^ np.uint32(0xFFFFFFFF))
        v = torch.from_numpy(np.where(kk != 0, u.view(np.float32), -np.inf).astype(np.float32))
        i = torch.from_numpy(np.where(kk != 0, (np.uint64(0xFFFFFFFF) - (kk & np.uint64(0xFFFFFFFF))).astype(np.int64), -1))
        return v, i

    packed = os.environ.get("ABC_TEST_PACKED") == "1"
    be = parallel.Backend(topk=_topk, rowdot=lambda a, b: (a.data * b.data).sum(-1), merge=_merge,
                          topk_packed=_topk_packed if packed else None, merge_packed=_merge_packed if packed else None)
    sc = parallel.ShardedCorpus(_Emb(c[lo:hi]), lo, c.shape[0], backend=be)
    idx, vals, s_pos = sc.mine_negatives(_Emb(q), 16, ratio=0.95, offset=3, fill_masked_tail=False)
    # chunked + overlapped exchange gives the same lists as one exchange
    v2, i2 = sc.topk(_Emb(q), 16, upper_bound=s_pos * 0.95, chunk=16)
    assert torch.equal(i2, idx) and torch.equal(v2, vals)
    filled, _, _ = sc.mine_negatives(_Emb(q), 16, ratio=0.95, offset=3)     # the driver's form: tail filled
    ret[rank] = (idx, vals, s_pos, filled)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,packed", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_mining_exchange_gloo(world, packed, monkeypatch):
    """The exchange logic with an exact checker bound in place of the kernels: the two-tensor form and the packed form
    (ONE all-gather of 8-byte keys per chunk, merges pipelined behind the next chunk's sweep)."""
    import oracle
    monkeypatch.setenv("ABC_TEST_PACKED", "1" if packed else "0")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29650 + world + (10 if packed else 0), ret), nprocs=world, join=True)
    q, c = oracle.make_clustered(40, 301, 64, seed=6, n_centres=4)
    c[100] = c[200]
    c[8] = -q[5]
    s_pos = (q * c[3:43]).sum(-1)
    rv, ri = oracle.topk_ref(q, c, 16, upper_bound=s_pos * 0.95)
    assert int((ri[5] < 0).sum()) > 0                                            # the noisy pair leaves holes
    # N = 1 reference of the filled form: the unsharded helper on the whole corpus (N = 1 and N > 1 must agree)
    from vllm_amd import parallel
    from vllm_amd.search import fill_masked_tail_rows
    want = ri.clone()
    fill_masked_tail_rows(want, s_pos * 0.95, lambda rows: oracle.topk_ref(q[rows], c, 16))
    assert bool((want >= 0).all()) and all(len(set(r.tolist())) == 16 for r in want)
    sc1 = parallel.ShardedCorpus(_Emb(c), 0, c.shape[0], backend=parallel.Backend(
        topk=lambda qe, ce, k, ub, off: oracle.topk_ref(qe.data, ce.data, k, upper_bound=ub), rowdot=lambda a, b: (a.data * b.data).sum(-1),
        merge=None))
    one, _, _ = sc1.mine_negatives(_Emb(q), 16, ratio=0.95, offset=3)             # world 1 through the same class
    assert torch.equal(one, want)
    for r in range(world):
        idx, vals, sp, filled = ret[r]
        assert torch.allclose(sp, s_pos, atol=1e-6)
        assert torch.equal(idx, ri) and torch.allclose(vals, rv, atol=1e-6)      # replicated and exact on every rank
        assert torch.equal(filled, want)                                         # same lists at every world size


def _gather_grad_worker(rank, world, port, ret):
    from vllm_amd import loss as L
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    c = torch.randn(3, 5, generator=g, requires_grad=True)
    w = torch.randn(world * 3, 5, generator=torch.Generator().manual_seed(500 + rank))   # this rank's "loss" weights
    c_all = L._AllGatherWithGrad.apply(c)
    (c_all * w).sum().backward()
    ret[rank] = (c_all.detach(), c.grad.clone(), w)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_differentiable_gather_gloo(world):
    """Forward: the reference's slot order (own block first, rank 0's block in slot `rank`); backward: the
    block of rank r receives the sum over ranks s of the weights rank s applied to the slot holding r."""
    from oracle.loss_oracle import gathered_column_order
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_grad_worker, args=(world, 29750 + world, ret), nprocs=world, join=True)
    blocks = [torch.randn(3, 5, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    for r in range(world):
        order = gathered_column_order(r, world)
        c_all, grad, _ = ret[r]
        assert torch.equal(c_all, torch.cat([blocks[o] for o in order]))
        want = torch.zeros(3, 5)
        for s_rank in range(world):
            slot = gathered_column_order(s_rank, world).index(r)
            want += ret[s_rank][2][slot * 3:(slot + 1) * 3]
        assert torch.allclose(grad, want, atol=1e-6)


def test_merge_rejects_indices_beyond_32_bits():
    """The merge packs indices into a 32-bit field: it refuses, before any launch, instead of wrapping."""
    from vllm_amd import _lib, parallel
    lib = _lib.load()
    dummy = ctypes.c_void_p(16)                     # never dereferenced: the argument check comes first
    assert lib.abc_topk_merge(dummy, dummy, 2, 4, 5, (1 << 32), dummy, dummy, None) == -1
    assert b"idx_limit" in lib.abc_last_error()
    assert lib.abc_topk_merge(dummy, dummy, 2, 4, 5, -1, dummy, dummy, None) == -1
    with pytest.raises(ValueError):
        parallel.ShardedCorpus(_Emb(torch.zeros(4, 8)), 0, 1 << 32, backend=parallel.Backend(None, None, None))
    parallel.ShardedCorpus(_Emb(torch.zeros(4, 8)), 0, (1 << 32) - 2, backend=parallel.Backend(None, None, None))


def test_debug_knobs_are_process_state_not_environment(monkeypatch):
    """ABC_* variables are read once, when the library is loaded: changing the environment afterwards does nothing;
    abc_debug_set is the only way to change a knob, and result-altering flag bits are masked in a production build."""
    from vllm_amd import kernels
    before = kernels.debug_get("plan_qb")
    monkeypatch.setenv("ABC_PLAN_QB", "16" if before != 16 else "24")
    monkeypatch.setenv("ABC_DEBUG_FLAGS", "3")
    assert kernels.debug_get("plan_qb") == before and kernels.debug_get("flags") & 3 == 0
    try:
        kernels.debug_set("plan_qb", 16)
        assert kernels.debug_get("plan_qb") == 16
        kernels.debug_set("flags", 1 | 2 | 8)
        assert kernels.debug_get("flags") == 8             # 1 and 2 (wrong results) do not exist outside -DABC_DIAG builds
    finally:
        kernels.debug_set("plan_qb", before)
        kernels.debug_set("flags", 0)
    from vllm_amd import _lib
    with pytest.raises(_lib.AbcError):
        kernels.debug_set("no_such_knob", 1)


def test_no_per_process_launch_state_in_the_library():
    """Boundary hygiene: kernel attributes are set per launch for the current device, not cached in a
    process-wide flag; the environment is read in one place (the tuning block's constructor)."""
    csrc = os.path.join(ROOT, "vllm_amd", "csrc")
    text = ""
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".inc", ".h")):
            text += open(os.path.join(csrc, fn)).read()
    assert not re.search(r"static\s+bool\s+\w*attr\w*", text)
    assert len(re.findall(r"getenv\(", text)) == 2 and text.count("SweepTuning()") == 1
