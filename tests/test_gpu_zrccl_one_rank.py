"""Every RCCL call of the product, executed under a ONE-rank "nccl" group (the boxes have one GPU and RCCL takes one rank per
device; the multi-rank tests in test_gpu_multiproc.py run under gloo with host staging).  Collected late on purpose: a box
whose RCCL cannot come up must not hide the tests after it."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import free_port

pytestmark = pytest.mark.gpu


def _rccl_one_rank_main(rank, port, ret):
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # as bench.py does for N > 1
        dist.barrier()
    except Exception as e:                                                         # the environment, not the product
        ret[0] = {"no_rccl": repr(e)}
        return
    from vllm_amd import loss as L, parallel, search
    out = {}
    # -- the sharded search's exchange: packed sweep -> all_gather_into_tensor (int64 keys) -> device merge;
    # all_reduce(SUM) of the positive scores; three chunks through the pipelined loop
    q, c = oracle.make_clustered(700, 6000, 128, seed=12, n_centres=8)
    q, c = oracle.bf16_round(q), oracle.bf16_round(c)
    ce = search.Embeddings(c, "bf16", dev)
    qe = search.Embeddings(q, "bf16", dev)
    ref_i, ref_v, ref_p = search.mine_negatives(qe, ce, 64)
    sc = parallel.ShardedCorpus(ce, 0, c.shape[0], exchange=True)
    assert sc.exchange and dist.get_backend() == "nccl" and not sc._host_staged(ce.data)
    got_i, got_v, got_p = sc.mine_negatives(qe, 64, chunk=256)
    out["search"] = bool(torch.equal(got_i, ref_i) and torch.equal(got_v, ref_v) and torch.equal(got_p, ref_p))
    v2, i2 = sc.topk(qe, 10, chunk=300)
    rv, ri, _ = search.topk(qe, ce, 10)
    out["topk"] = bool(torch.equal(v2, rv) and torch.equal(i2, ri))
    # -- the loss's gather (RCCL takes the device tensor as it is) and the opt-in gather with a gradient (reduce-scatter)
    x = torch.randn(96, 256, device=dev)
    out["gather"] = bool(torch.equal(L._all_gather_rows(x), x))
    y = x.clone().requires_grad_(True)
    g = L._AllGatherWithGrad.apply(y)
    (g * 3.0).sum().backward()
    out["gather_grad"] = bool(torch.equal(g, x) and torch.equal(y.grad, torch.full_like(x, 3.0)))
    # -- bench.py's own collectives for N > 1: barrier, MAX of a float64 scalar on the device
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["bench_calls"] = float(t.item()) == 1.25
    torch.cuda.synchronize()
    ret[0] = out
    dist.destroy_process_group()


def test_rccl_calls_with_one_rank():
    """The box has one GPU and RCCL takes one rank per device: a ONE-rank "nccl" group is the only way to execute the
    RCCL calls of the product here (every multi-rank test above runs under gloo with host staging).  ShardedCorpus with
    exchange=True goes through the full exchange -- packed sweep, all_gather_into_tensor, device merge,
    all_reduce of s_pos -- and must reproduce the unsharded search bit for bit; the loss's gather, its reduce-scatter
    backward and the collectives bench.py issues for N > 1 run too.  What one rank cannot show is xGMI traffic and
    cross-rank ordering: the first multi-rank RCCL run is the driver's."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_one_rank_main, args=(free_port(), ret), nprocs=1, join=True)
    if "no_rccl" in ret[0]:
        pytest.skip("RCCL did not come up on this box: " + ret[0]["no_rccl"])
    assert ret[0] == {"search": True, "topk": True, "gather": True, "gather_grad": True, "bench_calls": True}, dict(ret[0])
