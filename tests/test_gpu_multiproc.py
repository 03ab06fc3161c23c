"""The product's cross-rank loss under a LIVE process group on the GPU:

    loss.compute_gathered_loss                                model/abc_util.py:5-31
    loss.compute_gathered_loss(differentiable_gather=True)    UniIR clip_sf.py:103 (torch.distributed.nn.all_gather)

2 and 4 fresh child processes, every rank on cuda:0 (the box has one GPU), gloo group: RCCL refuses two ranks on
one device, so the collectives cross in host memory (loss._all_gather_rows) while the gather / slot-swap code, the
local-gradient rule and the kernels are the production ones.  Per-rank loss, accuracy, num_cand and the gradients
of q, the local c block and temp are compared with tests/golden/gathered_gloo.npz, which the reference's own
compute_gathered_loss produced under a gloo group (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import free_port

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gathered_gloo.npz")


def _rank_main(rank, world, port, differentiable, ret):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)          # a hung collective ends the child, with its stack
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vllm_amd import loss as L
    z = np.load(GOLDEN)
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(z[f"w{world}_r{rank}_q"]).to(dev).requires_grad_(True)
    c = torch.from_numpy(z[f"w{world}_r{rank}_c"]).to(dev).requires_grad_(True)
    temp = L.Temperature(float(z[f"w{world}_temp"])).to(dev)
    l, a, n = L.compute_gathered_loss(q, c, temp, label_smoothing=float(z[f"w{world}_eps"]),
                                      differentiable_gather=differentiable)
    l.backward()
    torch.cuda.synchronize()
    ret[rank] = dict(loss=l.detach().cpu(), acc=a.cpu(), num=n.cpu(), dq=q.grad.cpu(), dc=c.grad.cpu(),
                     dtemp=temp.temp.grad.cpu())
    dist.barrier()
    dist.destroy_process_group()


def _run(world, differentiable, port):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(world, port, differentiable, ret), nprocs=world, join=True)
    return {r: ret[r] for r in range(world)}


@pytest.mark.parametrize("world", [2, 4])
def test_compute_gathered_loss_under_process_group(world):
    z = np.load(GOLDEN)
    got = _run(world, False, free_port())
    for r in range(world):
        g = lambda k: torch.from_numpy(np.asarray(z[f"w{world}_r{r}_{k}"]))  # noqa: E731
        o = got[r]
        assert abs(float(o["loss"]) - float(g("loss"))) <= 1e-5 * abs(float(g("loss"))), (r, float(o["loss"]), float(g("loss")))
        assert float(o["acc"]) == float(g("acc")) and int(o["num"]) == int(g("num")[0])
        assert torch.allclose(o["dq"], g("dq"), rtol=1e-3, atol=2e-6)
        assert torch.allclose(o["dc"], g("dc"), rtol=1e-3, atol=2e-6)        # local block only: remote blocks are constants
        assert abs(float(o["dtemp"]) - float(g("dtemp"))) <= 1e-3 * abs(float(g("dtemp"))) + 1e-5


@pytest.mark.parametrize("world", [2, 4])
def test_differentiable_gather_under_process_group(world):
    """Opt-in variant: same forward values as the reference's gather, but every rank's loss sends a gradient to
    every candidate block (backward = reduce-scatter(SUM)).  Expected gradients: autograd over the sum of the
    per-rank losses with all blocks as leaves (torch-CPU, the oracle's compute_contrastive_loss)."""
    z = np.load(GOLDEN)
    eps, t = float(z[f"w{world}_eps"]), float(z[f"w{world}_temp"])
    qs = [torch.from_numpy(z[f"w{world}_r{r}_q"]).clone().requires_grad_(True) for r in range(world)]
    cs = [torch.from_numpy(z[f"w{world}_r{r}_c"]).clone().requires_grad_(True) for r in range(world)]
    temps = [oracle.OracleTemperature(t) for _ in range(world)]
    losses = []
    for r in range(world):
        order = oracle.gathered_column_order(r, world)
        l, _ = oracle.compute_contrastive_loss(qs[r], torch.cat([cs[o] for o in order], 0), temps[r], eps)
        losses.append(l)
    torch.stack(losses).sum().backward()
    got = _run(world, True, free_port())
    for r in range(world):
        o = got[r]
        assert abs(float(o["loss"]) - float(losses[r])) <= 1e-5 * abs(float(losses[r]))
        assert abs(float(o["loss"]) - float(z[f"w{world}_r{r}_loss"])) <= 1e-5 * abs(float(z[f"w{world}_r{r}_loss"]))
        assert int(o["num"]) == world * cs[r].shape[0]
        assert torch.allclose(o["dq"], qs[r].grad, rtol=1e-3, atol=2e-6)
        assert torch.allclose(o["dc"], cs[r].grad, rtol=1e-3, atol=2e-6)     # gradients from EVERY rank's loss
        assert abs(float(o["dtemp"]) - float(temps[r].temp.grad)) <= 1e-3 * abs(float(temps[r].temp.grad)) + 1e-5


def _mine_rank(rank, world, port, ret):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vllm_amd import parallel, search
    q, c = oracle.make_clustered(700, 6000, 256, seed=91, n_centres=12)
    c[4100] = c[33]                                            # a tie across shards
    lo, hi = parallel.shard_bounds(c.shape[0], world, rank)
    qe = search.Embeddings(q, "bf16", "cuda:0")
    sc = parallel.ShardedCorpus(search.Embeddings(c[lo:hi], "bf16", "cuda:0"), lo, c.shape[0])
    idx, vals, s_pos = sc.mine_negatives(qe, 48, chunk=256)    # 3 chunks through the exchange, ragged last chunk
    torch.cuda.synchronize()
    ret[rank] = (idx.cpu(), vals.cpu(), s_pos.cpu())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_mining_under_process_group(world):
    """parallel.ShardedCorpus with the production kernels: every rank sweeps its corpus shard, the lists cross as
    packed 8-byte keys in ONE all-gather per query chunk, merges run behind the next chunk's sweep; every rank ends
    with the single-GPU result (search.mine_negatives over the whole corpus), bit for bit."""
    from vllm_amd import search
    q, c = oracle.make_clustered(700, 6000, 256, seed=91, n_centres=12)
    c[4100] = c[33]
    qe, ce = search.Embeddings(q, "bf16", "cuda:0"), search.Embeddings(c, "bf16", "cuda:0")
    want = [t.cpu() for t in search.mine_negatives(qe, ce, 48)]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mine_rank, args=(world, free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        idx, vals, s_pos = ret[r]
        assert torch.equal(idx, want[0]), r
        assert torch.equal(vals, want[1]), r
        assert torch.equal(s_pos, want[2]), r


def _driver_rank(rank, world, port, path):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vllm_amd import mining
    mining.create_negative_index(path, top=100, compute="f32", block=96, resume=False)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_mining_driver_under_process_group(world, tmp_path, golden_dir):
    """mining.create_negative_index (the pipelined driver) with the corpus row-sharded over 2 / 3 ranks: same negatives.json,
    byte for byte, as the one-process run -- on the reference-run fixture inputs, whose short rows make every rank take the
    tail-fill path (a second, collective search) inside the write-out of a block -- and the reference's own lists."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from glue_util import Glue
    from vllm_amd import mining
    g = Glue(golden_dir)
    q, c = g.mining_inputs()
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    mining.save_embeddings(one, q, c)
    mining.save_embeddings(many, q, c)
    want = open(mining.create_negative_index(one, top=100, compute="f32", block=96)).read()
    mp.spawn(_driver_rank, args=(world, free_port(), many), nprocs=world, join=True)
    got = open(os.path.join(many, "negatives.json")).read()
    assert got == want
    assert not os.path.exists(os.path.join(many, mining.PARTS_DIR))
    g.check_negatives({int(k): v for k, v in json.loads(got).items()})
