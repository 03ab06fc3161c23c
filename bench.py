#!/usr/bin/env python3
"""Headline benchmark: exact top-64 hard-negative mining over a 3.3M x 4096 bf16 corpus.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic queries, inputs resident in HBM:
positive scores (rowdot) -> fused score-GEMM + top-k sweep with the 0.95-positive upper bound ->
list merge (and, for N > 1, one RCCL all-reduce of s_pos and one all-gather of the per-shard lists).
The corpus (BASELINE.json configs[2]) is row-sharded over the N GPUs; every step processes N x 8192
queries, so the work per GPU is fixed as N grows ("weak").  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

# multi-process GPU work on this image needs dmabuf IPC (the host driver has no legacy IPC: RCCL fails with
# "hipIpcGetMemHandle: invalid argument" without it); already exported on the boxes, kept here for any other launcher
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                                                              # noqa: E402
import torch.distributed as dist                                          # noqa: E402

ROOT =os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NC_FULL = 3_300_000
D = 4096
K = 64
QUERIES_PER_GPU_STEP = 8192
CHUNK = 65536
N_CENTRES = 4096
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def centres(dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn((N_CENTRES, D), generator=g, device=dev)


def gen_chunk(chunk_id: int, mu: torch.Tensor, seed: int):
    """Rows [chunk_id*CHUNK, +CHUNK) of the synthetic pair table: c_j =
    normalize(mu[j mod 4096] + 0.7 eps_j), q_j = normalize(c_j + 0.5 eps'_j / sqrt(d)); bf16 once."""
    dev = mu.device
    g = torch.Generator(device=dev).manual_seed(seed * 1_000_003 + chunk_id)
    j = torch.arange(chunk_id * CHUNK, (chunk_id + 1) * CHUNK, device=dev)
    c = mu[j % N_CENTRES] + 0.7 * torch.randn((CHUNK, D), generator=g, device=dev)
    c = torch.nn.functional.normalize(c, dim=-1)
    q = torch.nn.functional.normalize(c + (0.5 / math.sqrt(D)) * torch.randn((CHUNK, D), generator=g, device=dev), dim=-1)
    return c.to(torch.bfloat16), q.to(torch.bfloat16)


def gen_rows(lo: int, hi: int, mu, seed: int, want_q: bool):
    out = torch.empty((hi - lo, D), dtype=torch.bfloat16, device=mu.device)
    for cid in range(lo // CHUNK, (hi - 1) // CHUNK + 1):
        c, q = gen_chunk(cid, mu, seed)
        a, b = max(lo, cid * CHUNK), min(hi, (cid + 1) * CHUNK)
        out[a - lo:b - lo] = (q if want_q else c)[a - cid * CHUNK:b - cid * CHUNK]
        del c, q
    return out


_RESULT_OUT = None


def _only_json_on_stdout():
    """RCCL prints a version banner to STDOUT from C when the first communicator comes up (seen on the box: five lines
    before the result).  The contract is ONE JSON line on stdout: everything that writes to file descriptor 1 from here on
    goes to stderr, and the result line alone goes to the real stdout."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def _emit(result):
    print(json.dumps(result), file=_RESULT_OUT, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nc", type=int, default=NC_FULL, help="corpus rows (default: the 3.3M of BASELINE.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-job", action="store_true",
                    help="time the WHOLE mining job (every corpus row is a query: 3.3M queries, about 70 s on one GPU) "
                         "instead of --steps batches of it; --steps is then derived from the corpus size")
    ap.add_argument("--from-files", action="store_true",
                    help="time the file-driven mining job (create_negative_index.py:30-43): synthetic query.pt / cand.pt "
                         "(fp32, --nc rows, default 412 500) written to local disk -> negatives.json through "
                         "vllm_amd.mining.create_negative_index, beside the rate of the same job with the queries resident in HBM")
    args = ap.parse_args()
    _only_json_on_stdout()
    if args.from_files:
        return from_files_mode(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # rehearsal of the N > 1 path on a one-GPU box: ABC_BENCH_ONE_GPU=1 puts every rank on cuda:0 and
    # ABC_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device)
    if os.environ.get("ABC_BENCH_WATCHDOG"):              # seconds: dump every thread's stack and exit (hang diagnosis)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ABC_BENCH_WATCHDOG"]), exit=True)
    one_gpu = os.environ.get("ABC_BENCH_ONE_GPU") == "1"
    backend = os.environ.get("ABC_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # rank's-eye rehearsal of the N = 8 step on ONE GPU (ABC_BENCH_RANK_OF=8): this process plays one of 8 ranks -- a
    # 1/8 shard, 8 x 8192 queries per step, the whole exchange (packed sweep, all-gather, merge, all-reduce of s_pos)
    # through RCCL under a one-rank group -- so everything but the wire time of the collectives is measured
    rank_of = int(os.environ.get("ABC_BENCH_RANK_OF", "0"))
    if rank_of > 1 and world != 1:
        raise SystemExit("ABC_BENCH_RANK_OF is a one-process rehearsal")
    if rank_of > 1:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from vllm_amd import search, parallel, kernels

    seed = 1234
    nc = args.nc
    mu = centres(dev, seed)
    lo, hi = parallel.shard_bounds(nc, max(world, rank_of, 1), rank)
    shard = search.Embeddings(gen_rows(lo, hi, mu, seed, want_q=False), "bf16", dev)
    corpus = parallel.ShardedCorpus(shard, lo, hi if rank_of > 1 else nc, backend=None, exchange=True if rank_of > 1 else None)

    # time the sweep (the dominant kernel) on its own with events on the launch stream
    sweep_ms = []

    def timed(inner):
        def run(q, c, k, ub, off):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner(q, c, k, ub, off)
            e1.record()
            sweep_ms.append((e0, e1))
            return out
        return run
    corpus.backend.topk = timed(corpus.backend.topk)                    # N = 1: sweep + merge to (values, indices)
    if corpus.backend.topk_packed is not None:
        corpus.backend.topk_packed = timed(corpus.backend.topk_packed)  # N > 1: sweep + merge to packed keys

    bq = QUERIES_PER_GPU_STEP * max(world, rank_of, 1)
    batches = []
    if args.full_job:
        # every row of the pair table is a query (create_negative_index.py:37): all of them resident in HBM (27 GB
        # at the full size) before the clock starts, one step per bq of them, the last one ragged
        args.steps = (nc + bq - 1) // bq
        all_q = search.Embeddings(gen_rows(0, nc, mu, seed, want_q=True), "bf16", dev)
        for s in range(args.warmup):
            batches.append((0, all_q.rows(0, min(bq, nc))))
        for s in range(args.steps):
            batches.append((s * bq, all_q.rows(s * bq, min((s + 1) * bq, nc))))
    else:
        for s in range(args.steps + args.warmup):
            r0 = (s * bq) % max((hi - lo if rank_of > 1 else nc) - bq, 1)      # (rehearsal: positives inside the one shard)
            batches.append((r0, search.Embeddings(gen_rows(r0, r0 + bq, mu, seed, want_q=True), "bf16", dev)))
    n_batches = len(batches)
    n_queries = sum(b[1].shape[0] for b in batches[args.warmup:])
    del mu

    def step(s):
        r0, q = batches[s]
        return corpus.mine_negatives(q, K, ratio=0.95, offset=r0)

    for s in range(args.warmup):
        step(s)
    sweep_ms.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, n_batches):
        out = step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per step: every sweep launch of the step (N > 1 sweeps the step's queries in chunks of 8192)
    sweep_avg_ms = sum(a.elapsed_time(b) for a, b in sweep_ms) / max(args.steps, 1)
    flops_per_step = 2.0 * (n_queries / max(args.steps, 1)) * (hi - lo) * D   # algorithmic: 2*Nc_local*D per query
    if sweep_avg_ms <= 0:
        raise SystemExit("bench.py: no sweep launch was timed (the timing hook missed the backend's entry point)")
    achieved = flops_per_step / (sweep_avg_ms * 1e-3) / 1e12
    launches_per_step = max(len(sweep_ms) // max(args.steps, 1), 1)
    flops_per_launch = flops_per_step / launches_per_step
    # HBM/fabric bytes per launch: PMC counters need rocprofv3 passes of their own (scripts/profile_round.sh), so
    # this run can only CARRY OVER the figure of the last profiled run of the same kernel and shape, and says so
    traffic, traffic_source = None, "not measured in this run (no PMC pass)"
    tfile = os.path.join(ROOT, "bench_traffic.json")
    if os.path.exists(tfile) and world == 1 and nc == NC_FULL:
        try:
            tj = json.load(open(tfile))
            traffic = tj.get("sweep_hbm_bytes_per_launch")
            traffic_source = ("carried over from " + str(tj.get("source", "bench_traffic.json")) +
                              ": FETCH_SIZE x 2 of a separate rocprofv3 --pmc pass, NOT measured in this run")
        except Exception:
            traffic = None

    result = {
        "metric": "queries/sec exact top-64 over 3.3M x 4096 bf16 corpus",
        "value": n_queries / dt,
        "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"CC3M hard-negative mining (BASELINE configs[2]): corpus {nc} x {D} bf16 row-sharded over "
                               f"{world} GPU(s), {bq} queries/step, k={K}, 0.95-positive upper bound",
                   "queries_per_step": bq, "corpus_rows": nc, "d": D, "k": K, "parallelism": f"corpus-shard{world}",
                   "queries_timed": n_queries,
                   "job": ("the full job: every one of the corpus rows as a query" if args.full_job else
                           f"{n_queries} queries of the {nc}-query job timed; the job's rate follows from the per-step rate "
                           "(every step sweeps the whole corpus; --full-job times all of it)")},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": sweep_kernel_name(hi - lo), "avg_launch_ms": sweep_avg_ms / launches_per_step,
                     "sweep_ms_per_step": sweep_avg_ms,
                     "flops_per_launch": flops_per_launch},
    }

    if world > 1 and os.environ.get("ABC_BENCH_CHECK") == "1":
        # rehearsal only: the sharded result of the last step against an unsharded search of the same queries
        r0, qlast = batches[-1]
        mu2 = centres(dev, seed)
        full = search.Embeddings(gen_rows(0, nc, mu2, seed, want_q=False), "bf16", dev)
        ref_i, ref_v, _ = search.mine_negatives(qlast, full, K, ratio=0.95, offset=r0)
        assert torch.equal(ref_i, out[0]) and torch.equal(ref_v, out[1]), "sharded result differs from the unsharded search"
        result["rehearsal_check"] = "sharded == unsharded (bit-exact)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(batches[-1], shard, nc, out, dev)

    if rank_of > 1:
        result["rehearsal"] = (f"one process playing rank 0 of {rank_of}: shard of {hi - lo} rows, {bq} queries per step, the exchange "
                               "through RCCL under a one-rank group (no wire time)")
    if rank == 0:
        _emit(result)
    if world > 1 or rank_of > 1:
        dist.destroy_process_group()


def from_files_mode(args):
    """One GPU.  The whole job of pretrain/create_negative_index.py:30-43 from the files the embedding pass leaves
    (query.pt / cand.pt, fp32 [N, D]) to negatives.json, timed end to end and by phase, beside the same N queries mined
    from HBM-resident bf16 embeddings (what the default bench line times per step).  PCIe and file reads are inside
    the file-driven number; the files were written just before, so they come from the page cache, not the disk."""
    import shutil
    import tempfile
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("--from-files is a one-GPU measurement")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from vllm_amd import search, mining
    nc = args.nc if args.nc != NC_FULL else 412_500
    seed = 1234
    mu = centres(dev, seed)
    root = tempfile.mkdtemp(prefix="abc_mining_", dir=os.environ.get("ABC_BENCH_TMP", tempfile.gettempdir()))
    try:
        neg = os.path.join(root, "negative")
        os.makedirs(neg)
        t0 = time.perf_counter()
        for name, want_q in (("cand.pt", False), ("query.pt", True)):
            host = torch.empty((nc, D), dtype=torch.float32)
            for lo in range(0, nc, CHUNK):
                hi = min(nc, lo + CHUNK)
                host[lo:hi] = gen_rows(lo, hi, mu, seed, want_q).float().cpu()
            torch.save(host, os.path.join(neg, name))
            del host
        t_write = time.perf_counter() - t0

        # (a) queries resident in HBM: the per-step path of the default bench line over all nc queries
        c_res = search.Embeddings(gen_rows(0, nc, mu, seed, want_q=False), "bf16", dev)
        q_res = search.Embeddings(gen_rows(0, nc, mu, seed, want_q=True), "bf16", dev)
        bq = QUERIES_PER_GPU_STEP
        search.mine_negatives(q_res.rows(0, bq), c_res, K, offset=0)              # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep = []
        for s in range(0, nc, bq):
            keep.append(search.mine_negatives(q_res.rows(s, min(nc, s + bq)), c_res, K, offset=s)[0])
        torch.cuda.synchronize()
        t_res = time.perf_counter() - t0
        ref_idx = torch.cat(keep).cpu()
        del keep, c_res, q_res
        torch.cuda.empty_cache()

        # (b) from the files
        timings = {}
        t0 = time.perf_counter()
        out = mining.create_negative_index(neg, top=K, compute="bf16", block=bq, resume=False, timings=timings)
        t_files = time.perf_counter() - t0
        size = os.path.getsize(out)
        # same lists as the resident run (spot check: first, last and a stride of rows)
        with open(out, "rb") as f:
            got = json.loads(f.read())
        rows = sorted(set(list(range(0, nc, max(nc // 997, 1))) + [nc - 1]))
        same = all(got[str(r)] == [j for j in ref_idx[r].tolist() if j >= 0] for r in rows)
        if len(got) != nc or not same:
            raise SystemExit("bench.py --from-files: negatives.json differs from the resident-query run")
        loop_rate = nc / timings["loop"]
        result = {
            "metric": "queries/sec exact top-64 hard-negative mining, query.pt/cand.pt -> negatives.json",
            "value": nc / t_files, "unit": "queries/s", "n_gpus": 1, "steps": (nc + bq - 1) // bq, "warmup": 1,
            "ms_per_step": t_files / ((nc + bq - 1) // bq) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"create_negative_index from files: {nc} x {D} fp32 query.pt + cand.pt -> bf16 search, k={K}, "
                                   f"0.95-positive rule, blocks of {bq} -> negatives.json ({size / 1e6:.0f} MB)",
                       "corpus_rows": nc, "d": D, "k": K},
            "from_files": {"total_s": t_files, "phases_s": timings, "loop_queries_per_s": loop_rate,
                           "resident_queries_per_s": nc / t_res, "loop_vs_resident": loop_rate / (nc / t_res),
                           "total_vs_resident": (nc / t_files) / (nc / t_res),
                           "files_written_s": t_write, "page_cache": "files written just before the run: reads come from the page cache",
                           "check": f"negatives.json == resident-query lists on {len(rows)} sampled rows, {len(got)} rows in all"},
        }
        _emit(result)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def sweep_kernel_name(rows):
    return "topk_sweep16_kernel<bf16>"


def cpu_baseline(batch, shard, nc, gpu_out, dev):
    """The oracle (torch-CPU fp32 matmul + topk, the reference's semantics) timed on this host on a
    bounded sample, and a parity gate of the HIP path against it on that sample."""
    import oracle
    from vllm_amd import search
    r0, q = batch
    n_rows = min(524288, shard.shape[0])
    nq_s = 2048
    qs_dev = q.rows(0, nq_s)
    cs_dev = shard.rows(0, n_rows)
    q_cpu = qs_dev.data.float().cpu()
    c_cpu = cs_dev.data.float().cpu()
    threads = torch.get_num_threads()
    # s_pos against the full-corpus positives does not apply to a sub-corpus: use a plain top-k sample
    t0 = time.perf_counter()
    v_ref, i_ref = oracle.topk_ref(q_cpu, c_cpu, K, q_chunk=128)
    t_cpu = time.perf_counter() - t0
    v_gpu, i_gpu, _ = search.topk(qs_dev, cs_dev, K)
    torch.cuda.synchronize()
    npar = 64
    oracle.check_topk_parity(q_cpu[:npar], c_cpu, v_gpu[:npar].cpu(), i_gpu[:npar].cpu(), rtol=1e-3)
    same = float((i_gpu.cpu() == i_ref).float().mean())
    return {"value": nq_s / t_cpu * (n_rows / nc), "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"{nq_s} queries x {n_rows} corpus rows (fp32 upcast of the same bf16 data), torch-CPU matmul+topk, "
                      f"{t_cpu:.2f} s, scaled by rows/{nc}; parity gate on {npar} queries rtol=1e-3: ok, "
                      f"index agreement {same:.6f}"}


if __name__ == "__main__":
    main()
