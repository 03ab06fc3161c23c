#!/usr/bin/env python3
"""One-shot diagnosis of the hang recorded in round 1 (commit 9e18df9): gloo collectives on DEVICE tensors with
several ranks on one GPU.  Each case runs in fresh child processes under a watchdog that dumps every thread's
Python stack and exits, so a hang costs CASE_TIMEOUT seconds and leaves a trace instead of a stuck box.

    python scripts/gloo_device_diag.py [world=2]      -> gpurun_out/gloo_diag.log

Not a test and not run by the driver: it exists to document WHY vllm_amd stages device tensors through host
memory under a gloo group (parallel._host_staged, loss._all_gather_rows)."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CASE_TIMEOUT = 25


def _case(rank, world, port, name, log):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    f = open(f"{log}.{name}.r{rank}", "w")
    faulthandler.dump_traceback_later(CASE_TIMEOUT, exit=True, file=f)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.full((4, 8), float(rank), device=dev)
    t0 = time.time()
    if name == "all_gather_into_tensor_device":
        out = torch.empty((4 * world, 8), device=dev)
        dist.all_gather_into_tensor(out, x)
    elif name == "all_gather_list_device":
        outs = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(outs, x)
        out = torch.cat(outs)
    elif name == "all_reduce_device":
        out = x.clone()
        dist.all_reduce(out)
    elif name == "all_gather_into_tensor_device_async":
        out = torch.empty((4 * world, 8), device=dev)
        w = dist.all_gather_into_tensor(out, x, async_op=True)
        y = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)   # work enqueued behind it
        w.wait()
        del y
    elif name == "bench_pattern":
        # what ShardedCorpus.topk did in round 1: per chunk TWO async all-gathers (f32 values, i64 indices, [8192, 64])
        # behind a long kernel, several chunks pending, every wait() at the end
        pend = []
        a = torch.randn(8192, 8192, device=dev)
        for ch in range(4):
            for _ in range(6):
                a = (a @ a).clamp_(-1, 1)                       # ~10 ms of GPU work per chunk
            v = torch.full((8192, 64), float(rank + ch), device=dev)
            i = torch.full((8192, 64), rank + ch, dtype=torch.int64, device=dev)
            gv = torch.empty((world * 8192, 64), device=dev)
            gi = torch.empty((world * 8192, 64), dtype=torch.int64, device=dev)
            pend.append((dist.all_gather_into_tensor(gv, v, async_op=True),
                         dist.all_gather_into_tensor(gi, i, async_op=True), gv, gi, v, i))
        out = torch.zeros((), device=dev)
        for wv, wi, gv, gi, v, i in pend:
            wv.wait()
            wi.wait()
            out = out + gv.sum() + gi.sum()
    elif name == "all_reduce_device_after_kernels":
        a = torch.randn(8192, 8192, device=dev)
        for _ in range(12):
            a = (a @ a).clamp_(-1, 1)
        out = torch.zeros((8192,), device=dev) + rank
        dist.all_reduce(out)
    elif name == "all_gather_into_tensor_host":
        h = x.cpu()
        out = torch.empty((4 * world, 8))
        dist.all_gather_into_tensor(out, h)
        out = out.to(dev)
    torch.cuda.synchronize()
    f.write(f"{name}: rank {rank} done in {time.time() - t0:.3f} s, sum {float(out.sum())}\n")
    f.flush()
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    log = os.path.join("gpurun_out", f"gloo_diag_w{world}")
    os.makedirs("gpurun_out", exist_ok=True)
    summary = open(log + ".log", "w")
    summary.write(f"torch {torch.__version__}, world {world}, all ranks on cuda:0, backend gloo\n")
    cases = ["all_gather_into_tensor_host", "all_reduce_device", "all_gather_list_device",
             "all_gather_into_tensor_device", "all_gather_into_tensor_device_async", "all_reduce_device_after_kernels",
             "bench_pattern"]
    for i, name in enumerate(cases):
        t0 = time.time()
        try:
            mp.spawn(_case, args=(world, 29810 + i, name, log), nprocs=world, join=True)
            status = "ok"
        except Exception as e:                                   # a child hit the watchdog (exit code 1) or raised
            status = f"FAILED ({type(e).__name__}: {str(e).splitlines()[0][:160]})"
        summary.write(f"{name}: {status} after {time.time() - t0:.1f} s\n")
        for r in range(world):
            p = f"{log}.{name}.r{r}"
            if os.path.exists(p):
                summary.write(f"--- rank {r}\n" + open(p).read())
                os.remove(p)
        summary.flush()
    summary.close()
    print(open(log + ".log").read())


if __name__ == "__main__":
    main()
