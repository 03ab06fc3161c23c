"""Timing of the rank-counting (COUNT) sweep variants at the bench shape: python scripts/count_bench.py"""
import sys, torch
sys.path.insert(0, ".")
from vllm_amd import kernels
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
nq, nc, d = 8192, 3300000, 4096
def unit(n):
    out = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
    for s in range(0, n, 1 << 17):
        x = torch.randn((min(1 << 17, n - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)
    return out
q, c = unit(nq), unit(nc)
ref = torch.full((nq,), 0.05, device=dev)
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    v, i, r = kernels.topk(q, c, 10, rank_ref=ref)
    e1.record(); torch.cuda.synchronize()
    print(f"iter {it}: {e0.elapsed_time(e1):.2f} ms rank_sum={int(r.sum())}", flush=True)
