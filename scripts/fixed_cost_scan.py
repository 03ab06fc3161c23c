"""Fixed cost per launch of the sweep: one process, one box, sizes interleaved (clock drift averages out).
python3 scripts/fixed_cost_scan.py  -> table of ms per launch against corpus rows, with and without survivors"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
d, nq, k = 4096, 8192, 64
NC = 825000


def unit(n):
    out = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
    for s in range(0, n, 1 << 17):
        x = torch.randn((min(1 << 17, n - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(out.dtype)
    return out


q, c = unit(nq), unit(NC)
ub_none = torch.full((nq,), -2.0, device=dev)
sizes = [51563, 103125, 206250, 412500, 825000]
cases = [("topk", None, 0), ("nosurv", ub_none, 0), ("topk_qb16", None, 16)]
res = {}
for rep in range(4):
    for name, ub, qb in cases:
        kernels.debug_set("plan_qb", qb)
        for nc in sizes:
            cc = c[:nc]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            kernels.topk(q, cc, k, upper_bound=ub)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                res.setdefault((name, nc), []).append(e0.elapsed_time(e1))
kernels.debug_set("plan_qb", 0)
for name, _, _ in cases:
    prev = None
    for nc in sizes:
        ms = sorted(res[(name, nc)])[1]
        tiles = (nc + 255) // 256
        line = f"{name:10s} nc={nc:7d} tiles={tiles:5d} median {ms:8.3f} ms  {2.0 * nq * nc * d / ms / 1e9:7.1f} TFLOP/s"
        if prev:
            slope = (ms - prev[1]) / (tiles - prev[0])
            line += f"   slope {slope * 1e3:.2f} us/tile-column, intercept {ms - slope * tiles:.3f} ms"
        prev = (tiles, ms)
        print(line, flush=True)
