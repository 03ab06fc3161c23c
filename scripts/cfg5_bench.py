"""BASELINE.json configs[4] on its own (int8 corpus x 3584, k = 100), for rocprofv3 runs:
python3 scripts/cfg5_bench.py [rows=10000000] [iters=2]  -> one JSON line (events around the launches)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
n5 = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
D = 3584


def unit(n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn((n, D), generator=g, device=dev), dim=-1)


c8 = torch.empty((n5, D), dtype=torch.int8, device=dev)
cs = torch.empty((n5,), dtype=torch.float32, device=dev)
for s in range(0, n5, 1 << 18):
    a, b = kernels.quantize_i8(unit(min(1 << 18, n5 - s), 100 + s))
    c8[s:s + a.shape[0]], cs[s:s + a.shape[0]] = a, b
q8, qs = kernels.quantize_i8(unit(8192, 7))
kernels.topk(q8, c8, 100, q_scale=qs, c_scale=cs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = kernels.topk(q8, c8, 100, q_scale=qs, c_scale=cs)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
ops = 2.0 * 8192 * n5 * D
print(json.dumps({"config": f"cfg5 int8 8192 x {n5} x {D} k=100", "ms": round(ms, 3), "TOP/s": round(ops / ms / 1e9, 1),
                  "frac_of_5POPs": round(ops / ms / 1e9 / 5000, 4), "queries_per_s": round(8192 / ms * 1e3),
                  "algorithmic_ops_per_launch": ops}), flush=True)
