"""Ad-hoc timing of the sweep kernel: python scripts/quick_bench.py nq nc d k [dtype] [iters]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from vllm_amd import kernels

nq, nc, d, k = (int(x) for x in sys.argv[1:5])
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def unit(n):
    out = torch.empty((n, d), dtype=torch.bfloat16 if dtype != "f32" else torch.float32, device=dev)
    for s in range(0, n, 1 << 18):
        x = torch.randn((min(1 << 18, n - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(out.dtype)
    return out


q, c = unit(nq), unit(nc)
kw = {}
if dtype == "i8":
    q, qs = kernels.quantize_i8(q.float())
    cc = []
    css = []
    for s in range(0, nc, 1 << 18):
        a, b = kernels.quantize_i8(c[s:s + (1 << 18)].float())
        cc.append(a)
        css.append(b)
    c, cs = torch.cat(cc), torch.cat(css)
    kw = dict(q_scale=qs, c_scale=cs)
import os
if os.environ.get("QB_UB"):                      # a constant upper bound for every query (e.g. -2: nothing survives the filter)
    kw["upper_bound"] = torch.full((nq,), float(os.environ["QB_UB"]), device=dev)
torch.cuda.synchronize()
for it in range(iters + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    v, i, _ = kernels.topk(q, c, k, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = 2.0 * nq * nc * d
    print(f"iter {it}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  {nq / ms * 1e3:.1f} q/s", flush=True)
