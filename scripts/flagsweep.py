"""Timing of one data set under several ABC_DEBUG_FLAGS values in ONE process:
python scripts/flagsweep.py nq nc d k iters flag [flag ...]   (the library reads the flag at every call)"""
import os
import sys

import torch

sys.path.insert(0, ".")
from vllm_amd import kernels

nq, nc, d, k, iters = (int(x) for x in sys.argv[1:6])
flags = sys.argv[6:]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def unit(n):
    out = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
    for s in range(0, n, 1 << 18):
        x = torch.randn((min(1 << 18, n - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(out.dtype)
    return out


q, c = unit(nq), unit(nc)
torch.cuda.synchronize()
for rnd in range(iters):
    for f in flags:
        kernels.debug_set("flags", int(f))        # the environment is read once, at load: the knob is set through the hook
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        kernels.topk(q, c, k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"round {rnd} flags={f}: {ms:.3f} ms  {2.0 * nq * nc * d / ms / 1e9:.1f} TFLOP/s", flush=True)
