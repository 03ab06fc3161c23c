"""Yardstick, not product: what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on this box for the
score GEMM alone (bf16 in, bf16 out, no top-k), on the same kind of data the bench uses and on zeros.
python scripts/gemm_yardstick.py [rows_per_call]"""
import json
import sys

import torch

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nq, d = 8192, 4096
g = torch.Generator(device=dev).manual_seed(1)


def unit(rows):
    out = torch.empty((rows, d), dtype=torch.bfloat16, device=dev)
    for s in range(0, rows, 1 << 16):
        x = torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)
    return out


for name, q, c in (("unit-norm gaussian rows", unit(nq), unit(n)),
                   ("zeros", torch.zeros((nq, d), dtype=torch.bfloat16, device=dev),
                    torch.zeros((n, d), dtype=torch.bfloat16, device=dev))):
    out = torch.empty((nq, n), dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        torch.matmul(q, c.T, out=out)
    torch.cuda.synchronize()
    iters = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(q, c.T, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(json.dumps({"gemm": f"{nq} x {n} x {d} bf16 NT, bf16 out", "data": name, "ms": round(ms, 3),
                      "TFLOP/s": round(2.0 * nq * n * d / ms / 1e9, 1)}), flush=True)
