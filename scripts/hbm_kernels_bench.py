"""Achieved HBM bandwidth of the library's memory-bound kernels (algorithmic bytes / time, DESIGN.md 4.2's table).
python scripts/hbm_kernels_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from vllm_amd import head, kernels  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n, d = 1 << 20, 4096
x = torch.randn((n, d), device=dev)                                   # 17.2 GB fp32
ms = timed(lambda: kernels.cast_bf16(x))
print(f"cast_bf16      {n} x {d}: {ms:.3f} ms  {n * d * 6 / ms / 1e9:.2f} TB/s  (4 B read + 2 B written per element)")
ms = timed(lambda: kernels.quantize_i8(x))
print(f"quantize_i8    {n} x {d}: {ms:.3f} ms  {n * d * 5 / ms / 1e9:.2f} TB/s  (4 B read + 1 B written per element)")
xb = kernels.cast_bf16(x)
del x
ms = timed(lambda: kernels.rowdot(xb, xb))
print(f"rowdot bf16    {n} x {d}: {ms:.3f} ms  {n * d * 4 / ms / 1e9:.2f} TB/s  (2 x 2 B read per element; q == c here: half of it can hit the L2)")
yb = xb.flip(0).contiguous()
ms = timed(lambda: kernels.rowdot(xb, yb))
print(f"rowdot bf16    distinct operands: {ms:.3f} ms  {n * d * 4 / ms / 1e9:.2f} TB/s")
del xb, yb
B, S, H = 512, 1024, 3584
hs = torch.randn((B, S, H), device=dev).to(torch.bfloat16)            # 3.8 GB
ids = torch.randint(1, 1000, (B, S), device=dev)
ms = timed(lambda: head.get_mean_token_embed(ids, hs, 0))
print(f"mean_pool bf16 {B} x {S} x {H}: {ms:.3f} ms  {B * S * H * 2 / ms / 1e9:.2f} TB/s  (every position valid)")
