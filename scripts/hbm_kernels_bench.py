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
del hs, ids
# the embedding head (exact-fp32 MFMA GEMMs, 157 TFLOP/s peak): forward = 2 GEMMs, backward = 4, each 2 B H Hd flop
for Bh in (288, 8192):
    H = Hd = 3584
    mlp = head.MLP(H, Hd).to(dev)
    with torch.no_grad():
        mlp.linear_layer1.weight.add_(0.01 * torch.randn(Hd, H, device=dev))
        mlp.linear_layer2.weight.add_(0.01 * torch.randn(H, Hd, device=dev))
    xh = torch.randn((Bh, H), device=dev)
    with torch.no_grad():
        ms = timed(lambda: head._head(mlp, xh, normalize=True))
    print(f"head forward   {Bh} x {H} x {Hd}: {ms:.3f} ms  {4.0 * Bh * H * Hd / ms / 1e9:.1f} TFLOP/s fp32")
    xg = xh.clone().requires_grad_(True)
    g = torch.randn((Bh, H), device=dev)

    def fb():
        out = head._head(mlp, xg, normalize=True)[0]
        out.backward(g)
    ms2 = timed(fb)
    print(f"head fwd + bwd {Bh} x {H} x {Hd}: {ms2:.3f} ms  backward alone ~{ms2 - ms:.3f} ms = {8.0 * Bh * H * Hd / (ms2 - ms) / 1e9:.1f} TFLOP/s fp32")
