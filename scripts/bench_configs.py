"""Timings of the five BASELINE.json configs on one MI355X (parity-test shapes, not the headline bench).
python scripts/bench_configs.py [--small]  -> JSON lines"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from vllm_amd import kernels, search, loss as L

dev = torch.device("cuda:0")
small = "--small" in sys.argv


def unit(n, d, seed, dtype=torch.float32):
    g = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty((n, d), dtype=dtype, device=dev)
    for s in range(0, n, 1 << 17):
        x = torch.randn((min(1 << 17, n - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(dtype)
    return out


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, ms, flops=None, **kw):
    r = {"config": name, "ms": round(ms, 4)}
    if flops:
        r["TFLOP/s"] = round(flops / ms / 1e9, 1)
    r.update(kw)
    print(json.dumps(r), flush=True)


# cfg1: MSCOCO 5k: 5000 x 25000 (i2t) and 25000 x 5000 (t2i), fp32, k=10 (one search serves k=1,5,10)
img, txt = unit(5000, 4096, 1), unit(25000, 4096, 2)
for nm, a, b in (("cfg1 MSCOCO i2t 5000x25000 f32 k=10", img, txt), ("cfg1 MSCOCO t2i 25000x5000 f32 k=10", txt, img)):
    ms = timeit(lambda: kernels.topk(a, b, 10))
    report(nm, ms, 2.0 * a.shape[0] * b.shape[0] * 4096, queries_per_s=round(a.shape[0] / ms * 1e3))
# cfg2: Flickr 1k: 1000 x 5000 and 5000 x 1000, bf16, k=10
fi, ft = unit(1000, 4096, 3, torch.bfloat16), unit(5000, 4096, 4, torch.bfloat16)
for nm, a, b in (("cfg2 Flickr i2t 1000x5000 bf16 k=10", fi, ft), ("cfg2 Flickr t2i 5000x1000 bf16 k=10", ft, fi)):
    ms = timeit(lambda: kernels.topk(a, b, 10), iters=20)
    report(nm, ms, 2.0 * a.shape[0] * b.shape[0] * 4096, queries_per_s=round(a.shape[0] / ms * 1e3))
del img, txt
# cfg4: InfoNCE B=128, C=1024, D=4096 fp32 fwd+bwd (one rank's work; the all-gather is separate)
q = unit(128, 4096, 5).requires_grad_(True)
p = unit(1024, 4096, 6).requires_grad_(True)
temp = L.Temperature().to(dev)


def nce():
    l, a = L._loss_and_acc(q, p, temp, 0.1, 128)
    l.backward()


ms = timeit(nce, iters=20)
report("cfg4 InfoNCE fwd+bwd B=128 C=1024 D=4096 f32 (autograd.Function, incl. torch overhead)", ms,
       2.0 * 128 * 1024 * 4096 * 2 + 2.0 * 128 * 128 * 4096)


def nce_ref():
    s = (q @ p.t()) / temp.temp
    l = torch.nn.functional.cross_entropy(s, torch.arange(128, device=dev), label_smoothing=0.1)
    l.backward()


ms = timeit(nce_ref, iters=20)
report("cfg4 same math in eager torch (rocBLAS + aten), for scale only", ms)
# cfg5: int8 corpus x 3584, k=100 (full 10M x 3584 = 35.8 GB fits; --small uses 1M)
n5 = 1_000_000 if small else 10_000_000
c8 = torch.empty((n5, 3584), dtype=torch.int8, device=dev)
cs = torch.empty((n5,), dtype=torch.float32, device=dev)
for s in range(0, n5, 1 << 18):
    x = unit(min(1 << 18, n5 - s), 3584, 100 + s)
    a, b = kernels.quantize_i8(x)
    c8[s:s + a.shape[0]], cs[s:s + a.shape[0]] = a, b
q8, qs = kernels.quantize_i8(unit(8192, 3584, 7))
ms = timeit(lambda: kernels.topk(q8, c8, 100, q_scale=qs, c_scale=cs), iters=2, warm=1)
report(f"cfg5 int8 8192 x {n5} x 3584 k=100", ms, 2.0 * 8192 * n5 * 3584, queries_per_s=round(8192 / ms * 1e3),
       frac_of_5POPs=round(2.0 * 8192 * n5 * 3584 / ms / 1e9 / 5000, 3))
