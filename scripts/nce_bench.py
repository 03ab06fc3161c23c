"""InfoNCE fwd+bwd timing at BASELINE configs[3] (B=128 per rank, C=1024, D=4096 fp32): python scripts/nce_bench.py [iters]"""
import sys

import torch

sys.path.insert(0, ".")
from vllm_amd import loss as L

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator(device=dev).manual_seed(1)
q = torch.nn.functional.normalize(torch.randn(128, 4096, generator=g, device=dev), dim=-1).requires_grad_(True)
p = torch.nn.functional.normalize(torch.randn(1024, 4096, generator=g, device=dev), dim=-1).requires_grad_(True)
temp = L.Temperature().to(dev)
for it in range(iters + 5):
    if it == 5:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    l, a = L._loss_and_acc(q, p, temp, 0.1, 128)
    l.backward()
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / iters * 1e3:.1f} us per fwd+bwd")
