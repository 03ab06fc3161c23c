"""Where the host time of one InfoNCE step goes (cfg4 shape): python scripts/nce_host_profile.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import loss as L

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 4096, generator=g, device=dev), dim=-1)  # noqa: E731
q = unit(128).requires_grad_(True)
c_local = unit(128).requires_grad_(True)
c_global = torch.cat([c_local.detach(), unit(896)])
temp = L.Temperature().to(dev)


def step(set_none):
    l, a = L._loss_and_acc(q, c_global, temp, 0.1, 128, p_local=c_local)
    l.backward()
    if set_none:
        q.grad = c_local.grad = temp.temp.grad = None


for set_none in (True, False):
    for _ in range(20):
        step(set_none)
    torch.cuda.synchronize()
    n = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        step(set_none)
    host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    print(f"grads set to None={set_none}: host {host / n * 1e6:.1f} us/step issue time, device {e0.elapsed_time(e1) / n * 1e3:.1f} us/step")

# host cost of the pieces
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    l, a = L._loss_and_acc(q, c_global, temp, 0.1, 128, p_local=c_local)
torch.cuda.synchronize()
print(f"forward only: {(time.perf_counter() - t0) / n * 1e6:.1f} us/step (host+device, no backward)")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step(True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
