"""InfoNCE fwd+bwd timing at BASELINE configs[3] (B=128 per rank, C=1024, D=4096 fp32): python scripts/nce_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import loss as L

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator(device=dev).manual_seed(1)
q = torch.nn.functional.normalize(torch.randn(128, 4096, generator=g, device=dev), dim=-1).requires_grad_(True)
p = torch.nn.functional.normalize(torch.randn(1024, 4096, generator=g, device=dev), dim=-1).requires_grad_(True)
temp = L.Temperature().to(dev)
def timed(fn):
    for it in range(iters + 5):
        if it == 5:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# (a) the call the trainer makes (compute_gathered_loss on one rank of eight): the candidate matrix is the gathered
# [1024, 4096] block without autograd history, the rank's own [128, 4096] block is the leaf that receives the gradient
c_local = p.detach()[:128].clone().requires_grad_(True)
c_global = torch.cat([c_local.detach(), p.detach()[128:]])


def trainer_step():
    l, a = L._loss_and_acc(q, c_global, temp, 0.1, 128, p_local=c_local)
    l.backward()
    q.grad = c_local.grad = temp.temp.grad = None          # optimizer.zero_grad(set_to_none=True)


print(f"{timed(trainer_step):.1f} us per fwd+bwd through loss._loss_and_acc as compute_gathered_loss calls it (own block = leaf)")


# (b) the whole [1024, 4096] matrix as ONE leaf of which 128 rows are local: torch then zero-fills and accumulates a
# 16 MB gradient around the kernels (slice backward + grad accumulation) -- round 2's measurement
def full_leaf_step():
    l, a = L._loss_and_acc(q, p, temp, 0.1, 128)
    l.backward()


print(f"{timed(full_leaf_step):.1f} us per fwd+bwd with the full candidate matrix as one leaf (gradients accumulated)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

# the same through the C ABI alone (no autograd bookkeeping, no gradient accumulation kernels): what the library costs
from vllm_amd import _lib  # noqa: E402

lib = _lib.load()
B, C, D = 128, 1024, 4096
qd, pd = q.detach(), p.detach()
ws = torch.empty(lib.abc_infonce_workspace_bytes(B, C, D), dtype=torch.uint8, device=dev)
loss = torch.empty((), device=dev)
lse = torch.empty(B, device=dev)
am = torch.empty(B, dtype=torch.int32, device=dev)
t = torch.tensor([0.07], device=dev)
go = torch.ones((), device=dev)
dq, dp, dt = torch.empty_like(qd), torch.empty_like(pd), torch.empty((), device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(iters + 5):
    if it == 5:
        torch.cuda.synchronize()
        e0.record()
    lib.abc_infonce_fwd(qd.data_ptr(), pd.data_ptr(), t.data_ptr(), 0.1, B, C, D, loss.data_ptr(), lse.data_ptr(), am.data_ptr(),
                        None, ws.data_ptr(), ws.numel(), st)
    lib.abc_infonce_bwd(qd.data_ptr(), pd.data_ptr(), lse.data_ptr(), t.data_ptr(), 0.1, go.data_ptr(), B, C, D, 128,
                        dq.data_ptr(), dp.data_ptr(), dt.data_ptr(), ws.data_ptr(), ws.numel(), st)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / iters * 1e3:.1f} us per fwd+bwd through the C ABI (memset + 2 kernels)")
