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
                        ws.data_ptr(), ws.numel(), st)
    lib.abc_infonce_bwd(qd.data_ptr(), pd.data_ptr(), lse.data_ptr(), t.data_ptr(), 0.1, go.data_ptr(), B, C, D, 128,
                        dq.data_ptr(), dp.data_ptr(), dt.data_ptr(), ws.data_ptr(), ws.numel(), st)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / iters * 1e3:.1f} us per fwd+bwd through the C ABI (memset + 2 kernels)")
