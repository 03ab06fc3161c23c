"""For scale only (not the product, not the oracle): the reference's own formulation of the search -- similarity
matrix with torch.matmul, then torch.topk (evaluate/eval_mscoco.py:18-30, pretrain/create_negative_index.py:45-52)
-- run on THIS GPU through the vendor libraries, chunked over the corpus so the scores fit, against the fused sweep.
python scripts/torch_gpu_baseline.py [corpus_rows]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from vllm_amd import kernels

dev = torch.device("cuda:0")
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
nq, d, k, chunk = 8192, 4096, 64, 262144
g = torch.Generator(device=dev).manual_seed(1)


def unit(rows):
    out = torch.empty((rows, d), dtype=torch.bfloat16, device=dev)
    for s in range(0, rows, 1 << 16):
        x = torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev)
        out[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)
    return out


q, c = unit(nq), unit(nc)


def torch_path():
    vals, idxs = [], []
    for s in range(0, nc, chunk):
        sims = torch.matmul(q, c[s:s + chunk].T)                 # bf16 scores of one chunk (4.3 GB)
        v, i = torch.topk(sims, k, dim=1)
        vals.append(v)
        idxs.append(i + s)
    v = torch.cat(vals, dim=1)
    i = torch.cat(idxs, dim=1)
    vv, sel = torch.topk(v, k, dim=1)
    return vv, torch.gather(i, 1, sel)


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


ms_t, (tv, ti) = timed(torch_path, 2)
ms_f, (fv, fi, _) = timed(lambda: kernels.topk(q, c, k), 3)
# no index comparison here: torch.matmul returns bf16-ROUNDED scores, which tie in their thousands on random data;
# parity of the sweep is tested against the fp32 oracle in tests/, not against this path
print(json.dumps({"shape": f"{nq} queries x {nc} rows x {d} bf16, k={k}",
                  "torch_matmul_topk_ms": round(ms_t, 2), "fused_sweep_ms": round(ms_f, 2),
                  "speedup": round(ms_t / ms_f, 2)}), flush=True)
