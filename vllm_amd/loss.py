"""Host-side mirror of the reference's contrastive loss, same names and argument meaning.

    Temperature               model/modeling_abc.py:36-43
    compute_contrastive_loss  model/abc_util.py:43-56
    compute_gathered_loss     model/abc_util.py:5-31

The logits / logsumexp / label-smoothed cross entropy / argmax and their backward run in
libabcsearch.so (abc_infonce_fwd / abc_infonce_bwd, exact fp32 on the matrix cores); the candidate
all-gather is torch.distributed (RCCL over xGMI on ROCm).  No CPU path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib
from .kernels import _on, _stream


class Temperature(torch.nn.Module):
    """Scalar learned temperature, init 0.07; forward(x) = x / temp (model/modeling_abc.py:36-43)."""

    def __init__(self, init: float = 0.07):
        super().__init__()
        self.temp = torch.nn.Parameter(torch.tensor(init, requires_grad=True, dtype=torch.float32))

    def forward(self, x):
        return x / (self.temp.float())


# Workspaces (logits for the backward, split-K slabs, statistics, counters: ~9 MB at B=128 x C=1024) are leased from a
# small pool per (device, stream, size) instead of allocated per call: a lease returns to the pool when the autograd
# node that holds it dies (after backward, or when the loss is dropped), so two losses alive at once get two buffers and
# a training loop reuses one.  Work on one stream runs in order, which is what makes reuse within a key safe.
_ws_pool: dict = {}
_WS_POOL_MAX = 4


class _Lease:
    __slots__ = ("key", "buf")

    def __init__(self, nbytes: int, device: torch.device, stream: int):
        self.key = (device.index, stream, nbytes)
        free = _ws_pool.get(self.key)
        self.buf = free.pop() if free else torch.empty(nbytes, dtype=torch.uint8, device=device)

    def __del__(self):
        try:
            free = _ws_pool.setdefault(self.key, [])
            if len(free) < _WS_POOL_MAX:
                free.append(self.buf)
        except Exception:                                    # interpreter shutdown
            pass


def release_workspaces() -> None:
    """Drop the pooled InfoNCE workspaces (buffers held by live autograd graphs stay with them)."""
    _ws_pool.clear()


def _f32c(x: torch.Tensor) -> torch.Tensor:
    return x if (x.dtype is torch.float32 and x.is_contiguous()) else x.float().contiguous()


class _InfoNCE(torch.autograd.Function):
    """loss, accuracy = InfoNCE(q, p, temp).  `p` [C, d] is the full candidate matrix and carries NO gradient
    itself; `p_local` is the tensor whose rows are p[:n] and that receives their gradient (p itself when every row is
    local, the rank's own block under the gather: remote blocks are constants, model/abc_util.py:19-21)."""

    @staticmethod
    def forward(ctx, q, p, p_local, temp, eps: float):
        if not (q.is_cuda and p.is_cuda):
            raise _lib.AbcError("InfoNCE kernels run on the GPU only (no CPU fallback)")
        lib = _lib.load()
        q, p = _f32c(q), _f32c(p)
        B, d = q.shape
        C = p.shape[0]
        need = lib.abc_infonce_workspace_bytes(B, C, d)
        if need == 0:
            raise _lib.AbcError(f"InfoNCE: unsupported shape B={B} C={C} d={d} (need B <= C, d % 8 == 0)")
        dev = q.device
        # the device scalar the kernels read: the parameter's own storage when it is an fp32 tensor on this device
        t = temp if (temp.dtype is torch.float32 and temp.device == dev and temp.numel() == 1) else \
            temp.detach().to(device=dev, dtype=torch.float32).reshape(1)
        # loss and accuracy are tensors of their own (a caller may change them in place, as it may the reference's);
        # logsumexp and argmax stay internal: [lse[B] f32, argmax[B] i32 bits]
        loss = torch.empty((), dtype=torch.float32, device=dev)
        acc = torch.empty((), dtype=torch.float32, device=dev)
        aux = torch.empty(2 * B, dtype=torch.float32, device=dev)
        with _on(q):
            st = _stream(q)
            lease = _Lease(need, dev, st)
            ws = lease.buf
            base = aux.data_ptr()
            rc = lib.abc_infonce_fwd(q.data_ptr(), p.data_ptr(), t.data_ptr(), eps, B, C, d, loss.data_ptr(), base,
                                     base + 4 * B, acc.data_ptr(), ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "abc_infonce_fwd")
        ctx.save_for_backward(q, p, t)
        ctx.keep = (aux, lease)                                              # lse lives in `aux`, the logits in the lease
        ctx.meta = (eps, B, C, d, int(p_local.shape[0]), temp.dtype)
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, grad_loss, _grad_acc):
        q, p, t = ctx.saved_tensors
        aux, lease = ctx.keep
        eps, B, C, d, local_rows, tdtype = ctx.meta
        lib = _lib.load()
        need_q, need_p, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        go = _f32c(grad_loss)
        dq = torch.empty_like(q) if need_q else None
        dp = torch.empty((local_rows, d), dtype=torch.float32, device=q.device) if need_p else None
        dt = torch.empty((), dtype=torch.float32, device=q.device) if need_t else None
        ws = lease.buf
        with _on(q):
            rc = lib.abc_infonce_bwd(q.data_ptr(), p.data_ptr(), aux.data_ptr(), t.data_ptr(), eps, go.data_ptr(), B, C, d,
                                     local_rows if need_p else 0, None if dq is None else dq.data_ptr(),
                                     None if dp is None else dp.data_ptr(), None if dt is None else dt.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream(q))
        _lib.check(rc, "abc_infonce_bwd")
        if dt is not None and tdtype is not torch.float32:
            dt = dt.to(tdtype)
        return dq, None, dp, dt, None


def _temp_param(temperature) -> torch.Tensor:
    t = getattr(temperature, "temp", temperature)
    if not torch.is_tensor(t):
        t = torch.tensor(float(t), dtype=torch.float32)
    return t


def _loss_and_acc(q_embeds, p_embeds, temperature, label_smoothing, local_rows, p_local=None):
    """p_embeds: the full candidate matrix; its first `local_rows` rows are `p_local` (default: p_embeds[:local_rows]),
    the only rows that receive a gradient."""
    temp = _temp_param(temperature)
    if temp.device != q_embeds.device:
        temp = temp.to(q_embeds.device)
    if p_local is None:
        p_local = p_embeds if local_rows == p_embeds.shape[0] else p_embeds[:local_rows]
    loss, accuracy = _InfoNCE.apply(q_embeds, p_embeds.detach(), p_local, temp, float(label_smoothing))
    return loss, accuracy


def compute_contrastive_loss(q_embeds, p_embeds, temperature, label_smoothing=0.0):
    """score = temperature(q @ p.T); cross entropy against arange(bs) with label smoothing; argmax
    accuracy.  Same call and return as model/abc_util.py:43-56; differentiable w.r.t. q_embeds,
    p_embeds and temperature.temp."""
    return _loss_and_acc(q_embeds, p_embeds, temperature, label_smoothing, p_embeds.size(0))


def _all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """[n, ...] -> [world * n, ...], ONE collective.  RCCL (backend "nccl") takes the device tensor as it is; under
    a gloo group (CPU-side tests, one-GPU rehearsals) device tensors cross in host memory, because gloo's
    collectives on device tensors do not complete on this stack (vllm_amd/parallel.py)."""
    world = dist.get_world_size()
    x = x.detach().contiguous()
    staged = x.is_cuda and dist.get_backend() == "gloo"
    src = x.cpu() if staged else x
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    return out.to(x.device) if staged else out


def gathered_candidates(c_emb: torch.Tensor):
    """all_gather of the candidate block, local block swapped into slot 0 (model/abc_util.py:15-27).
    Returns (c_global [W*n, D] -- rows [0, n) are the local block -- and n).  c_global carries no autograd history:
    the loss function takes c_emb itself as the tensor rows [0, n) send their gradient to."""
    n = c_emb.size(0)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return c_emb, n
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = _all_gather_rows(c_emb)                                       # one collective, no gradient
    order = _slot_order(world, rank)
    if rank == 0:
        return gathered, n                                                   # already [c_0, c_1, ...]: slot 0 is the own block
    gathered = gathered.view((world,) + tuple(c_emb.shape))
    gathered[[0, rank]] = gathered[[rank, 0]]                                # swap slots 0 and rank in place (2 blocks move)
    assert order[0] == rank
    return gathered.view((world * n,) + tuple(c_emb.shape[1:])), n


def _slot_order(world: int, rank: int):
    """Which rank's block sits in each slot of c_global on `rank`: own block first, rank 0's block in
    slot `rank` (model/abc_util.py:24-27)."""
    order = list(range(world))
    order[0], order[rank] = order[rank], order[0]
    return order


class _AllGatherWithGrad(torch.autograd.Function):
    """all_gather whose backward is a reduce-scatter(SUM): every rank's loss sends a gradient to every
    candidate block.  Opt-in; the reference's gather is gradient-free
    (model/abc_util.py:19-21), UniIR's is this one (torch.distributed.nn.all_gather, clip_sf.py:103)."""

    @staticmethod
    def forward(ctx, c_emb):
        world, rank = dist.get_world_size(), dist.get_rank()
        gathered = _all_gather_rows(c_emb).view((world,) + tuple(c_emb.shape))
        ctx.order = _slot_order(world, rank)
        ctx.shape = tuple(c_emb.shape)
        return torch.cat([gathered[r] for r in ctx.order], dim=0)

    @staticmethod
    def backward(ctx, g):
        world, rank = dist.get_world_size(), dist.get_rank()
        gv = g.contiguous().view((world,) + ctx.shape)
        canon = torch.empty_like(gv)                       # slot j holds the gradient for rank order[j]'s block
        for j, r in enumerate(ctx.order):
            canon[r] = gv[j]
        if dist.get_backend() == "nccl":                   # RCCL: one reduce-scatter over xGMI
            out = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, canon.view((world * ctx.shape[0],) + ctx.shape[1:]), op=dist.ReduceOp.SUM)
            return out
        # gloo (CPU tests, one-GPU rehearsals) has no reduce-scatter; device tensors cross in host memory
        h = canon.cpu() if canon.is_cuda else canon
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        return h[rank].to(g.device).clone()


def compute_gathered_loss(q_emb, c_emb, temperature, label_smoothing=0.0, differentiable_gather=False):
    """Compute the loss by gathering candidates across GPUs; the first [batch_size] entries of c_emb
    are the positives.  Same call and return as model/abc_util.py:5-31: (loss, accuracy, num_cand);
    queries are not gathered and only the local candidate block receives a gradient.

    differentiable_gather=True (not in the reference, off by default) lets the gradient of every
    rank's loss reach every candidate block: backward adds one reduce-scatter of the [W*n, D] gradient."""
    q_emb = q_emb.float()
    c_emb = c_emb.float()
    if differentiable_gather and dist.is_initialized() and dist.get_world_size() > 1:
        c_all = _AllGatherWithGrad.apply(c_emb)
        loss, acc = _loss_and_acc(q_emb, c_all, temperature, label_smoothing, c_all.size(0))
        return loss, acc, _num_cand(c_all.size(0), loss.device)
    c_global, n_local = gathered_candidates(c_emb)
    # the gathered blocks are constants: only rows [0, n_local) of c_global (= c_emb) get a gradient, and the backward
    # kernel writes exactly those rows
    loss, acc = _loss_and_acc(q_emb, c_global, temperature, label_smoothing, n_local, p_local=c_emb)
    return loss, acc, _num_cand(c_global.size(0), loss.device)


def _num_cand(n: int, device) -> torch.Tensor:
    """torch.tensor([n], device=...) of the reference (abc_util.py:30) without the host-to-device copy: a fill."""
    return torch.full((1,), n, dtype=torch.int64, device=device)
