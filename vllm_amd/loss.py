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


class _InfoNCE(torch.autograd.Function):
    """loss, argmax = InfoNCE(q, p, temp); gradient reaches q, the first `local_rows` rows of p, temp."""

    @staticmethod
    def forward(ctx, q, p, temp, eps: float, local_rows: int):
        if not (q.is_cuda and p.is_cuda):
            raise _lib.AbcError("InfoNCE kernels run on the GPU only (no CPU fallback)")
        lib = _lib.load()
        q, p = q.float().contiguous(), p.float().contiguous()
        B, d = q.shape
        C = p.shape[0]
        need = lib.abc_infonce_workspace_bytes(B, C, d)
        if need == 0:
            raise _lib.AbcError(f"InfoNCE: unsupported shape B={B} C={C} d={d} (need B <= C, d % 8 == 0)")
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)      # also carries the logits to backward
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        lse = torch.empty((B,), dtype=torch.float32, device=q.device)
        amax = torch.empty((B,), dtype=torch.int32, device=q.device)
        t = temp.detach().to(device=q.device, dtype=torch.float32).reshape(1).contiguous()   # stays on the device
        with _on(q):
            rc = lib.abc_infonce_fwd(q.data_ptr(), p.data_ptr(), t.data_ptr(), float(eps), B, C, d, loss.data_ptr(),
                                     lse.data_ptr(), amax.data_ptr(), ws.data_ptr(), ws.numel(), _stream(q))
        _lib.check(rc, "abc_infonce_fwd")
        ctx.save_for_backward(q, p, lse, ws, t)
        ctx.meta = (float(eps), B, C, d, int(local_rows), temp.dtype)
        ctx.mark_non_differentiable(amax)
        return loss, amax

    @staticmethod
    def backward(ctx, grad_loss, _grad_amax):
        q, p, lse, ws, t = ctx.saved_tensors
        eps, B, C, d, local_rows, tdtype = ctx.meta
        lib = _lib.load()
        need_q, need_p, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        go = grad_loss.detach().float().contiguous()
        dq = torch.empty_like(q) if need_q else None
        dp = None
        if need_p:
            dp = torch.zeros_like(p) if local_rows < C else torch.empty_like(p)
        dt = torch.empty((), dtype=torch.float32, device=q.device) if need_t else None
        with _on(q):
            rc = lib.abc_infonce_bwd(q.data_ptr(), p.data_ptr(), lse.data_ptr(), t.data_ptr(), eps, go.data_ptr(), B, C, d,
                                     local_rows if need_p else 0, None if dq is None else dq.data_ptr(),
                                     None if dp is None else dp.data_ptr(), None if dt is None else dt.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream(q))
        _lib.check(rc, "abc_infonce_bwd")
        return dq, dp, (None if dt is None else dt.to(tdtype)), None, None


def _temp_param(temperature) -> torch.Tensor:
    t = getattr(temperature, "temp", temperature)
    if not torch.is_tensor(t):
        t = torch.tensor(float(t), dtype=torch.float32)
    return t


def _loss_and_acc(q_embeds, p_embeds, temperature, label_smoothing, local_rows):
    temp = _temp_param(temperature)
    loss, amax = _InfoNCE.apply(q_embeds, p_embeds, temp.to(q_embeds.device), float(label_smoothing), local_rows)
    bs = q_embeds.size(0)
    targets = torch.arange(bs, device=amax.device, dtype=amax.dtype)
    accuracy = (amax == targets).sum() / bs
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
    Returns (c_global [W*n, D] -- rows [0, n) are the local, grad-carrying block -- and n)."""
    n = c_emb.size(0)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return c_emb, n
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = _all_gather_rows(c_emb)                                       # one collective, no gradient
    gathered = gathered.view((world,) + tuple(c_emb.shape))
    blocks = [c_emb if r == rank else gathered[r] for r in _slot_order(world, rank)]
    return torch.cat(blocks, dim=0), n


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
        return loss, acc, torch.tensor([c_all.size(0)], device=loss.device)
    c_global, n_local = gathered_candidates(c_emb)
    if c_global is c_emb:
        loss, acc = _loss_and_acc(q_emb, c_emb, temperature, label_smoothing, n_local)
    else:
        # the gathered blocks are constants: only rows [0, n_local) of c_global get a gradient, and the
        # kernel writes exactly those (torch.cat routes them back to c_emb)
        loss, acc = _loss_and_acc(q_emb, c_global, temperature, label_smoothing, n_local)
    num_cand = torch.tensor([c_global.size(0)], device=loss.device)
    return loss, acc, num_cand
