"""Host-side mirror of the embedding epilogue that feeds the search / loss path, same names as the
reference:

    get_mean_token_embed   model/abc_util.py:33-41
    Linear, MLP            model/modeling_abc.py:9-34   (identity / zero initialisation, fp32 parameters)
    embed_head             F.normalize(self.mlp_head(pooled).float(), dim=-1), model/modeling_abc.py:151-154,
                           :178-179 -- the `embed` / `inst_embed` inference path, plus the bf16 / int8 corpus
                           forms the search kernels contract in

The arithmetic runs in libabcsearch.so (abc_mean_pool_*, abc_embed_head_f32) and so does its backward
(abc_mean_pool_bwd_*, abc_embed_head_bwd): under autograd the three functions stand at the training call site
model/modeling_abc.py:148-154, between the backbone's hidden states and compute_gathered_loss.  No torch-op fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import init

from . import _lib, kernels, search


def _mean_pool_fwd(input_ids, hidden_state, padding_token_id, instruction_mask, keep_fp32):
    lib = _lib.load()
    B, S, H = hidden_state.shape
    hs = hidden_state.contiguous()
    ids = input_ids.to(torch.int64).contiguous()
    im = None if instruction_mask is None else instruction_mask.to(torch.float32).contiguous()
    out = torch.empty((B, H), dtype=torch.float32, device=hs.device)
    with kernels._on(hs):
        st = kernels._stream(hs)
        if hs.dtype == torch.bfloat16:
            rc = lib.abc_mean_pool_bf16(hs.data_ptr(), ids.data_ptr(), int(padding_token_id), kernels._ptr(im), B, S, H,
                                        0 if keep_fp32 else 1, out.data_ptr(), st)
        else:
            rc = lib.abc_mean_pool_f32(hs.data_ptr(), ids.data_ptr(), int(padding_token_id), kernels._ptr(im), B, S, H,
                                       out.data_ptr(), st)
    _lib.check(rc, "abc_mean_pool")
    out = out if (keep_fp32 or hs.dtype == torch.float32) else out.to(torch.bfloat16)
    return out, ids, im


class _MeanPool(torch.autograd.Function):
    """get_mean_token_embed under autograd: d hidden[b, s, :] = ((d out[b, :] / S) * [ids != pad]) * instruction_mask."""

    @staticmethod
    def forward(ctx, hidden_state, input_ids, padding_token_id, instruction_mask, keep_fp32):
        out, ids, im = _mean_pool_fwd(input_ids, hidden_state, padding_token_id, instruction_mask, keep_fp32)
        ctx.save_for_backward(ids, im)
        ctx.meta = (tuple(hidden_state.shape), hidden_state.dtype, int(padding_token_id))
        return out

    @staticmethod
    def backward(ctx, g):
        ids, im = ctx.saved_tensors
        (B, S, H), dtype, pad = ctx.meta
        lib = _lib.load()
        g = g.to(torch.float32).contiguous()
        dh = torch.empty((B, S, H), dtype=dtype, device=g.device)
        with kernels._on(g):
            fn = lib.abc_mean_pool_bwd_bf16 if dtype == torch.bfloat16 else lib.abc_mean_pool_bwd_f32
            rc = fn(g.data_ptr(), ids.data_ptr(), pad, kernels._ptr(im), B, S, H, dh.data_ptr(), kernels._stream(g))
        _lib.check(rc, "abc_mean_pool_bwd")
        return dh, None, None, None, None


def get_mean_token_embed(input_ids: torch.Tensor, hidden_state: torch.Tensor, padding_token_id: int,
                         instruction_mask: Optional[torch.Tensor] = None, keep_fp32: bool = False) -> torch.Tensor:
    """Mean over the sequence axis of the non-pad (and instruction-masked) hidden states; divides by the
    sequence length like the reference's torch.mean.  Returns hidden_state's dtype (bf16 inputs: rounded
    as the reference's bf16 tensors are) unless keep_fp32.  Differentiable w.r.t. hidden_state."""
    kernels._need_gpu(input_ids, hidden_state, instruction_mask)
    if hidden_state.dim() != 3 or input_ids.shape != hidden_state.shape[:2]:
        raise _lib.AbcError(f"shape mismatch: ids {tuple(input_ids.shape)} vs hidden {tuple(hidden_state.shape)}")
    if hidden_state.dtype not in (torch.bfloat16, torch.float32):
        raise _lib.AbcError(f"hidden_state dtype {hidden_state.dtype} not supported (bf16 / f32)")
    if torch.is_grad_enabled() and hidden_state.requires_grad:
        return _MeanPool.apply(hidden_state, input_ids, padding_token_id, instruction_mask, keep_fp32)
    return _mean_pool_fwd(input_ids, hidden_state, padding_token_id, instruction_mask, keep_fp32)[0]


class Linear(nn.Linear):
    """modeling_abc.py:9-14: identity weight, zero bias."""

    def reset_parameters(self) -> None:
        init.eye_(self.weight)
        if self.bias is not None:
            init.zeros_(self.bias)


class MLP(nn.Module):
    """modeling_abc.py:16-34, same parameter names (state dicts load unchanged)."""

    def __init__(self, embed_size: int, hidden_size: Optional[int] = None):
        super().__init__()
        self.embed_size = embed_size
        self.hidden_size = hidden_size if hidden_size is not None else embed_size
        self.linear_layer1 = Linear(self.embed_size, self.hidden_size, dtype=torch.float32)
        self.linear_layer2 = Linear(self.hidden_size, self.embed_size, dtype=torch.float32)
        self.act = nn.SELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _head(self, x, normalize=False)[0]


def _head_params(mlp: MLP):
    w1, b1 = mlp.linear_layer1.weight, mlp.linear_layer1.bias
    w2, b2 = mlp.linear_layer2.weight, mlp.linear_layer2.bias
    for t in (w1, w2):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise _lib.AbcError("MLP parameters must be fp32 on the GPU")
    return w1, b1, w2, b2


def _head_fwd(x, w1, b1, w2, b2, normalize, want_bf16=False, want_i8=False, train=False):
    """abc_embed_head_f32 on detached operands; train: also returns h1 = selu(W1 x + b1) and ||y|| for the backward."""
    lib = _lib.load()
    B, H = x.shape
    Hd = w1.shape[0]
    dev = x.device
    out = torch.empty((B, H), dtype=torch.float32, device=dev)
    o16 = torch.empty((B, H), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    o8 = torch.empty((B, H), dtype=torch.int8, device=dev) if want_i8 else None
    sc = torch.empty((B,), dtype=torch.float32, device=dev) if want_i8 else None
    h1 = torch.empty((B, Hd), dtype=torch.float32, device=dev) if train else None
    nrm = torch.empty((B,), dtype=torch.float32, device=dev) if train else None
    if B == 0:
        return out, o16, o8, sc, h1, nrm
    need = lib.abc_embed_head_workspace_bytes(B, H, Hd)
    if need == 0:
        raise _lib.AbcError(f"embed head: unsupported sizes H={H}, hidden={Hd} (multiples of 8)")
    with kernels._on(x):
        ws = kernels.workspace(need, dev)
        rc = lib.abc_embed_head_f32(x.data_ptr(), w1.data_ptr(), kernels._ptr(b1), w2.data_ptr(), kernels._ptr(b2), B, H, Hd,
                                    1 if normalize else 0, out.data_ptr(), kernels._ptr(o16), kernels._ptr(o8),
                                    kernels._ptr(sc), kernels._ptr(h1), kernels._ptr(nrm), ws.data_ptr(), ws.numel(),
                                    kernels._stream(x))
    _lib.check(rc, "abc_embed_head_f32")
    return out, o16, o8, sc, h1, nrm


class _Head(torch.autograd.Function):
    """out = normalize(x + W2 selu(W1 x + b1) + b2) with the gradients of x and of the four parameters
    (abc_embed_head_bwd: four exact-fp32 MFMA GEMMs and the element-wise pieces between them)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, normalize: bool):
        xd, w1d, w2d = x.detach().contiguous(), w1.detach().contiguous(), w2.detach().contiguous()
        b1d = None if b1 is None else b1.detach().contiguous()
        b2d = None if b2 is None else b2.detach().contiguous()
        out, _, _, _, h1, nrm = _head_fwd(xd, w1d, b1d, w2d, b2d, normalize, train=True)
        ctx.save_for_backward(xd, w1d, w2d, out, h1, nrm)
        ctx.normalize = bool(normalize)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w1, w2, out, h1, nrm = ctx.saved_tensors
        lib = _lib.load()
        B, H = x.shape
        Hd = w1.shape[0]
        need = ctx.needs_input_grad
        dev = x.device
        g = g.to(torch.float32).contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        dx = new(B, H) if need[0] else None
        dw1 = new(Hd, H) if need[1] else None
        db1 = new(Hd) if (need[2] and ctx.has_bias[0]) else None
        dw2 = new(H, Hd) if need[3] else None
        db2 = new(H) if (need[4] and ctx.has_bias[1]) else None
        if B == 0:
            for t in (dw1, db1, dw2, db2):
                if t is not None:
                    t.zero_()
            return dx, dw1, db1, dw2, db2, None
        nbytes = lib.abc_embed_head_bwd_workspace_bytes(B, H, Hd)
        with kernels._on(x):
            ws = kernels.workspace(nbytes, dev)
            rc = lib.abc_embed_head_bwd(g.data_ptr(), out.data_ptr(), nrm.data_ptr(), x.data_ptr(), h1.data_ptr(), w1.data_ptr(),
                                        w2.data_ptr(), B, H, Hd, 1 if ctx.normalize else 0, kernels._ptr(dx), kernels._ptr(dw1),
                                        kernels._ptr(db1), kernels._ptr(dw2), kernels._ptr(db2), ws.data_ptr(), ws.numel(),
                                        kernels._stream(x))
        _lib.check(rc, "abc_embed_head_bwd")
        return dx, dw1, db1, dw2, db2, None


def _head(mlp: MLP, x: torch.Tensor, normalize: bool, want_bf16: bool = False, want_i8: bool = False):
    kernels._need_gpu(x)
    w1, b1, w2, b2 = _head_params(mlp)
    x = x.to(torch.float32)
    if x.dim() != 2 or x.shape[1] != mlp.embed_size:
        raise _lib.AbcError(f"expected [B, {mlp.embed_size}], got {tuple(x.shape)}")
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mlp.parameters())):
        if want_bf16 or want_i8:
            raise _lib.AbcError("the bf16 / int8 corpus forms are inference outputs: ask for them under torch.no_grad()")
        return _Head.apply(x, w1, b1, w2, b2, normalize), None, None, None
    det = lambda t: None if t is None else t.detach().contiguous()  # noqa: E731
    return _head_fwd(x.detach().contiguous(), det(w1), det(b1), det(w2), det(b2), normalize, want_bf16, want_i8)[:4]


def embed_head(input_ids: torch.Tensor, hidden_state: torch.Tensor, padding_token_id: int, mlp: MLP,
               instruction_mask: Optional[torch.Tensor] = None, compute: Optional[str] = None):
    """The reference's `embed` after the backbone (modeling_abc.py:172-179) and, under autograd, the training forward's
    modeling_abc.py:148-154: pooled -> mlp_head -> .float() -> F.normalize.  Returns the fp32 unit vectors [B, H]
    (differentiable w.r.t. hidden_state and the head's parameters); with compute = "bf16" / "i8" (inference) also a
    resident search.Embeddings in that form, produced in the same pass."""
    pooled = get_mean_token_embed(input_ids, hidden_state, padding_token_id, instruction_mask)
    out, o16, o8, sc = _head(mlp, pooled.float(), normalize=True, want_bf16=compute == "bf16", want_i8=compute == "i8")
    if compute is None:
        return out
    emb = object.__new__(search.Embeddings)
    emb.compute = compute
    emb.scale = sc if compute == "i8" else None
    emb.data = {"f32": out, "bf16": o16, "i8": o8}[compute]
    return out, emb
