"""Host-side mirror of the embedding epilogue that feeds the search / loss path, same names as the
reference:

    get_mean_token_embed   model/abc_util.py:33-41
    Linear, MLP            model/modeling_abc.py:9-34   (identity / zero initialisation, fp32 parameters)
    embed_head             F.normalize(self.mlp_head(pooled).float(), dim=-1), model/modeling_abc.py:151-154,
                           :178-179 -- the `embed` / `inst_embed` inference path, plus the bf16 / int8 corpus
                           forms the search kernels contract in

The arithmetic runs in libabcsearch.so (abc_mean_pool_*, abc_embed_head_f32).  Inference only: the
fused head has no backward, and asking for one raises instead of falling back to torch ops.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import init

from . import _lib, kernels, search


def get_mean_token_embed(input_ids: torch.Tensor, hidden_state: torch.Tensor, padding_token_id: int,
                         instruction_mask: Optional[torch.Tensor] = None, keep_fp32: bool = False) -> torch.Tensor:
    """Mean over the sequence axis of the non-pad (and instruction-masked) hidden states; divides by the
    sequence length like the reference's torch.mean.  Returns hidden_state's dtype (bf16 inputs: rounded
    as the reference's bf16 tensors are) unless keep_fp32."""
    kernels._need_gpu(input_ids, hidden_state, instruction_mask)
    if hidden_state.dim() != 3 or input_ids.shape != hidden_state.shape[:2]:
        raise _lib.AbcError(f"shape mismatch: ids {tuple(input_ids.shape)} vs hidden {tuple(hidden_state.shape)}")
    if hidden_state.dtype not in (torch.bfloat16, torch.float32):
        raise _lib.AbcError(f"hidden_state dtype {hidden_state.dtype} not supported (bf16 / f32)")
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
    return out if (keep_fp32 or hs.dtype == torch.float32) else out.to(torch.bfloat16)


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


def _head(mlp: MLP, x: torch.Tensor, normalize: bool, want_bf16: bool = False, want_i8: bool = False):
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mlp.parameters())):
        raise _lib.AbcError("the fused embedding head is inference-only: call it under torch.no_grad()")
    kernels._need_gpu(x)
    lib = _lib.load()
    x = x.to(torch.float32).contiguous()
    if x.dim() != 2 or x.shape[1] != mlp.embed_size:
        raise _lib.AbcError(f"expected [B, {mlp.embed_size}], got {tuple(x.shape)}")
    B, H = x.shape
    Hd = mlp.hidden_size
    w1, b1 = mlp.linear_layer1.weight.detach(), mlp.linear_layer1.bias
    w2, b2 = mlp.linear_layer2.weight.detach(), mlp.linear_layer2.bias
    for t in (w1, w2):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise _lib.AbcError("MLP parameters must be fp32 on the GPU")
    w1, w2 = w1.contiguous(), w2.contiguous()
    b1 = None if b1 is None else b1.detach().contiguous()
    b2 = None if b2 is None else b2.detach().contiguous()
    out = torch.empty((B, H), dtype=torch.float32, device=x.device)
    o16 = torch.empty((B, H), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    o8 = torch.empty((B, H), dtype=torch.int8, device=x.device) if want_i8 else None
    sc = torch.empty((B,), dtype=torch.float32, device=x.device) if want_i8 else None
    if B == 0:
        return out, o16, o8, sc
    need = lib.abc_embed_head_workspace_bytes(B, H, Hd)
    if need == 0:
        raise _lib.AbcError(f"embed head: unsupported sizes H={H}, hidden={Hd} (multiples of 8)")
    with kernels._on(x):
        ws = kernels.workspace(need, x.device)
        rc = lib.abc_embed_head_f32(x.data_ptr(), w1.data_ptr(), kernels._ptr(b1), w2.data_ptr(), kernels._ptr(b2), B, H, Hd,
                                    1 if normalize else 0, out.data_ptr(), kernels._ptr(o16), kernels._ptr(o8),
                                    kernels._ptr(sc), ws.data_ptr(), ws.numel(), kernels._stream(x))
    _lib.check(rc, "abc_embed_head_f32")
    return out, o16, o8, sc


def embed_head(input_ids: torch.Tensor, hidden_state: torch.Tensor, padding_token_id: int, mlp: MLP,
               instruction_mask: Optional[torch.Tensor] = None, compute: Optional[str] = None):
    """The reference's `embed` after the backbone (modeling_abc.py:172-179): pooled -> mlp_head -> .float()
    -> F.normalize.  Returns the fp32 unit vectors [B, H]; with compute = "bf16" / "i8" also a resident
    search.Embeddings in that form, produced in the same pass."""
    pooled = get_mean_token_embed(input_ids, hidden_state, padding_token_id, instruction_mask)
    out, o16, o8, sc = _head(mlp, pooled.float(), normalize=True, want_bf16=compute == "bf16", want_i8=compute == "i8")
    if compute is None:
        return out
    emb = object.__new__(search.Embeddings)
    emb.compute = compute
    emb.scale = sc if compute == "i8" else None
    emb.data = {"f32": out, "bf16": o16, "i8": o8}[compute]
    return out, emb
