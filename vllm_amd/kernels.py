"""Tensor-level entry points over the C ABI: torch owns device memory and streams, the HIP library
does the arithmetic.  Every function raises if the tensors are not on a ROCm device."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_DT = {torch.bfloat16: _lib.DT_BF16, torch.int8: _lib.DT_I8, torch.float32: _lib.DT_F32}
_workspaces = {}
MAX_INDEX = 0xFFFFFFFF          # packed keys carry a 32-bit index field: global row indices must be below this


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _on(t: torch.Tensor):
    """Context that makes t's device the current HIP device for the duration of a library call: the C ABI
    enqueues on the stream it is handed and sets per-device kernel attributes for the CURRENT device, so a
    tensor on cuda:1 must not be launched while cuda:0 is current."""
    return torch.cuda.device(t.device)


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AbcError("abcsearch kernels run on the GPU only; got a CPU tensor (no CPU fallback exists)")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream) (caller-side allocation, as the ABI requires).  Work enqueued on
    one stream runs in order, so calls that share a stream may share the candidate lists; two streams (or two
    threads, each on its own stream) get two buffers.  Two threads launching on the SAME stream still race on the
    host side of this cache: give each thread its own stream."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, index, torch.cuda.current_stream(torch.device("cuda", index)).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _workspaces.pop(key, None)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces(device: Optional[torch.device] = None) -> int:
    """Drop the cached scratch buffers (of one device, or all) and return the bytes released.  The cache holds one
    buffer per (device, stream) that ever ran a search -- up to ~540 MB each at 8192-query launches -- and never shrinks
    on its own: call this when a stream is retired, or between phases of a job that used many streams.  Work already
    enqueued keeps its buffer alive (the caching allocator recycles it only after the stream has passed it)."""
    freed = 0
    for key in list(_workspaces):
        if device is None or (key[0], key[1]) == (device.type, device.index if device.index is not None else key[1]):
            freed += _workspaces.pop(key).numel()
    return freed


def topk(q: torch.Tensor, c: torch.Tensor, k: int, upper_bound: Optional[torch.Tensor] = None,
         idx_offset: int = 0, rank_ref: Optional[torch.Tensor] = None, rank_self_offset: Optional[int] = None,
         q_scale: Optional[torch.Tensor] = None, c_scale: Optional[torch.Tensor] = None
         ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Exact per-row top-k of q @ c.T on the matrix cores; returns (values f32 [nq,k], indices i64
    [nq,k], rank counts i64 [nq] or None).  dtype bf16 / f32 / int8 (with per-vector scales)."""
    _need_gpu(q, c, upper_bound, rank_ref, q_scale, c_scale)
    if q.dtype != c.dtype or q.dtype not in _DT:
        raise _lib.AbcError(f"unsupported dtypes {q.dtype}/{c.dtype}")
    if q.dim() != 2 or c.dim() != 2 or q.shape[1] != c.shape[1]:
        raise _lib.AbcError(f"shape mismatch {tuple(q.shape)} vs {tuple(c.shape)}")
    lib = _lib.load()
    q, c = q.contiguous(), c.contiguous()
    nq, d = q.shape
    nc = c.shape[0]
    dt = _DT[q.dtype]
    out_val = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    out_idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    out_rank = None
    if upper_bound is not None:
        upper_bound = upper_bound.to(torch.float32).contiguous()
        assert upper_bound.numel() == nq
    if rank_ref is not None:
        rank_ref = rank_ref.to(torch.float32).contiguous()
        assert rank_ref.numel() == nq
        out_rank = torch.empty((nq,), dtype=torch.int64, device=q.device)
    if c.device != q.device:
        raise _lib.AbcError(f"query on {q.device} but corpus on {c.device}")
    need = lib.abc_topk_workspace_bytes(nq, nc, d, k, dt)
    if need == 0:
        _lib.check(-1, "abc_topk_workspace_bytes")
    self_off = _lib.NO_SELF if rank_self_offset is None else int(rank_self_offset)
    with _on(q):
        ws = workspace(need, q.device)
        st = _stream(q)
        if dt == _lib.DT_I8:
            if q_scale is None or c_scale is None:
                raise _lib.AbcError("int8 search needs q_scale and c_scale")
            q_scale, c_scale = q_scale.to(torch.float32).contiguous(), c_scale.to(torch.float32).contiguous()
            rc = lib.abc_topk_i8(q.data_ptr(), q_scale.data_ptr(), nq, c.data_ptr(), c_scale.data_ptr(), nc, d, k,
                                 _ptr(upper_bound), idx_offset, out_val.data_ptr(), out_idx.data_ptr(),
                                 _ptr(rank_ref), self_off, _ptr(out_rank), ws.data_ptr(), ws.numel(), st)
        else:
            fn = lib.abc_topk_bf16 if dt == _lib.DT_BF16 else lib.abc_topk_f32
            rc = fn(q.data_ptr(), nq, c.data_ptr(), nc, d, k, _ptr(upper_bound), idx_offset, out_val.data_ptr(),
                    out_idx.data_ptr(), _ptr(rank_ref), self_off, _ptr(out_rank), ws.data_ptr(), ws.numel(), st)
    _lib.check(rc, "abc_topk")
    return out_val, out_idx, out_rank


def topk_packed(q: torch.Tensor, c: torch.Tensor, k: int, upper_bound: Optional[torch.Tensor] = None, idx_offset: int = 0,
                q_scale: Optional[torch.Tensor] = None, c_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same search with packed output: int64 [nq, k] holding the library's 64-bit keys (orderable score << 32 |
    0xFFFFFFFF - global index, 0 = no entry), the form a shard's lists cross the GPUs in (ONE tensor of 8 B per entry)."""
    _need_gpu(q, c, upper_bound, q_scale, c_scale)
    if q.dtype != c.dtype or q.dtype not in _DT:
        raise _lib.AbcError(f"unsupported dtypes {q.dtype}/{c.dtype}")
    if q.dim() != 2 or c.dim() != 2 or q.shape[1] != c.shape[1] or c.device != q.device:
        raise _lib.AbcError(f"shape/device mismatch {tuple(q.shape)} on {q.device} vs {tuple(c.shape)} on {c.device}")
    lib = _lib.load()
    q, c = q.contiguous(), c.contiguous()
    nq, d = q.shape
    nc = c.shape[0]
    dt = _DT[q.dtype]
    keys = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    if upper_bound is not None:
        upper_bound = upper_bound.to(torch.float32).contiguous()
    if dt == _lib.DT_I8:
        if q_scale is None or c_scale is None:
            raise _lib.AbcError("int8 search needs q_scale and c_scale")
        q_scale, c_scale = q_scale.to(torch.float32).contiguous(), c_scale.to(torch.float32).contiguous()
    need = lib.abc_topk_workspace_bytes(nq, nc, d, k, dt)
    if need == 0:
        _lib.check(-1, "abc_topk_workspace_bytes")
    with _on(q):
        ws = workspace(need, q.device)
        rc = lib.abc_topk_packed(dt, q.data_ptr(), _ptr(q_scale), nq, c.data_ptr(), _ptr(c_scale), nc, d, k, _ptr(upper_bound),
                                 int(idx_offset), keys.data_ptr(), ws.data_ptr(), ws.numel(), _stream(q))
    _lib.check(rc, "abc_topk_packed")
    return keys


def merge_packed(keys: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge [parts, nq, k] packed key lists (each sorted descending) into values f32 [nq, k] / indices i64 [nq, k]."""
    _need_gpu(keys)
    lib = _lib.load()
    keys = keys.contiguous()
    if keys.dtype != torch.int64 or keys.dim() != 3:
        raise _lib.AbcError("merge_packed: expected int64 [parts, nq, k]")
    parts, nq, k = keys.shape
    out_val = torch.empty((nq, k), dtype=torch.float32, device=keys.device)
    out_idx = torch.empty((nq, k), dtype=torch.int64, device=keys.device)
    with _on(keys):
        rc = lib.abc_topk_merge_packed(keys.data_ptr(), parts, nq, k, out_val.data_ptr(), out_idx.data_ptr(), _stream(keys))
    _lib.check(rc, "abc_topk_merge_packed")
    return out_val, out_idx


def rowdot(q: torch.Tensor, c: torch.Tensor, q_scale=None, c_scale=None) -> torch.Tensor:
    """out[i] = q[i] . c[i] (fp32).  The positive score of row-aligned pairs."""
    _need_gpu(q, c)
    if q.dtype != c.dtype or q.dtype not in _DT or q.shape != c.shape:
        raise _lib.AbcError("rowdot: q and c must have the same dtype and shape")
    lib = _lib.load()
    q, c = q.contiguous(), c.contiguous()
    n, d = q.shape
    out = torch.empty((n,), dtype=torch.float32, device=q.device)
    with _on(q):
        st = _stream(q)
        if q.dtype == torch.int8:
            rc = lib.abc_rowdot_i8(q.data_ptr(), q_scale.contiguous().data_ptr(), c.data_ptr(),
                                   c_scale.contiguous().data_ptr(), n, d, out.data_ptr(), st)
        elif q.dtype == torch.bfloat16:
            rc = lib.abc_rowdot_bf16(q.data_ptr(), c.data_ptr(), n, d, out.data_ptr(), st)
        else:
            rc = lib.abc_rowdot_f32(q.data_ptr(), c.data_ptr(), n, d, out.data_ptr(), st)
    _lib.check(rc, "abc_rowdot")
    return out


def merge_topk(vals: torch.Tensor, idx: torch.Tensor, idx_limit: int = MAX_INDEX) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge [parts, nq, k] sorted lists (global indices, -1 = missing) into [nq, k].  `idx_limit` is the exclusive
    upper bound of the indices (the corpus' total row count): beyond 2^32-1 the library refuses instead of wrapping."""
    _need_gpu(vals, idx)
    lib = _lib.load()
    vals, idx = vals.to(torch.float32).contiguous(), idx.to(torch.int64).contiguous()
    parts, nq, k = vals.shape
    out_val = torch.empty((nq, k), dtype=torch.float32, device=vals.device)
    out_idx = torch.empty((nq, k), dtype=torch.int64, device=vals.device)
    with _on(vals):
        rc = lib.abc_topk_merge(vals.data_ptr(), idx.data_ptr(), parts, nq, k, int(idx_limit), out_val.data_ptr(),
                                out_idx.data_ptr(), _stream(vals))
    _lib.check(rc, "abc_topk_merge")
    return out_val, out_idx


def quantize_i8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric per-vector absmax int8 quantiser (SURVEY.md 8c rule)."""
    _need_gpu(x)
    lib = _lib.load()
    x = x.to(torch.float32).contiguous()
    n, d = x.shape
    q = torch.empty((n, d), dtype=torch.int8, device=x.device)
    s = torch.empty((n,), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = lib.abc_quantize_i8(x.data_ptr(), n, d, q.data_ptr(), s.data_ptr(), _stream(x))
    _lib.check(rc, "abc_quantize_i8")
    return q, s


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round-to-nearest-even) through the library's streaming kernel."""
    _need_gpu(x)
    lib = _lib.load()
    x = x.to(torch.float32).contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with _on(x):
        rc = lib.abc_cast_bf16(x.data_ptr(), x.numel(), out.data_ptr(), _stream(x))
    _lib.check(rc, "abc_cast_bf16")
    return out


def debug_set(name: str, value: int) -> None:
    """Diagnostics hook of the library (tests, timing scripts): "flags", "plan_qb"."""
    _lib.check(_lib.load().abc_debug_set(name.encode(), int(value)), "abc_debug_set")


def debug_get(name: str) -> int:
    return int(_lib.load().abc_debug_get(name.encode()))
