"""ctypes binding of libabcsearch.so (the C ABI declared in include/abcsearch.h).

The product path has no fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABC_LIB") or os.path.join(_HERE, "libabcsearch.so")   # ABC_LIB: A/B experiments only

DT_BF16, DT_I8, DT_F32 = 0, 1, 2
MAX_K = 128
NO_SELF = -(1 << 63)

_vp, _i64, _int, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); kept in one table so tests can check every declared symbol exports
SIGNATURES = {
    "abc_version": (C.c_char_p, []),
    "abc_last_error": (C.c_char_p, []),
    "abc_topk_workspace_bytes": (_sz, [_i64, _i64, _int, _int, _int]),
    "abc_topk_bf16": (_int, [_vp, _i64, _vp, _i64, _int, _int, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "abc_topk_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "abc_topk_i8": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _int, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "abc_rowdot_bf16": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "abc_rowdot_f32": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "abc_rowdot_i8": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "abc_topk_merge": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp, _vp, _vp]),
    "abc_topk_packed": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _i64, _int, _int, _vp, _i64, _vp, _vp, _sz, _vp]),
    "abc_topk_merge_packed": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp]),
    "abc_debug_set": (_int, [C.c_char_p, _i64]),
    "abc_format_negatives": (_i64, [_vp, _i64, _int, _i64, _vp, _i64]),
    "abc_debug_get": (_i64, [C.c_char_p]),
    "abc_quantize_i8": (_int, [_vp, _i64, _int, _vp, _vp, _vp]),
    "abc_cast_bf16": (_int, [_vp, _i64, _vp, _vp]),
    "abc_infonce_workspace_bytes": (_sz, [_int, _int, _int]),
    "abc_infonce_fwd": (_int, [_vp, _vp, _vp, _f32, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "abc_infonce_noalloc_max_bytes": (_sz, []),
    "abc_infonce_fwd_noalloc": (_int, [_vp, _vp, _vp, _f32, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "abc_mean_pool_bf16": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int, _int, _vp, _vp]),
    "abc_mean_pool_f32": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int, _vp, _vp]),
    "abc_embed_head_workspace_bytes": (_sz, [_int, _int, _int]),
    "abc_embed_head_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "abc_embed_head_bwd_workspace_bytes": (_sz, [_int, _int, _int]),
    "abc_embed_head_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "abc_mean_pool_bwd_bf16": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int, _vp, _vp]),
    "abc_mean_pool_bwd_f32": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int, _vp, _vp]),
    "abc_infonce_bwd": (_int, [_vp, _vp, _vp, _vp, _f32, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


class AbcError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library (once).  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AbcError(f"{LIB_PATH} is missing: build it with `make -C vllm_amd/csrc` "
                           "(or __graft_entry__.build()); there is no fallback path")
        # torch ships its own libamdhip64.so; import it first so that this library binds to the SAME
        # HIP runtime (two runtimes in one process do not share devices, streams or allocations)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().abc_last_error().decode("utf-8", "replace")
        raise AbcError(f"{what} failed (code {rc}): {msg}")
