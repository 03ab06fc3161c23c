// C ABI of libabcsearch.so + the small streaming kernels (rowdot, quantise, cast).
#include "common.h"
#include "topk_sweep.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int abc_fail_hip(hipError_t e, const char* where) { return fail(ABC_E_HIP, "%s: %s", where, hipGetErrorString(e)); }
int abc_fail_invalid(const char* msg) { return fail(ABC_E_INVALID, "%s", msg); }
int abc_fail_ws(size_t need, size_t have) { return fail(ABC_E_WORKSPACE, "workspace too small: need %zu bytes, have %zu", need, have); }

extern "C" const char* abc_version(void) { return "abcsearch 0.1.0 (gfx950)"; }
extern "C" const char* abc_last_error(void) { return g_err; }

static int elem_size(int dt) { return dt == ABC_DT_BF16 ? 2 : (dt == ABC_DT_I8 ? 1 : (dt == ABC_DT_F32 ? 4 : 0)); }

static int check_common(int64_t nq, int64_t nc, int d, int k, int dt) {
    if (elem_size(dt) == 0) return fail(ABC_E_INVALID, "unknown dtype %d", dt);
    if (nq < 0 || nc < 0) return fail(ABC_E_INVALID, "negative row count");
    if (d <= 0 || ((int64_t)d * elem_size(dt)) % 128 != 0)
        return fail(ABC_E_INVALID, "d=%d: d*sizeof(elem) must be a positive multiple of 128 bytes", d);
    if (k < 1 || k > ABC_MAX_K) return fail(ABC_E_INVALID, "k=%d out of range [1,%d]", k, ABC_MAX_K);
    if (nc >= 0xFFFFFFFFll) return fail(ABC_E_INVALID, "nc=%lld exceeds 2^32-2 rows per call", (long long)nc);
    return ABC_OK;
}

extern "C" size_t abc_topk_workspace_bytes(int64_t nq, int64_t nc, int d, int k, int dtype) {
    if (check_common(nq, nc, d, k, dtype) != ABC_OK) return 0;
    if (nq == 0 || nc == 0) return 16;
    return sweep_workspace_bytes(nq, nc, k);
}

// fill [nq,k] outputs with (-inf, -1): the answer for an empty corpus
__global__ void fill_empty_kernel(float* v, int64_t* ix, int64_t n, int64_t* rank, int64_t nq) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { v[i] = -INFINITY; ix[i] = -1; }
    if (rank && i < nq) rank[i] = 0;
}

__global__ void fill_zero_keys_kernel(uint64_t* keys, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = 0ull;
}

static int topk_entry(int dt, const void* q, const float* qs, int64_t nq, const void* c, const float* cs, int64_t nc,
                      int d, int k, const float* ub, int64_t idx_offset, float* out_val, int64_t* out_idx, uint64_t* out_keys,
                      const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(nq, nc, d, k, dt);
    if (rc) return rc;
    if (nq == 0) return ABC_OK;
    if (out_keys) {
        if (idx_offset < 0 || idx_offset + nc > 0xFFFFFFFFll)
            return fail(ABC_E_INVALID, "packed keys: idx_offset + nc = %lld exceeds the 32-bit index field (2^32-1)", (long long)(idx_offset + nc));
    } else if (!out_val || !out_idx) return fail(ABC_E_INVALID, "null output pointer");
    if ((rank_ref == nullptr) != (out_rank == nullptr)) return fail(ABC_E_INVALID, "rank_ref and out_rank must be given together");
    hipStream_t st = (hipStream_t)stream;
    if (nc == 0 && out_keys) {
        const int64_t n = nq * k;
        hipLaunchKernelGGL(fill_zero_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_keys, n);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "fill_zero_keys");
    }
    if (nc == 0) {
        const int64_t n = nq * k;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_val, out_idx, n, out_rank, nq);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "fill_empty");
    }
    if (!q || !c) return fail(ABC_E_INVALID, "null input pointer");
    if (((uintptr_t)q | (uintptr_t)c) & 15) return fail(ABC_E_INVALID, "q and c must be 16-byte aligned");
    if (dt == ABC_DT_I8 && (!qs || !cs)) return fail(ABC_E_INVALID, "int8 needs q_scale and c_scale");
    const size_t need = sweep_workspace_bytes(nq, nc, k);
    if (!ws || ws_bytes < need) return abc_fail_ws(need, ws_bytes);
    if ((uintptr_t)ws & 15) return fail(ABC_E_INVALID, "workspace must be 16-byte aligned");
    hipError_t he = hipSuccess;
    rc = sweep_run(dt, q, qs, nq, c, cs, nc, d, k, ub, idx_offset, out_val, out_idx, out_keys, rank_ref, rank_self_offset, out_rank, ws, st, &he);
    if (rc == ABC_E_HIP) return abc_fail_hip(he, "topk sweep launch");
    return rc;
}

extern "C" int abc_topk_bf16(const void* q, int64_t nq, const void* c, int64_t nc, int d, int k, const float* ub,
                             int64_t idx_offset, float* out_val, int64_t* out_idx, const float* rank_ref,
                             int64_t rank_self_offset, int64_t* out_rank, void* ws, size_t ws_bytes, void* stream) {
    return topk_entry(ABC_DT_BF16, q, nullptr, nq, c, nullptr, nc, d, k, ub, idx_offset, out_val, out_idx, nullptr, rank_ref, rank_self_offset, out_rank, ws, ws_bytes, stream);
}
extern "C" int abc_topk_f32(const void* q, int64_t nq, const void* c, int64_t nc, int d, int k, const float* ub,
                            int64_t idx_offset, float* out_val, int64_t* out_idx, const float* rank_ref,
                             int64_t rank_self_offset, int64_t* out_rank, void* ws, size_t ws_bytes, void* stream) {
    return topk_entry(ABC_DT_F32, q, nullptr, nq, c, nullptr, nc, d, k, ub, idx_offset, out_val, out_idx, nullptr, rank_ref, rank_self_offset, out_rank, ws, ws_bytes, stream);
}
extern "C" int abc_topk_i8(const void* q, const float* qs, int64_t nq, const void* c, const float* cs, int64_t nc, int d,
                           int k, const float* ub, int64_t idx_offset, float* out_val, int64_t* out_idx,
                           const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank, void* ws, size_t ws_bytes,
                           void* stream) {
    return topk_entry(ABC_DT_I8, q, qs, nq, c, cs, nc, d, k, ub, idx_offset, out_val, out_idx, nullptr, rank_ref, rank_self_offset, out_rank, ws, ws_bytes, stream);
}

// Packed output for the cross-GPU exchange: out_keys[nq,k] = orderable(score) << 32 | (0xFFFFFFFF - global index), 0 = no entry.
extern "C" int abc_topk_packed(int dtype, const void* q, const float* qs, int64_t nq, const void* c, const float* cs, int64_t nc,
                               int d, int k, const float* ub, int64_t idx_offset, uint64_t* out_keys, void* ws, size_t ws_bytes,
                               void* stream) {
    if (!out_keys && nq > 0) return fail(ABC_E_INVALID, "null output pointer");
    return topk_entry(dtype, q, qs, nq, c, cs, nc, d, k, ub, idx_offset, nullptr, nullptr, out_keys, nullptr, INT64_MIN, nullptr, ws,
                      ws_bytes, stream);
}

extern "C" int abc_topk_merge_packed(const uint64_t* keys, int parts, int64_t nq, int k, float* out_val, int64_t* out_idx,
                                     void* stream) {
    if (parts < 1 || nq < 0 || k < 1 || k > ABC_MAX_K) return fail(ABC_E_INVALID, "bad merge shape parts=%d nq=%lld k=%d", parts, (long long)nq, k);
    if (nq == 0) return ABC_OK;
    if (!keys || !out_val || !out_idx) return fail(ABC_E_INVALID, "null pointer");
    hipError_t he = hipSuccess;
    int rc = merge_keys_run(keys, parts, nq, k, out_val, out_idx, (hipStream_t)stream, &he);
    if (rc == ABC_E_HIP) return abc_fail_hip(he, "merge launch");
    return rc;
}

extern "C" int abc_topk_merge(const float* vals, const int64_t* idx, int parts, int64_t nq, int k, int64_t idx_limit,
                              float* out_val, int64_t* out_idx, void* stream) {
    if (parts < 1 || nq < 0 || k < 1 || k > ABC_MAX_K) return fail(ABC_E_INVALID, "bad merge shape parts=%d nq=%lld k=%d", parts, (long long)nq, k);
    // the merge packs (score, index) into one 64-bit key with a 32-bit index field (0xFFFFFFFF - index, key 0 = hole)
    if (idx_limit < 0 || idx_limit > 0xFFFFFFFFll)
        return fail(ABC_E_INVALID, "merge: idx_limit=%lld: indices must lie in [0, 2^32-1) (32-bit index field of the packed keys)",
                    (long long)idx_limit);
    if (nq == 0) return ABC_OK;
    if (!vals || !idx || !out_val || !out_idx) return fail(ABC_E_INVALID, "null pointer");
    hipError_t he = hipSuccess;
    int rc = merge_lists_run(vals, idx, parts, nq, k, out_val, out_idx, (hipStream_t)stream, &he);
    if (rc == ABC_E_HIP) return abc_fail_hip(he, "merge launch");
    return rc;
}

// Diagnostics hook (tests, timing scripts): "flags", "plan_qb"; see topk_sweep.h.
extern "C" int abc_debug_set(const char* name, int64_t value) {
    const int rc = sweep_debug_set(name, (long)value);
    return rc == ABC_OK ? ABC_OK : fail(ABC_E_INVALID, "abc_debug_set: unknown knob");
}
extern "C" int64_t abc_debug_get(const char* name) { return (int64_t)sweep_debug_get(name); }

// ------------------------------------------------------------------------------------------------
// rowdot for shapes the sweep does not take (row bytes not a multiple of 128): one wave per row, 16-byte loads,
// fp32 accumulate, fixed reduction order.  Every other shape goes through rowdot_run (the sweep's MFMA arithmetic).
template <int DT>
__global__ __launch_bounds__(256) void rowdot_kernel(const char* __restrict__ q, const char* __restrict__ c,
                                                     const float* __restrict__ qs, const float* __restrict__ cs,
                                                     int64_t n, int rowbytes, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const v4i* a = (const v4i*)(q + row * rowbytes);
    const v4i* b = (const v4i*)(c + row * rowbytes);
    float accf = 0.f;
    int acci = 0;
    for (int i = lane; i < rowbytes / 16; i += 64) {
        const v4i x = a[i], y = b[i];
        if constexpr (DT == ABC_DT_BF16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t u = (uint32_t)x[j], w = (uint32_t)y[j];
                accf = fmaf(__uint_as_float(u << 16), __uint_as_float(w << 16), accf);
                accf = fmaf(__uint_as_float(u & 0xFFFF0000u), __uint_as_float(w & 0xFFFF0000u), accf);
            }
        } else if constexpr (DT == ABC_DT_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) accf = fmaf(__int_as_float(x[j]), __int_as_float(y[j]), accf);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int b8 = 0; b8 < 4; ++b8) acci += (int)(int8_t)(x[j] >> (8 * b8)) * (int)(int8_t)(y[j] >> (8 * b8));
            }
        }
    }
    if constexpr (DT == ABC_DT_I8) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) acci += __shfl_xor(acci, m, 64);
        if (lane == 0) out[row] = (float)acci * (qs[row] * cs[row]);
    } else {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) accf += __shfl_xor(accf, m, 64);
        if (lane == 0) out[row] = accf;
    }
}

static int rowdot_entry(int dt, const void* q, const float* qs, const void* c, const float* cs, int64_t n, int d,
                        float* out, void* stream) {
    if (elem_size(dt) == 0 || n < 0 || d <= 0 || ((int64_t)d * elem_size(dt)) % 16 != 0)
        return fail(ABC_E_INVALID, "rowdot: bad shape n=%lld d=%d", (long long)n, d);
    if (n == 0) return ABC_OK;
    if (!q || !c || !out) return fail(ABC_E_INVALID, "null pointer");
    if (((uintptr_t)q | (uintptr_t)c) & 15) return fail(ABC_E_INVALID, "q and c must be 16-byte aligned");
    const int rowbytes = d * elem_size(dt);
    const unsigned grid = (unsigned)((n + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (rowbytes % 128 == 0) {
        // every shape the sweep accepts: the sweep's own MFMA arithmetic, bit-identical to the tile's score (i, i)
        hipError_t he = hipSuccess;
        const int rc = rowdot_run(dt, q, qs, c, cs, n, d, out, st, &he);
        return rc == ABC_E_HIP ? abc_fail_hip(he, "rowdot launch") : rc;
    }
    if (dt == ABC_DT_BF16) hipLaunchKernelGGL(rowdot_kernel<ABC_DT_BF16>, dim3(grid), dim3(256), 0, st, (const char*)q, (const char*)c, qs, cs, n, rowbytes, out);
    else if (dt == ABC_DT_F32) hipLaunchKernelGGL(rowdot_kernel<ABC_DT_F32>, dim3(grid), dim3(256), 0, st, (const char*)q, (const char*)c, qs, cs, n, rowbytes, out);
    else hipLaunchKernelGGL(rowdot_kernel<ABC_DT_I8>, dim3(grid), dim3(256), 0, st, (const char*)q, (const char*)c, qs, cs, n, rowbytes, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "rowdot launch");
}
extern "C" int abc_rowdot_bf16(const void* q, const void* c, int64_t n, int d, float* out, void* stream) {
    return rowdot_entry(ABC_DT_BF16, q, nullptr, c, nullptr, n, d, out, stream);
}
extern "C" int abc_rowdot_f32(const void* q, const void* c, int64_t n, int d, float* out, void* stream) {
    return rowdot_entry(ABC_DT_F32, q, nullptr, c, nullptr, n, d, out, stream);
}
extern "C" int abc_rowdot_i8(const void* q, const float* qs, const void* c, const float* cs, int64_t n, int d, float* out,
                             void* stream) {
    if (!qs || !cs) return fail(ABC_E_INVALID, "int8 needs scales");
    return rowdot_entry(ABC_DT_I8, q, qs, c, cs, n, d, out, stream);
}

// ------------------------------------------------------------------------------------------------
// int8 quantiser: one wave per row.  Rows of up to 8192 floats whose length is a multiple of 4 are read ONCE with 16-byte
// loads and stay in registers between the absmax and the quantisation; four quantised values leave in one 4-byte store
// (round 3: 3.65 -> 5.5 TB/s, profiles/r3_hbm_kernels.log; the arithmetic -- IEEE division, rintf, clamp -- is unchanged).
// Other rows take two passes of scalar accesses.
__device__ __forceinline__ float quant1(float x, float scale) {
    float r = rintf(__fdiv_rn(x, scale));             // round-half-even, IEEE division as torch
    return fminf(fmaxf(r, -127.f), 127.f);
}
template <bool VEC>
__global__ __launch_bounds__(256) void quantize_i8_kernel(const float* __restrict__ x, int64_t n, int d,
                                                          int8_t* __restrict__ oq, float* __restrict__ os) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + row * d;
    float amax = 0.f;
    if constexpr (VEC) {
        typedef __attribute__((ext_vector_type(4))) float v4f;
        const v4f* xr4 = (const v4f*)xr;
        const int nv = d >> 2;                                // <= 2048: at most 32 pieces per lane
        v4f buf[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int i = lane + 64 * j;
            buf[j] = i < nv ? xr4[i] : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) amax = fmaxf(amax, fabsf(buf[j][t]));
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
        const float scale = amax > 0.f ? __fdiv_rn(amax, 127.0f) : 1.0f;
        if (lane == 0) os[row] = scale;
        uint32_t* o4 = (uint32_t*)(oq + row * d);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                uint32_t w = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) w |= ((uint32_t)(uint8_t)(int8_t)quant1(buf[j][t], scale)) << (8 * t);
                o4[i] = w;
            }
        }
    } else {
        for (int i = lane; i < d; i += 64) amax = fmaxf(amax, fabsf(xr[i]));
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
        const float scale = amax > 0.f ? __fdiv_rn(amax, 127.0f) : 1.0f;
        if (lane == 0) os[row] = scale;
        for (int i = lane; i < d; i += 64) oq[row * d + i] = (int8_t)quant1(xr[i], scale);
    }
}
extern "C" int abc_quantize_i8(const float* x, int64_t n, int d, int8_t* out_q, float* out_scale, void* stream) {
    if (n < 0 || d <= 0) return fail(ABC_E_INVALID, "quantize: bad shape");
    if (n == 0) return ABC_OK;
    if (!x || !out_q || !out_scale) return fail(ABC_E_INVALID, "null pointer");
    const bool vec = (d & 3) == 0 && d <= 8192 && (((uintptr_t)x | (uintptr_t)out_q) & 15) == 0;
    if (vec) hipLaunchKernelGGL(quantize_i8_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, d, out_q, out_scale);
    else hipLaunchKernelGGL(quantize_i8_kernel<false>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, d, out_q, out_scale);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "quantize launch");
}

// fp32 -> bf16, round to nearest even (a plain grid-stride loop: 5.1 TB/s of read + write; 16-byte loads change nothing)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, int64_t n, uint16_t* __restrict__ o) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        o[i] = (uint16_t)f32_to_bf16_bits(x[i]);
    }
}
extern "C" int abc_cast_bf16(const float* x, int64_t n, void* out_bf16, void* stream) {
    if (n < 0) return fail(ABC_E_INVALID, "cast: negative n");
    if (n == 0) return ABC_OK;
    if (!x || !out_bf16) return fail(ABC_E_INVALID, "null pointer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, (uint16_t*)out_bf16);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "cast launch");
}

// ------------------------------------------------------------------------------------------------
// negatives.json text (host code, no GPU): the bytes json.dump(negative_dict, f) writes for rows first_row ..
// first_row + n_rows - 1 of the {int: [int, ...]} dict of pretrain/create_negative_index.py:34-43 -- keys as
// strings, ", " between entries and between items, no braces (the caller writes "{" / "}" and the ", " between blocks).
// Entries < 0 (no candidate) are dropped.  Returns the byte count, or -(bytes needed) when `cap` is too small.
static inline char* put_i64(char* p, int64_t v) {
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

extern "C" int64_t abc_format_negatives(const int64_t* idx, int64_t n_rows, int k, int64_t first_row, char* out, int64_t cap) {
    if (!idx || n_rows < 0 || k < 0) { fail(ABC_E_INVALID, "abc_format_negatives: bad arguments"); return 0; }
    // worst case per row: '"' + 20 digits + '": [' + k * (20 digits + ", ") + "]" + ", "
    const int64_t worst = n_rows * (28 + (int64_t)k * 22 + 3);
    if (!out || cap < worst) return -worst;
    char* p = out;
    for (int64_t r = 0; r < n_rows; ++r) {
        if (r) { *p++ = ','; *p++ = ' '; }
        *p++ = '"';
        p = put_i64(p, first_row + r);
        *p++ = '"'; *p++ = ':'; *p++ = ' '; *p++ = '[';
        bool first = true;
        const int64_t* row = idx + r * k;
        for (int j = 0; j < k; ++j) {
            if (row[j] < 0) continue;
            if (!first) { *p++ = ','; *p++ = ' '; }
            p = put_i64(p, row[j]);
            first = false;
        }
        *p++ = ']';
    }
    return (int64_t)(p - out);
}
