// Shared device helpers for libabcsearch (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) int v4i;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(8))) short v8s;    // 8 x bf16
typedef __attribute__((ext_vector_type(16))) float v16f;
typedef __attribute__((ext_vector_type(16))) int v16i;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// f32 -> bf16 bits, round-to-nearest-even, NaN stays NaN
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// ---- ordered 64-bit keys: larger key == better (larger score; on ties smaller index) ----------
// key = orderable(score) << 32 | (0xFFFFFFFF - index);  key 0 is reserved for "no entry".
__device__ __forceinline__ uint32_t f32_orderable(float v) {
    uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float v, uint32_t idx) {
    return ((uint64_t)f32_orderable(v) << 32) | (uint64_t)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ float key_val(uint64_t k) { return orderable_f32((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_idx(uint64_t k) { return 0xFFFFFFFFu - (uint32_t)k; }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t x, int m) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t x, int src) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// One compare-exchange of a bitonic network on element i (this lane's copy `a`) against its
// partner's value `b`: keep the larger when `keep_max`.
__device__ __forceinline__ uint64_t cex(uint64_t a, uint64_t b, bool keep_max) {
    const bool a_gt = a > b;
    return (a_gt == keep_max) ? a : b;
}

// Wave-wide bitonic sort, DESCENDING, of 64*E keys; element i = e*64 + lane lives in key[e] of
// `lane`.  E is a power of two.
template <int E>
__device__ __forceinline__ void wave_sort_desc(uint64_t (&key)[E], int lane) {
    constexpr int N = 64 * E;
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int es = stride >> 6;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & es) == 0) {
                        const int i = e * 64;              // lane bits do not matter for (i & size) when size > 64
                        const bool desc = ((i & size) == 0) || size == N;
                        const uint64_t a = key[e], b = key[e | es];
                        const bool a_gt = a > b;
                        // lower index keeps max when descending
                        key[e] = (a_gt == desc) ? a : b;
                        key[e | es] = (a_gt == desc) ? b : a;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = e * 64 + lane;
                    const bool desc = ((i & size) == 0) || size == N;
                    const bool lower = (lane & stride) == 0;
                    const uint64_t other = shfl_xor_u64(key[e], stride);
                    key[e] = cex(key[e], other, lower == desc);
                }
            }
        }
    }
}

// Sort a BITONIC sequence of 64*E keys into descending order (the tail of a bitonic merge).
template <int E>
__device__ __forceinline__ void wave_bitonic_merge_desc(uint64_t (&key)[E], int lane) {
#pragma unroll
    for (int stride = 32 * E; stride > 0; stride >>= 1) {
        if (stride >= 64) {
            const int es = stride >> 6;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((e & es) == 0) {
                    const uint64_t a = key[e], b = key[e | es];
                    const bool a_gt = a > b;
                    key[e] = a_gt ? a : b;
                    key[e | es] = a_gt ? b : a;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool lower = (lane & stride) == 0;
                const uint64_t other = shfl_xor_u64(key[e], stride);
                key[e] = cex(key[e], other, lower);
            }
        }
    }
}
