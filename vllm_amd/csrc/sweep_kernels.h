#pragma once
// Fused score-GEMM + running top-k sweep for gfx950 (MI355X): shared device helpers + the kernel (sweep16.inc).
//
// One workgroup (4 waves, one per SIMD) owns a block of 256 queries and a contiguous range of 256-row corpus tiles (a
// "split").  For each tile it contracts the 256 x 256 score tile on the matrix cores (corpus rows on the MFMA row axis,
// queries on the lane/column axis) and then filters the accumulators, still in registers, against each query's running
// threshold.  Survivors are appended to that query's candidate list in the workspace; a full list is compacted (wave-wide
// bitonic sort, keep the best k, raise the threshold).  The score matrix never exists in memory.
#include "common.h"
#include "topk_sweep.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

#define TILE 256

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, uint32_t bytes) {
    // wave-uniform by construction (kernel arguments and blockIdx only); readfirstlane makes it provable
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// ------------------------------------------------------------------------------------------------
// Compaction of one query's candidate list: sort (desc), keep the best k, return the new threshold and
// the value at rank pub_m-1 (the bound this list publishes).  Called by the whole wave (uniform);
// `buf` is wave-uniform.  E = keys per lane, chosen by the caller from the list length (n <= 64*E).
template <int E>
__device__ __forceinline__ void compact_list_e(uint64_t* buf, int n, int k, int pub_m, int lane, float& new_thr,
                                               int& new_cnt, float& pub_val, bool finalize) {
    uint64_t key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        key[e] = (i < n) ? buf[i] : 0ull;
    }
    wave_sort_desc<E>(key, lane);
    const int keep = n < k ? n : k;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        if (i < keep) buf[i] = key[e];
        else if (finalize && i < k) buf[i] = 0ull;
    }
    if (finalize && E == 1 && k > 64 && lane + 64 < k) buf[lane + 64] = 0ull;    // pad the part of [keep,k) beyond this E
    // value at rank k-1 becomes the threshold once the list is full
    uint64_t kth = 0;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e == ((k - 1) >> 6)) kth = key[e];
    kth = shfl_u64(kth, (k - 1) & 63);
    new_cnt = keep;
    new_thr = (n >= k) ? key_val(kth) : -INFINITY;
    // every list of the query publishes its value at rank pub_m-1 (selected like kth: pub_m > 64 happens with one list
    // per query and k > 64); with L lists and L*pub_m >= k the minimum over the lists is a lower bound of the query's
    // global k-th best
    uint64_t mth = 0;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e == ((pub_m - 1) >> 6)) mth = key[e];
    mth = shfl_u64(mth, (pub_m - 1) & 63);
    pub_val = (n >= pub_m && pub_m <= 64 * E) ? key_val(mth) : -INFINITY;
}

#include "sweep16.inc"
