// Exact-fp32 "NT" GEMM partials on v_mfma_f32_32x32x2_f32, shared by the InfoNCE logits and the embedding
// head: part[s][i][j] = sum over split s of the K range of q[i][k] * p[j][k].  64x64 tiles, K split over
// splitk workgroups (blockIdx.y) and the 8 waves of each; fixed-order tree reduction through LDS.
// (A template only so that the header can be included from several translation units.)
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Tree-reduce the per-wave accumulator tiles of an 8-wave workgroup through LDS (fixed order:
// (w, w+4), then (w, w+2), then (0, 1)).  NV = floats per lane.  Result valid in wave 0.
template <int NV>
__device__ __forceinline__ void wg_tree_reduce(float (&v)[NV], float* lds, int wave, int lane) {
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* dst = lds + (size_t)(wave - half) * NV * 64;
#pragma unroll
            for (int i = 0; i < NV; ++i) dst[i * 64 + lane] = v[i];
        }
        __syncthreads();
        if (wave < half) {
            const float* src = lds + (size_t)wave * NV * 64;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += src[i * 64 + lane];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// part[s][i][j] = sum over the K range of split s of q[i][k] * p[j][k]      (i < B, j < C)
template <int UNUSED>
__global__ __launch_bounds__(512) void nt_gemm_partials_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                           int B, int C, int d, int splitk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = (float*)smem_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (C + 63) / 64;
    const int m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
    const int s = blockIdx.y;
    const int units = d / 8;
    const int slice = s * 8 + wave, nslices = splitk * 8;
    const int u0 = (int)(((int64_t)slice * units) / nslices), u1 = (int)(((int64_t)(slice + 1) * units) / nslices);

    const float* qa[2];
    const float* pb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int r = m0 + b * 32 + l31; if (r >= B) r = B - 1;
        int c = n0 + b * 32 + l31; if (c >= C) c = C - 1;
        qa[b] = q + (size_t)r * d + 4 * h;
        pb[b] = p + (size_t)c * d + 4 * h;
    }
    v16f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll 2
    for (int u = u0; u < u1; ++u) {
        v4f a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *(const v4f*)(qa[i] + 8 * u);
            b[i] = *(const v4f*)(pb[i] + 8 * u);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    float v[64];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[(i * 2 + j) * 16 + r] = acc[i][j][r];
    wg_tree_reduce<64>(v, lds, wave, lane);
    if (wave == 0) {
        float* out = part + (size_t)s * B * C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < B && col < C) out[(size_t)row * C + col] = v[(i * 2 + j) * 16 + r];
                }
            }
    }
}

