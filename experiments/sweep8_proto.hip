// EXPERIMENT, not part of libabcsearch (round 3): sweep4c_proto.hip's loop with EIGHT waves, two per SIMD, each holding half
// the accumulators (128 corpus rows x 64 queries: 8 x 4 tiles of 16 x 16 = 128 AGPRs, 256 registers per wave), to see
// whether a second wave per SIMD that issues MFMAs while the first is busy issuing LDS-DMA pieces and fragment reads
// gets past the ~1.5 PFLOP/s where every one-wave-per-SIMD 256 x 256 loop ends.
// Derived from: sweep4b_proto.hip's loop (4 waves, 128-byte phases, two 64 KB slots, one barrier
// per phase) on v_mfma_f32_16x16x32_bf16 instead of 32x32x16.  Same LDS image (chunk ^ ((row >> 1) & 7)), same staging.
// A k-step is 32 bf16 (64 B): two per phase, 64 MFMAs of 16 clocks each.  Wave tile MB x NB tiles of 16 x 16:
//   -DMB=8 -DNB=8   128 corpus rows x 128 queries per wave (wm = wave >> 1, wn = wave & 1), 16 fragment reads per k-step
//   -DMB=16 -DNB=4  256 corpus rows x  64 queries per wave (wn = wave),                    20 fragment reads per k-step
// The A fragments are single-buffered and refilled half by half behind the MFMAs that used them, the B fragments are
// double-buffered per k-step.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 experiments/sweep4c_proto.hip -o experiments/sweep4c_proto
#include "../vllm_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define MB 8
#define NB 4
#define NW 8
#define TILE 256
#define ROWB 128
#define STAGE_BYTES (TILE * ROWB)
#define SLOT_BYTES (2 * STAGE_BYTES)
#define RING_BYTES (2 * SLOT_BYTES)
typedef __attribute__((ext_vector_type(4))) float v4f_;

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sweep8_kernel(const char* __restrict__ q, const char* __restrict__ c, int nq, int nc, int rowbytes, int kbytes, int n_qblocks,
                    int n_splits, float* __restrict__ out, unsigned long long* __restrict__ stats, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint64_t t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // corpus half, query quarter
    const int l15 = lane & 15, g4 = lane >> 4;

    int qb, split;
    {
        const int id = blockIdx.x;
        const int g = 2;
        const int per = (n_qblocks >> 3) * g;
        const int x = id & 7, j = id >> 3;
        qb = (x / g) * per + (j % per);
        split = (x % g) * (n_splits / g) + j / per;
    }
    const int n_tiles = nc / TILE;
    const int t_begin = (int)(((int64_t)split * n_tiles) / n_splits);
    const int t_end = (int)(((int64_t)(split + 1) * n_tiles) / n_splits);
    const int nph = kbytes / ROWB;

    int voff2[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        voff2[par] = (wave * 32 + (lane >> 3)) * rowbytes + (((lane & 7) ^ ((par * 4 + (lane >> 4)) & 7)) << 4);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (int64_t)qb * TILE * rowbytes, (flags & 64) ? 0u : (uint32_t)(TILE * rowbytes));
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(c + (int64_t)t_begin * TILE * rowbytes,
                                                (flags & 64) ? 0u : (uint32_t)((int64_t)(t_end - t_begin) * TILE * rowbytes));
    const int piece = 8 * rowbytes;

    // fragment of k-step s (0, 1), tile row t*16 + l15: 16-byte chunk 4s + g4, swizzled with (row >> 1) & 7
    int xa[2], xb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int x = l15 * ROWB + (((4 * s + g4) ^ ((l15 >> 1) & 7)) << 4);
        xa[s] = wm * (MB * 16) * ROWB + x;
        xb[s] = STAGE_BYTES + wn * (NB * 16) * ROWB + x;
    }

    v4f_ acc[MB][NB];
    float best[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) best[nb] = -INFINITY;

    int is_k = 0, is_cbase = 0, is_cnt = 0;
#ifndef PF_AT
#define PF_AT 10
#endif
#ifndef PF_DIST
#define PF_DIST 1
#endif
    int pf_sink = 0, pf_k = 0, pf_cb = 0;
    const int pf_voff = (wave * 64 + lane) * rowbytes;
#define ISSUE_HALF(HALF)                                                                                         \
    {                                                                                                            \
        char* dst_ = smem + (is_cnt & 1) * SLOT_BYTES + (HALF) * STAGE_BYTES + wave * 4096;                      \
        const int so_ = ((HALF) ? 0 : is_cbase) + is_k * ROWB;                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds((HALF) ? rq : rc, LDS_PTR(dst_ + i * 1024), 16, voff2[i & 1], so_ + i * piece, 0, 0); \
    }
#define ISSUE_ADVANCE()                                                                                          \
    {                                                                                                            \
        ++is_cnt;                                                                                                \
        if (++is_k == nph) {                                                                                     \
            is_k = 0;                                                                                            \
            is_cbase += TILE * rowbytes;                                                                         \
        }                                                                                                        \
    }
#define LDF(P) (*(const v4i*)(P))
#define READ_A(SB, S, LO, HI) _Pragma("unroll") for (int mb = (LO); mb < (HI); ++mb) a[mb] = LDF((SB) + xa[S] + mb * 16 * ROWB);
#define READ_B(F, SB, S, LO, HI) _Pragma("unroll") for (int nb = (LO); nb < (HI); ++nb) F[nb] = LDF((SB) + xb[S] + nb * 16 * ROWB);
    // accumulators in place in the AGPRs: left to the builtin, hipcc unties two thirds of the 64 accumulator tuples and
    // shuffles them through v_accvgpr_mov / read / write (2204 clocks per 64-byte phase instead of 1024)
#define MFMA16(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
    // a half-step = 32 MFMAs of 16 clocks in source order, pinned by sched_barrier(0): behind MFMA pair i < 8 the
    // fragment reads RD(j) of slot i (NR reads spread over the 8 slots), behind pair i >= 8 (DMA) one LDS-DMA piece:
    // one piece per 32 clocks, as in sweep4b
#define HALF_STEP(FB, LO, NR, RD, DMA, HALF)                                                                     \
    {                                                                                                            \
        char* dst_ = smem + (is_cnt & 1) * SLOT_BYTES + (HALF) * STAGE_BYTES + wave * 4096;                      \
        const int so_ = ((HALF) ? 0 : is_cbase) + is_k * ROWB;                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                      \
                const int m_ = 2 * i + t;                                                                        \
                MFMA16(acc[(LO) + m_ / NB][m_ % NB], a[(LO) + m_ / NB], FB[m_ % NB]);                            \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if (i < 4) {                                                                                         \
                _Pragma("unroll") for (int j = (i * (NR)) / 4; j < ((i + 1) * (NR)) / 4; ++j) { RD(j) }          \
            } else if (DMA) {                                                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds((HALF) ? rq : rc, LDS_PTR(dst_ + (i & 3) * 1024), 16, voff2[i & 1], so_ + (i & 3) * piece, 0, 0); \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    }
#define RD_H0(j) if ((j) < MB / 2) a[MB / 2 + (j)] = LDF(sb + xa[0] + (MB / 2 + (j)) * 16 * ROWB); else b1[((j) - MB / 2) & (NB - 1)] = LDF(sb + xb[1] + ((j) - MB / 2) * 16 * ROWB);
#define RD_H1(j) if ((j) < NB / 2) b1[NB / 2 + (j)] = LDF(sb + xb[1] + (NB / 2 + (j)) * 16 * ROWB); else a[((j) - NB / 2) & (MB - 1)] = LDF(sb + xa[1] + ((j) - NB / 2) * 16 * ROWB);
#define RD_H2(j) a[MB / 2 + (j)] = LDF(sb + xa[1] + (MB / 2 + (j)) * 16 * ROWB);
#define RD_H3(j) if ((j) < NB) b0[(j) & (NB - 1)] = LDF(sn + xb[0] + (j) * 16 * ROWB); else a[((j) - NB) & (MB - 1)] = LDF(sn + xa[0] + ((j) - NB) * 16 * ROWB);

    if (t_begin < t_end) {
        ISSUE_HALF(0);
        ISSUE_HALF(1);
        ISSUE_ADVANCE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        v4i a[MB], b0[NB], b1[NB];
        READ_B(b0, smem, 0, 0, NB);
        READ_A(smem, 0, 0, MB / 2);
#ifndef NO_DMA
        ISSUE_HALF(0);                                         // corpus pieces of phase 1
#endif
        int ph = 0;
        for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0;
            for (int kk = 0; kk < nph; ++kk, ++ph) {
                const char* sb = smem + (ph & 1) * SLOT_BYTES;
                const char* sn = smem + ((ph + 1) & 1) * SLOT_BYTES;
                // H0: k-step 0, first half of the A tiles; reads: the second half of the A tiles, half of k-step 1's B;
                // query pieces of phase ph+1 -> the other slot
#ifndef NO_DMA
                HALF_STEP(b0, 0, MB / 2 + NB / 2, RD_H0, 1, 1);
                ISSUE_ADVANCE();
#else
                HALF_STEP(b0, 0, MB / 2 + NB / 2, RD_H0, 0, 1);
#endif
                // H1: k-step 0, second half; reads: the rest of k-step 1's B, the first half of its A
                HALF_STEP(b0, MB / 2, MB / 2 + NB / 2, RD_H1, 0, 0);
                // H2: k-step 1, first half; reads: the second half of its A.  Then every piece of phase ph+1 has landed and
                // every read of this slot has returned before anybody crosses the barrier
                HALF_STEP(b1, 0, MB / 2, RD_H2, 0, 0);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // H3: k-step 1, second half; reads: k-step 0 of phase ph+1 (all of B, the first half of A) from the other
                // slot; corpus pieces of phase ph+2 -> this slot
#ifndef NO_DMA
                HALF_STEP(b1, MB / 2, MB / 2 + NB, RD_H3, 1, 0);
#else
                HALF_STEP(b1, MB / 2, MB / 2 + NB, RD_H3, 0, 0);
#endif
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA results -> v_accvgpr_read (hipcc pads nothing for asm)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[mb][nb][r]));
                        best[nb] = fmaxf(best[nb], x);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink));
    if (pf_sink == 0x7fffffff) stats[511] = 1;             // keeps the prefetch register live to the end
    if (tid == 0) {
        stats[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0c;
        stats[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float m = fmaxf(best[nb], __shfl_xor(best[nb], 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        // one partial per (split, wm) and query
        if (g4 == 0) out[((int64_t)(split * 2 + wm)) * nq + qb * TILE + wn * (NB * 16) + nb * 16 + l15] = m;
    }
}

// ---- harness (as sweep4b_proto) -------------------------------------------------------------------------------------
__global__ void fill_kernel(uint16_t* x, int64_t n, uint32_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t s = (uint32_t)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        uint32_t s2 = s * 747796405u + 2891336453u;
        s2 ^= s2 >> 16; s2 *= 2246822519u; s2 ^= s2 >> 13;
        const float u1 = ((float)(s >> 8) + 1.0f) / 16777217.0f, u2 = (float)(s2 >> 8) / 16777216.0f;
        const float v = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * (1.0f / 64.0f);
        x[i] = (uint16_t)f32_to_bf16_bits(v);
    }
}

__global__ void naive_max_kernel(const uint16_t* q, const uint16_t* c, int d, int ld, int nc, int qrow, float* out) {
    __shared__ float red[256];
    float m = -INFINITY;
    for (int r = threadIdx.x; r < nc; r += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < d; ++k)
            s += __uint_as_float((uint32_t)q[(int64_t)qrow * ld + k] << 16) * __uint_as_float((uint32_t)c[(int64_t)r * ld + k] << 16);
        m = fmaxf(m, s);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

int main(int argc, char** argv) {
    const int nc = argc > 1 ? atoi(argv[1]) / TILE * TILE : 1000192;
    const int iters = argc > 2 ? atoi(argv[2]) : 3;
    const int flags = argc > 3 ? atoi(argv[3]) : 0;
    const int nq = 8192, d = 4096, rowbytes = d * 2, n_qblocks = nq / TILE, n_splits = 8;
    uint16_t *q, *c;
    float *out, *ref;
    unsigned long long* stats;
    CHECK(hipMalloc(&stats, 256 * 2 * 8));
    CHECK(hipMalloc(&q, (size_t)nq * rowbytes));
    CHECK(hipMalloc(&c, (size_t)nc * rowbytes));
    CHECK(hipMalloc(&out, (size_t)2 * n_splits * nq * 4));
    CHECK(hipMemset(out, 0xFF, (size_t)2 * n_splits * nq * 4));      // NaN pattern; fmaxf below ignores unwritten partials
    CHECK(hipMalloc(&ref, 4));
    fill_kernel<<<2048, 256>>>(q, (int64_t)nq * (rowbytes / 2), 1u);
    fill_kernel<<<2048, 256>>>(c, (int64_t)nc * (rowbytes / 2), 7u);
    CHECK(hipFuncSetAttribute((const void*)sweep8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int it = 0; it <= iters; ++it) {
        CHECK(hipEventRecord(e0));
        sweep8_kernel<<<n_qblocks * n_splits, 512, RING_BYTES>>>((const char*)q, (const char*)c, nq, nc, rowbytes, d * 2, n_qblocks,
                                                                   n_splits, out, stats, flags);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hs[512];
        CHECK(hipMemcpy(hs, stats, sizeof(hs), hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (int i = 0; i < 256; ++i) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
        const double phases = (double)(nc / TILE) / n_splits * (d * 2 / 64);
        printf("8 waves MB=%d NB=%d iter %d: %.3f ms  %.1f TFLOP/s  clock %.3f GHz  %.0f shader clocks per 64-byte phase (MFMA alone: 1024)\n", MB, NB,
               it, ms, 2.0 * nq * (double)nc * d / ms / 1e9, cyc / rt * 0.1, cyc / 256 / phases);
        fflush(stdout);
    }
    std::vector<float> ho((size_t)2 * n_splits * nq);
    CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    if (nc <= 65536) {
        int bad = 0;
        const int rows[8] = {0, 37, 255, 256, 4099, 8191, 130, 77};
        for (int t = 0; t < 8; ++t) {
            naive_max_kernel<<<1, 256>>>(q, c, d, rowbytes / 2, nc, rows[t], ref);
            float hr;
            CHECK(hipMemcpy(&hr, ref, 4, hipMemcpyDeviceToHost));
            float m = -INFINITY;
            for (int s = 0; s < 2 * n_splits; ++s) m = fmaxf(m, ho[(size_t)s * nq + rows[t]]);
            const bool ok = fabsf(m - hr) <= 1e-4f * fmaxf(1.f, fabsf(hr));
            printf("query %d: sweep %.6f naive %.6f %s\n", rows[t], m, hr, ok ? "ok" : "MISMATCH");
            bad += !ok;
        }
        printf(bad ? "CHECK FAILED\n" : "CHECK OK\n");
        return bad != 0;
    }
    return 0;
}
