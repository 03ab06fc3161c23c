// EXPERIMENT, not part of libabcsearch: the one-wave-per-SIMD contraction loop in the shape of the vendor's gfx950 GEMM
// (hipBLASLt "MT256x256x64 MI16x16x1", DirectToLds, 4 waves of 128x128): phases of 128 BYTES per row, so that every
// LDS-DMA piece is 8 rows x one full 128-byte line from the row-major operands, two 64 KB slots (double buffer), ONE
// barrier per phase.  Fragments are double-buffered per 16-k step (4 steps per phase), 32x32x16 MFMAs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 experiments/sweep4b_proto.hip -o experiments/sweep4b_proto
#include "../vllm_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define TILE 256
#define ROWB 128
#define STAGE_BYTES (TILE * ROWB)          // 32 KB per operand and phase
#define SLOT_BYTES (2 * STAGE_BYTES)
#define N_SLOTS 2
#ifndef AUX_C
#define AUX_C 0            /* cache policy of the corpus pieces: 2 = nt (streaming) */
#endif
#ifndef AUX_Q
#define AUX_Q 0
#endif
#define RING_BYTES (N_SLOTS * SLOT_BYTES)

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

template <int W>
__device__ __forceinline__ void sweep4b_body(char* smem, const char* __restrict__ q, const char* __restrict__ c, int nq, int nc,
                                             int rowbytes, int kbytes, int n_qblocks, int n_splits, float* __restrict__ out,
                                             unsigned long long* __restrict__ stats, int flags) {
    const uint64_t t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int wave = W;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    int qb, split;
    {
        const int id = blockIdx.x;
        const int gs = (flags >> 4) & 3;
        const int g = gs == 0 ? 2 : gs == 1 ? 1 : gs == 2 ? 4 : 8;
        const int per = (n_qblocks >> 3) * g;
        const int x = id & 7, j = id >> 3;
        qb = (x / g) * per + (j % per);
        split = (x % g) * (n_splits / g) + j / per;
    }
    const int n_tiles = nc / TILE;
    const int t_begin = (int)(((int64_t)split * n_tiles) / n_splits);
    const int t_end = (int)(((int64_t)(split + 1) * n_tiles) / n_splits);
    const int nph = kbytes / ROWB;

    // staging: wave w brings rows [64w, 64w+64) of both operands, 8 + 8 pieces of 8 rows x 128 B; the 16-byte chunk
    // index is XOR-swizzled with row & 7 on the source address (LDS-DMA destinations are lane-linear)
    const int voff = (wave * 64 + (lane >> 3)) * rowbytes + (((lane & 7) ^ (lane >> 3)) << 4);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (int64_t)qb * TILE * rowbytes, (flags & 64) ? 0u : (uint32_t)(TILE * rowbytes));
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(c + (int64_t)t_begin * TILE * rowbytes,
                                                (flags & 64) ? 0u : (uint32_t)((int64_t)(t_end - t_begin) * TILE * rowbytes));
    const int piece = 8 * rowbytes;

    int xa[4], xb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int x = l31 * ROWB + (((2 * s + h) ^ (l31 & 7)) << 4);
        xa[s] = wm * 128 * ROWB + x;
        xb[s] = STAGE_BYTES + wn * 128 * ROWB + x;
    }

    v16f acc[4][4];
    float best[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) best[nb] = -INFINITY;

    int is_k = 0, is_cbase = 0, is_cnt = 0;
    // half of a phase's pieces: HALF 0 = the corpus operand, 1 = the query operand (8 pieces each)
#define ISSUE_HALF(HALF)                                                                                         \
    {                                                                                                            \
        char* dst_ = smem + (is_cnt & 1) * SLOT_BYTES + (HALF) * STAGE_BYTES + wave * 8192;                      \
        const int so_ = ((HALF) ? 0 : is_cbase) + is_k * ROWB;                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds((HALF) ? rq : rc, LDS_PTR(dst_ + i * 1024), 16, voff, so_ + i * piece, 0, (HALF) ? AUX_Q : AUX_C); \
    }
#define ISSUE_ADVANCE()                                                                                          \
    {                                                                                                            \
        ++is_cnt;                                                                                                \
        if (++is_k == nph) {                                                                                     \
            is_k = 0;                                                                                            \
            is_cbase += TILE * rowbytes;                                                                         \
        }                                                                                                        \
    }
#define READ_A(F, SB, S) _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) F[mb] = *(const v4i*)((SB) + xa[S] + mb * 32 * ROWB);
#define READ_B(F, SB, S) _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) F[nb] = *(const v4i*)((SB) + xb[S] + nb * 32 * ROWB);
#define MMA(FA, FB)                                                                                             \
    _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)           \
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8s, FA[mb]),                  \
                                                              __builtin_bit_cast(v8s, FB[nb]), acc[mb][nb], 0, 0, 0);
    // 16 MFMAs with one fragment read behind each of the first 8 and (DMA = 1) one LDS-DMA piece behind each of the last 8
#ifndef SPARSE_DMA
#define SCHED(DMA)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                       \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        if (DMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                              \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
#else
// a wave cannot issue LDS-DMA pieces faster than about one per 64 clocks without stalling its MFMA stream (measured:
// one per 32-clock MFMA slot costs 1562 clocks per phase, one per two slots 1180): M r M D, eight times
#define SCHED(DMA)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        if (DMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                              \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
#endif

    if (t_begin < t_end) {
        ISSUE_HALF(0);
        ISSUE_HALF(1);
        ISSUE_ADVANCE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        v4i a0[4], b0[4], a1[4], b1[4];
        READ_A(a0, smem, 0);
#ifndef SPLIT_WAIT
        READ_B(b0, smem, 0);
#endif
#ifndef NO_DMA
        ISSUE_HALF(0);                                         // corpus pieces of phase 1
#ifdef BOTH_IN_STEP3
        ISSUE_HALF(1);
        ISSUE_ADVANCE();
#endif
#endif
        int ph = 0;
        for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0;
            for (int kk = 0; kk < nph; ++kk, ++ph) {
                const char* sb = smem + (ph & 1) * SLOT_BYTES;
                const char* sn = smem + ((ph + 1) & 1) * SLOT_BYTES;
                // step 0: reads of step 1; query pieces of phase ph+1 -> the other slot (its corpus pieces went out in
                // step 3 of the phase before: that slot has been free since the barrier of step 2 there)
#ifdef SPLIT_WAIT
                // the query operand of THIS phase became visible at the barrier that ended the phase before: its step-0
                // fragments are read here, right before their first use (exposed LDS latency, one more step of flight)
                READ_B(b0, sb, 0);
#endif
                READ_A(a1, sb, 1);
                READ_B(b1, sb, 1);
#if !defined(NO_DMA) && !defined(BOTH_IN_STEP3)
                ISSUE_HALF(1);
                ISSUE_ADVANCE();
#endif
                MMA(a0, b0);
#ifdef SPLIT_WAIT
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#endif
#if !defined(NO_DMA) && !defined(BOTH_IN_STEP3)
                SCHED(1);
#else
                SCHED(0);
#endif
                // step 1: reads of step 2
                READ_A(a0, sb, 2);
                READ_B(b0, sb, 2);
                MMA(a1, b1);
                SCHED(0);
                // step 2: reads of step 3; then every piece of phase ph+1 must have landed and every read of this
                // slot must have returned before anybody crosses the barrier
                READ_A(a1, sb, 3);
                READ_B(b1, sb, 3);
                MMA(a0, b0);
                SCHED(0);
#ifdef SPLIT_WAIT
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // the corpus pieces of ph+1 (the query pieces may fly on)
#else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // step 3: reads of step 0 of phase ph+1; corpus pieces of phase ph+2 -> this phase's slot, which nobody
                // reads any more (every wave's reads had returned before the barrier)
                READ_A(a0, sn, 0);
#ifndef SPLIT_WAIT
                READ_B(b0, sn, 0);
#endif
#ifndef NO_DMA
                ISSUE_HALF(0);
#ifdef BOTH_IN_STEP3
                ISSUE_HALF(1);
                ISSUE_ADVANCE();
#endif
#endif
                MMA(a1, b1);
#if defined(BOTH_IN_STEP3) && !defined(NO_DMA)
                // all 8 fragment reads first (a burst), then one piece behind every MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#elif !defined(NO_DMA)
                SCHED(1);
#else
                SCHED(0);
#endif
#ifdef SPLIT_WAIT
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                // the query pieces of ph+1 (issued before the corpus pieces of ph+2)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float x;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[mb][nb][r]));
                        best[nb] = fmaxf(best[nb], x);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tid == 0) {
        stats[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0c;
        stats[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const float m = fmaxf(best[nb], __shfl_xor(best[nb], 32, 64));
        if (h == 0) out[((int64_t)(split * 2 + wm)) * nq + qb * TILE + wn * 128 + nb * 32 + l31] = m;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void sweep4b_kernel(const char* __restrict__ q, const char* __restrict__ c, int nq, int nc, int rowbytes, int kbytes, int n_qblocks,
                    int n_splits, float* __restrict__ out, unsigned long long* __restrict__ stats, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) sweep4b_body<0>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
    else if (wave == 1) sweep4b_body<1>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
    else if (wave == 2) sweep4b_body<2>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
    else sweep4b_body<3>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
}

// ---- harness --------------------------------------------------------------------------------------------------------
__global__ void fill_kernel(uint16_t* x, int64_t n, uint32_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t s = (uint32_t)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        // gaussian elements with std 1/64 (unit-norm rows at d = 4096, the bench's data: the clock follows the data)
        uint32_t s2 = s * 747796405u + 2891336453u;
        s2 ^= s2 >> 16; s2 *= 2246822519u; s2 ^= s2 >> 13;
        const float u1 = ((float)(s >> 8) + 1.0f) / 16777217.0f, u2 = (float)(s2 >> 8) / 16777216.0f;
        const float v = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * (1.0f / 64.0f);
        x[i] = (uint16_t)f32_to_bf16_bits(v);
    }
}

__global__ void naive_max_kernel(const uint16_t* q, const uint16_t* c, int d, int ld, int nc, int qrow, float* out) {
    // one workgroup: max over all corpus rows of q[qrow] . c[row]  (fp32 accumulation)
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
    const int flags = argc > 3 ? atoi(argv[3]) : 0;     // 1: every workgroup stages query block 0; 2: every split starts at tile 0; 4: ... and stays there
    const int pad = argc > 4 ? atoi(argv[4]) : 0;           // bytes of padding after each row (both operands)
    const int nq = 8192, d = 4096, rowbytes = d * 2 + pad, n_qblocks = nq / TILE, n_splits = 8;
    uint16_t *q, *c;
    float *out, *ref;
    unsigned long long* stats;
    CHECK(hipMalloc(&stats, 256 * 2 * 8));
    CHECK(hipMalloc(&q, (size_t)nq * rowbytes));
    CHECK(hipMalloc(&c, (size_t)nc * rowbytes));
    CHECK(hipMalloc(&out, (size_t)2 * n_splits * nq * 4));
    CHECK(hipMalloc(&ref, 4));
    fill_kernel<<<2048, 256>>>(q, (int64_t)nq * (rowbytes / 2), 1u);
    fill_kernel<<<2048, 256>>>(c, (int64_t)nc * (rowbytes / 2), 7u);
    CHECK(hipFuncSetAttribute((const void*)sweep4b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int it = 0; it <= iters; ++it) {
        CHECK(hipEventRecord(e0));
        sweep4b_kernel<<<n_qblocks * n_splits, 256, RING_BYTES>>>((const char*)q, (const char*)c, nq, nc, rowbytes, d * 2, n_qblocks,
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
        printf("iter %d: %.3f ms  %.1f TFLOP/s  clock %.3f GHz  %.0f shader clocks per phase (MFMA alone: 1024)\n", it, ms,
               2.0 * nq * (double)nc * d / ms / 1e9, cyc / rt * 0.1, cyc / 256 / phases);
        fflush(stdout);
    }
    // check a few queries against the naive kernel on the first 4096 corpus rows... the sweep covered all rows, so
    // compare over the whole corpus only when it is small
    std::vector<float> ho((size_t)2 * n_splits * nq);
    CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    if (nc <= 65536) {
        int bad = 0;
        const int rows[6] = {0, 37, 255, 256, 4099, 8191};
        for (int t = 0; t < 6; ++t) {
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
