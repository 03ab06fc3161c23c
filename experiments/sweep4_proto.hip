// EXPERIMENT, not part of libabcsearch: contraction loop of a one-wave-per-SIMD sweep (DESIGN.md 4.1,
// "a larger wave tile").  4 waves per workgroup, 256x256 tile, wave tile 128x128 (256 accumulator
// registers), fragments double-buffered per 16-k step inside the wave, one barrier per 64-byte phase.
// Per phase and CU: 64 KB of fragment reads (8-wave kernel: 96 KB) + 32 KB of LDS-DMA staging.
// Output: the maximum score of every query over each half-split (enough to check the contraction).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 experiments/sweep4_proto.hip -o experiments/sweep4_proto
//   experiments/sweep4_proto [nc=1000000] [iters=3]
#include "../vllm_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define TILE 256
#define ROWB 64
#define STAGE_BYTES (TILE * ROWB)
#define SLOT_BYTES (2 * STAGE_BYTES)
#define N_SLOTS 4
#ifndef DMA_SZ
#define DMA_SZ 16          /* -DDMA_SZ=4: same instruction count, a quarter of the bytes (timing only) */
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

// W: the wave's index in the workgroup, a template parameter because the four waves run DIFFERENT schedules: all of
// them reaching the texture-address unit with their LDS-DMA pieces at the same point of the same program costs every
// wave ~66 clocks per piece (4 x 16.5: the unit takes one 1 KB piece per 16.5 clocks) and its MFMA stream waits behind
// the piece.  Wave W issues its 8 pieces of a phase in quarter W of the phase's 32 MFMA slots.
template <int W>
__device__ __forceinline__ void sweep4_body(char* smem, const char* __restrict__ q, const char* __restrict__ c, int nq, int nc,
                                            int rowbytes, int kbytes, int n_qblocks, int n_splits, float* __restrict__ out,
                                            unsigned long long* __restrict__ stats, int flags) {
    const uint64_t t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef LOCKSTEP
    constexpr int SLOT = 1;                                          // every wave issues in the second quarter
#else
    constexpr int SLOT = W;
#endif
    constexpr int wave = W;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    int qb, split;
    {
        const int id = blockIdx.x;
        const int gs = (flags >> 4) & 3;
        const int g = gs == 0 ? 4 : gs == 1 ? 1 : gs == 2 ? 2 : 8;     // XCDs that share a group of query blocks
        const int per = (n_qblocks >> 3) * g;
        const int x = id & 7, j = id >> 3;
        qb = (x / g) * per + (j % per);
        split = (x % g) * (n_splits / g) + j / per;
    }
    const int n_tiles = nc / TILE;                                   // prototype: nc % 256 == 0
    const int t_begin = (int)(((int64_t)split * n_tiles) / n_splits);
    const int t_end = (int)(((int64_t)(split + 1) * n_tiles) / n_splits);
    const int nph = kbytes / ROWB;                                     // rows are `rowbytes` apart (>= kbytes: padded)

    // staging: wave w brings rows [64w, 64w+64) of both operands, 4 + 4 pieces of 16 rows x 64 B
    const int row0 = wave * 64 + (lane >> 2);
    const int swz = (lane >> 4) & 3;
    const int voff = row0 * rowbytes + (((lane & 3) ^ swz) << 4);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (int64_t)((flags & 1) ? 0 : qb) * TILE * rowbytes, (flags & 64) ? 0u : (uint32_t)(TILE * rowbytes));   // flags & 64: every load out of range (no memory traffic)
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(c + (int64_t)((flags & 2) ? 0 : t_begin) * TILE * rowbytes,
                                                (flags & 64) ? 0u : (uint32_t)((int64_t)(t_end - t_begin) * TILE * rowbytes));
    const int piece = 16 * rowbytes;

    const int swz_r = (l31 >> 2) & 3;
    int xa[2], xb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int x = l31 * ROWB + (((2 * s + h) ^ swz_r) << 4);
        xa[s] = wm * 128 * ROWB + x;
        xb[s] = STAGE_BYTES + wn * 128 * ROWB + x;
    }

    v16f acc[4][4];
    float best[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) best[nb] = -INFINITY;

    int is_k = 0, is_cbase = 0, is_cnt = 0;
    auto issue = [&]() {
        char* dst = smem + (is_cnt & (N_SLOTS - 1)) * SLOT_BYTES + wave * 4096;
#ifdef PACKED
        {
            // TIMING ONLY: "phase-major" operand layout, every phase of a tile is one contiguous 16 KB block
            // ([tile][phase][row][64 B]); same bytes, same reuse, contiguous 1 KB per instruction
            const int so = is_cbase + is_k * STAGE_BYTES + wave * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, LDS_PTR(dst + i * 1024), DMA_SZ, lane * 16, so + i * 1024, 0, 0);
            const int sq = is_k * STAGE_BYTES + wave * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, LDS_PTR(dst + STAGE_BYTES + i * 1024), DMA_SZ, lane * 16, sq + i * 1024, 0, 0);
        }
#else
        {
        const int so = is_cbase + is_k * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, LDS_PTR(dst + i * 1024), 16, voff, so + i * piece, 0, 0);
        const int sq = is_k * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, LDS_PTR(dst + STAGE_BYTES + i * 1024), 16, voff, sq + i * piece, 0, 0);
        }
#endif
        ++is_cnt;
        if (++is_k == nph) {
            is_k = 0;
            if (!(flags & 4)) is_cbase += TILE * rowbytes;       // flags & 4: every tile is the split's first (L2-resident)
        }
    };
#define READ_A(F, SB, S) _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) F[mb] = *(const v4i*)((SB) + xa[S] + mb * 32 * ROWB);
#define READ_B(F, SB, S) _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) F[nb] = *(const v4i*)((SB) + xb[S] + nb * 32 * ROWB);
#ifdef NO_MMA                                    /* staging only: how fast can a CU take its 32 KB per phase? */
#undef READ_A
#undef READ_B
#define READ_A(F, SB, S)
#define READ_B(F, SB, S)
#define MMA(FA, FB)
#else
#define MMA(FA, FB)                                                                                             \
    _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)           \
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8s, FA[mb]),                  \
                                                              __builtin_bit_cast(v8s, FB[nb]), acc[mb][nb], 0, 0, 0);
#endif

#ifdef STAGE_VGPR
    // TIMING ONLY (phase-major addressing, wrong data): operands go global -> VGPR (four register sets, loaded four
    // phases ahead) -> ds_write_b128 into a ring of TWO LDS slots, instead of LDS-DMA.  Two shorter instructions per
    // 1 KB piece, each of which may fit into the 32-clock gap between two MFMAs.
    if (t_begin < t_end) {
        v4i S0[8], S1[8], S2[8], S3[8];
        int lk = 0, lcbase = 0;
#define LOADSET(S)                                                                                               \
    {                                                                                                            \
        const int so_ = lcbase + lk * STAGE_BYTES + wave * 4096, sq_ = lk * STAGE_BYTES + wave * 4096;           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            S[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16, so_ + i * 1024, 0)); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            S[4 + i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rq, lane * 16, sq_ + i * 1024, 0)); \
        if (++lk == nph) {                                                                                       \
            lk = 0;                                                                                              \
            lcbase += TILE * rowbytes;                                                                           \
        }                                                                                                        \
    }
#define WRITESET(S, SLOTOFF)                                                                                     \
    {                                                                                                            \
        char* d_ = smem + (SLOTOFF) + wave * 4096 + lane * 16;                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *(v4i*)(d_ + i * 1024) = S[i];                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *(v4i*)(d_ + STAGE_BYTES + i * 1024) = S[4 + i];            \
    }
        LOADSET(S0);
        LOADSET(S1);
        LOADSET(S2);
        LOADSET(S3);
        WRITESET(S0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        v4i a0[4], b0[4], a1[4], b1[4];
        READ_A(a0, smem, 0);
        READ_B(b0, smem, 0);
#define PHASEV(SW, SL, CUR, NXT)                                                                                 \
    {                                                                                                            \
        READ_A(a1, smem + (CUR), 1);                                                                             \
        READ_B(b1, smem + (CUR), 1);                                                                             \
        WRITESET(SW, NXT);                                                                                       \
        MMA(a0, b0);                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                   \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        READ_A(a0, smem + (NXT), 0);                                                                             \
        READ_B(b0, smem + (NXT), 0);                                                                             \
        LOADSET(SL);                                                                                             \
        MMA(a1, b1);                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                   \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
        for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0;
            for (int kk = 0; kk < nph; kk += 4) {              // phase j: register set j % 4, LDS slot j % 2
                PHASEV(S1, S0, 0, SLOT_BYTES);
                PHASEV(S2, S1, SLOT_BYTES, 0);
                PHASEV(S3, S2, 0, SLOT_BYTES);
                PHASEV(S0, S3, SLOT_BYTES, 0);
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
    }
#else
    if (t_begin < t_end) {
        issue();
        issue();
        issue();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        v4i a0[4], b0[4], a1[4], b1[4];
        READ_A(a0, smem, 0);
        READ_B(b0, smem, 0);
        int ph = 0;
        for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0;
            for (int kk = 0; kk < nph; ++kk, ++ph) {
                const char* sb = smem + (ph & (N_SLOTS - 1)) * SLOT_BYTES;
                const char* sn = smem + ((ph + 1) & (N_SLOTS - 1)) * SLOT_BYTES;
                READ_A(a1, sb, 1);
                READ_B(b1, sb, 1);
#ifndef NO_DMA
                if constexpr (SLOT < 2) issue();               // phase ph+3 -> slot (ph-1)%4, free since the last barrier
#endif
                MMA(a0, b0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {                  // MFMA slots 0-7: fragment reads (and wave 0's pieces)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#ifndef NO_DMA
                    if constexpr (SLOT == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {                  // slots 8-15 (wave 1's pieces)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#ifndef NO_DMA
                    if constexpr (SLOT == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                // own pieces of phase ph+1 landed, own reads of slot ph done
                if constexpr (SLOT < 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
#ifndef NO_BARRIER
                __builtin_amdgcn_s_barrier();
#endif
                __builtin_amdgcn_sched_barrier(0);
                READ_A(a0, sn, 0);
                READ_B(b0, sn, 0);
#ifndef NO_DMA
                if constexpr (SLOT >= 2) issue();
#endif
                MMA(a1, b1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {                  // slots 16-23 (wave 2's pieces)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#ifndef NO_DMA
                    if constexpr (SLOT == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {                  // slots 24-31 (wave 3's pieces)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#ifndef NO_DMA
                    if constexpr (SLOT == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float x;                             // volatile asm keeps the AGPR reads where they are written:
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[mb][nb][r]));   // one block at a time
                        best[nb] = fmaxf(best[nb], x);
                    }
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    if (tid == 0) {
        stats[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0c;          // shader clocks
        stats[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0r;  // 100 MHz ticks
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const float m = fmaxf(best[nb], __shfl_xor(best[nb], 32, 64));
        if (h == 0) out[((int64_t)(split * 2 + wm)) * nq + qb * TILE + wn * 128 + nb * 32 + l31] = m;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void sweep4_kernel(const char* __restrict__ q, const char* __restrict__ c, int nq, int nc, int rowbytes, int kbytes, int n_qblocks,
                   int n_splits, float* __restrict__ out, unsigned long long* __restrict__ stats, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) sweep4_body<0>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
    else if (wave == 1) sweep4_body<1>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
    else if (wave == 2) sweep4_body<2>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
    else sweep4_body<3>(smem, q, c, nq, nc, rowbytes, kbytes, n_qblocks, n_splits, out, stats, flags);
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
    CHECK(hipFuncSetAttribute((const void*)sweep4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int it = 0; it <= iters; ++it) {
        CHECK(hipEventRecord(e0));
        sweep4_kernel<<<n_qblocks * n_splits, 256, RING_BYTES>>>((const char*)q, (const char*)c, nq, nc, rowbytes, d * 2, n_qblocks,
                                                                   n_splits, out, stats, flags);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hs[512];
        CHECK(hipMemcpy(hs, stats, sizeof(hs), hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (int i = 0; i < 256; ++i) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
        const double phases = (double)(nc / TILE) / n_splits * (d * 2 / ROWB);
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
